"""Host coder scaling on the GPU box: 24 images x 128 maps x 1536 symbols, real-ish statistics."""
import os, sys, time
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoencoder_based_image_compression_amd.kodak.lossless import compression

rng = numpy.random.RandomState(0)
scale = rng.uniform(0.05, 1.5, size=(1, 128, 1))
sym = numpy.round(rng.laplace(size=(24, 128, 1536))*scale).astype(numpy.int16)
probs = numpy.clip(rng.rand(128, 10), 0.05, 0.95)
print('cpu_count', os.cpu_count(), 'symbols', sym.size)
for mode in (True, False):
    for nt in (1, 8, 16, 32, 64, 128, 254):
        compression.code_planar_symbols(sym, probs, 67, nb_threads=nt, roundtrip=mode)
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            (rec, bits) = compression.code_planar_symbols(sym, probs, 67, nb_threads=nt, roundtrip=mode)
            ts.append(time.perf_counter() - t)
        print('roundtrip' if mode else 'encode   ', 'threads', nt, 'min ms', round(min(ts)*1e3, 3), 'median ms', round(sorted(ts)[2]*1e3, 3),
              'Msym/s', round(sym.size/min(ts)/1e6, 1), 'bits/sym', round(bits.sum()/sym.size, 3))
