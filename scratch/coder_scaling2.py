"""Host coder on bench-like statistics (~0.4 bits/symbol, mostly zeros), 24 x 128 x 1536 symbols."""
import os, sys, time
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoencoder_based_image_compression_amd.kodak.lossless import compression
from autoencoder_based_image_compression_amd import _native
rng = numpy.random.RandomState(0)
scale = rng.uniform(0.02, 0.45, size=(1, 128, 1))
sym = numpy.round(rng.laplace(size=(24, 128, 1536))*scale).astype(numpy.int16)
# probabilities from the data itself (like stats.compute_binary_probabilities)
L = 10
zeros = numpy.zeros((128, L), dtype=numpy.int64); ones = numpy.zeros((128, L), dtype=numpy.int64)
planar = numpy.ascontiguousarray(sym.transpose(1, 0, 2).reshape(128, -1))
_native.coder().eae_coder_count_binary_decisions(128, planar.shape[1], _native.ptr(planar, _native.c_i16p), L, _native.ptr(zeros, _native.c_i64p), _native.ptr(ones, _native.c_i64p), 8)
with numpy.errstate(invalid='ignore'):
    probs = zeros/(zeros + ones).astype(numpy.float64)
probs[numpy.isnan(probs)] = 0.5; probs[probs == 0.] = 0.01; probs[probs == 1.] = 0.99
for mode in (True, False):
    for nt in (1, 16, 30, 62, 126, 254):
        compression.code_planar_symbols(sym, probs, 67, nb_threads=nt, roundtrip=mode)
        ts = []
        for _ in range(7):
            t = time.perf_counter()
            (rec, bits) = compression.code_planar_symbols(sym, probs, 67, nb_threads=nt, roundtrip=mode)
            ts.append(time.perf_counter() - t)
        print('roundtrip' if mode else 'encode   ', 'threads', nt, 'min ms', round(min(ts)*1e3, 3), 'median ms', round(sorted(ts)[3]*1e3, 3),
              'Msym/s', round(sym.size/min(ts)/1e6, 1), 'bits/sym', round(bits.sum()/sym.size, 3), 'zero frac', round(float((sym == 0).mean()), 3))
