"""Why do maps of a large image leave the SIMD encoder for the general kernel? Host-side census of the retry conditions
(coder_simd.hip: invalid probability in any context, stream near capacity) for bench.py's synthetic setup."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
import bench
from autoencoder_based_image_compression_amd import pipeline, device as dev, _native
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var

(h, w, batch) = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
variables = bench.synthetic_model(1.)
device = torch.device('cuda', 0)
images = torch.from_numpy(bench.synthetic_images(1000, batch, h, w)).to(device)
enc = pipeline.DeviceEncoder(variables, False, device)
y0 = enc(images)
mm = dev.map_means(y0).cpu().numpy()
probs = lossless_stats.compute_binary_probabilities(y0.cpu().numpy(), variables[var.BIN_WIDTHS_NAME], mm, 10)
bad_rows = numpy.where(~((probs > 0.) & (probs < 1.)).all(axis=1))[0]
print('probability rows with a value outside ]0, 1[:', bad_rows.size, bad_rows[:10], probs[bad_rows[:3]] if bad_rows.size else '')
y = y0.cpu().numpy()
sym = numpy.round((y - mm)/variables[var.BIN_WIDTHS_NAME]).astype(numpy.int16)
print('max |symbol|', numpy.abs(sym).max(), 'map_size', (h//16)*(w//16))
cap = _native.coder().eae_coder_stream_capacity_bytes((h//16)*(w//16), 10)
print('stream capacity bytes per map and stream:', cap)
from autoencoder_based_image_compression_amd.kodak.lossless import compression
planar = torch.from_numpy(numpy.ascontiguousarray(sym.transpose(0, 3, 1, 2).reshape(batch, 128, -1))).to(device)
(_, nb_bits) = compression.code_planar_symbols_device(planar, probs, 67)
bits = nb_bits.reshape(-1)
print('bits per map: mean', bits.mean(), 'max', bits.max(), 'maps over 2048 bits', int((bits > 2048).sum()), 'over 6144', int((bits > 6144).sum()),
      'over 14336', int((bits > 14336).sum()), 'of', bits.size)
