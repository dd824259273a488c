"""The bench's own symbols (24 Kodak-sized images, bin width argv[1]) through the batch coder, against the host library map by
map: where do bytes / bit counts / decoded symbols first differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy, torch
import bench
import test_coder_device as T
from autoencoder_based_image_compression_amd import device as dev, pipeline
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats

bw = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 24
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
variables = bench.synthetic_model(bw)
images = torch.from_numpy(bench.synthetic_images(seed, batch, 512, 768)).cuda()
bin_widths = variables[var.BIN_WIDTHS_NAME]
y = pipeline.DeviceEncoder(variables, False)(images)
map_mean = dev.map_means(y)
probabilities = lossless_stats.compute_binary_probabilities(y.cpu().numpy(), bin_widths, map_mean.cpu().numpy(), 10)
q = dev.quantize_maps(y, torch.from_numpy(bin_widths).cuda(), map_mean, want_symbols=True)
planar = q['symbols'].reshape(batch*128, -1).cpu().numpy()
rows = numpy.tile(numpy.arange(128, dtype=numpy.int32), batch)
rows[67::128] = -1
(streams, sym, p, r) = T.batch_code(dev, planar, probabilities, rows)
st = streams.status.cpu().numpy()
print('encode statuses', {int(k): int((st == k).sum()) for k in numpy.unique(st)})
try:
    T.assert_equals_host(streams, planar, probabilities, rows, 'bench')
    print('encode: equals the host coder')
except AssertionError as exc:
    print('ENCODE DIFFERS', str(exc)[:400])
enc_status = streams.status.clone()
dev.coder_decode_batch(streams, p, r, expected=sym)
st2 = streams.status.cpu().numpy()
print('verify statuses', {int(k): int((st2 == k).sum()) for k in numpy.unique(st2)})
bad = numpy.flatnonzero(st2 != st)
print('maps whose status changed in verify:', bad[:20], [int(rows[b]) for b in bad[:20]])
out = dev.coder_decode_batch(streams, p, r).cpu().numpy()
good = rows >= 0
diff = numpy.flatnonzero((out != planar).any(axis=1) & good)
print('maps decoded differently:', diff[:20])
for m in diff[:3]:
    w = numpy.flatnonzero(out[m] != planar[m])
    print('map', m, 'row', rows[m], 'first wrong symbols at', w[:10], 'got', out[m][w[:10]], 'want', planar[m][w[:10]],
          'bac_bits', int(streams.bac_bits[m]), 'byp_bits', int(streams.bypass_bits[m]), 'absmax', int(numpy.abs(planar[m].astype(int)).max()),
          'p', probabilities[rows[m]][:4])
