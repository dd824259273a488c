import numpy, sys, os
sys.path.insert(0, '/root/repo')
from autoencoder_based_image_compression_amd.kodak.lossless import stats
g = numpy.load('/root/repo/tests/golden/tools_golden.npz')
y = g['stats_y']
d = stats.map_divergences(y)
bad = numpy.flatnonzero(d != g['stats_divergences'])
print('bad', bad[:10], len(bad))
for i in bad[:3]:
    (lo, hi, counts) = stats._unit_interval_counts(y)[i]
    ref = numpy.histogram(y[:, :, :, i], bins=numpy.linspace(numpy.floor(y[..., i].min()), numpy.ceil(y[..., i].max()), num=int(numpy.ceil(y[..., i].max()) - numpy.floor(y[..., i].min())) + 1))[0]
    print(i, lo, hi, counts, ref, d[i], g['stats_divergences'][i])
