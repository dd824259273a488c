import numpy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoencoder_based_image_compression_amd.kodak.lossless import stats
y = (numpy.random.RandomState(15).standard_normal(size=(5, 32, 48, 128))*3 + 0.7).astype(numpy.float32)
got = stats.compute_map_mean(y)
ref = numpy.mean(y, axis=(0, 1, 2)); exact = numpy.mean(y.astype(numpy.float64), axis=(0, 1, 2))
print(numpy.abs(got.astype(numpy.float64) - exact).max(), numpy.spacing(numpy.float32(numpy.abs(exact).max())), numpy.abs(got - ref).max(), got[:4], exact[:4], ref[:4])
