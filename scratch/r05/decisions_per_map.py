"""r05: binary decisions per feature map of the bench's synthetic Kodak image (truncated unary, L = 10: |s| + 1 decisions for |s| < L, L otherwise):
the serial cores run as long as the LONGEST map of a group of 64 needs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy, torch
import bench
from autoencoder_based_image_compression_amd import device as dev, pipeline
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
for width in (1.0, 0.05):
    v = bench.synthetic_model(width)
    images = torch.from_numpy(bench.synthetic_images(1000, 1, 512, 768)).cuda()
    y = pipeline.DeviceEncoder(v, False, 'cuda')(images)
    mean = dev.map_means(y)
    res = dev.quantize_maps(y.view(1, -1, 128), torch.from_numpy(v[var.BIN_WIDTHS_NAME]).cuda(), mean, want_symbols=True)
    s = numpy.abs(res['symbols'].cpu().numpy().astype(numpy.int64)).reshape(128, -1)
    L = 10
    nd = numpy.where(s < L, s + 1, L).sum(axis=1)
    order = numpy.argsort(nd)
    print('bin width %.3f: decisions per map: min %d median %d max %d (symbols per map %d); the two groups of 64 maps: max %d and %d' % (
        width, nd.min(), int(numpy.median(nd)), nd.max(), s.shape[1], nd[:64].max(), nd[64:].max()))
