"""BatchCodec(use_graphs=True) with ONE transform stream: the configuration that failed in capture (round 2)."""
import os, sys, traceback
import numpy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autoencoder_based_image_compression_amd import codec, device as dev, pipeline
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats
nts = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variables = bench.synthetic_model(1.)
images = torch.from_numpy(bench.synthetic_images(1000, 4, 64, 96)).cuda()
enc = pipeline.DeviceEncoder(variables, False, 'cuda')
y0 = enc(images)
mm = dev.map_means(y0).cpu().numpy()
probs = lossless_stats.compute_binary_probabilities(y0.cpu().numpy(), variables[var.BIN_WIDTHS_NAME], mm, 10)
try:
    with codec.BatchCodec(variables, False, variables[var.BIN_WIDTHS_NAME], mm, probs, 67, 4, 64, 96, use_graphs=True, nb_transform_streams=nts) as c:
        for i in range(6):
            r = c.submit(images).result()
            print(i, int(r['nb_bits'].sum()))
except Exception:
    traceback.print_exc()
