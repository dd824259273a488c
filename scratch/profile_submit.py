import cProfile, pstats, os, sys, numpy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autoencoder_based_image_compression_amd import codec
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
v = bench.synthetic_model(1.)
probs = numpy.full((128, 10), 0.7)
c = codec.BatchCodec(v, False, v[var.BIN_WIDTHS_NAME], numpy.zeros(128, numpy.float32), probs, 67, 1, 512, 768, nb_in_flight=3)
x = torch.from_numpy(bench.synthetic_images(1, 1, 512, 768)).cuda()
for _ in range(30): c.submit(x)
c.drain()
pr = cProfile.Profile()
pr.enable()
for _ in range(300): c.submit(x)
pr.disable()
c.drain()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(18)
