"""One image at a time (submit -> result): where the host side of the 1.19 ms goes. Stamps (perf_counter, us from submit start):
submit returns; the worker has the job; the worker sees the step counters; the worker has set the ticket; result() returns."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import numpy          # noqa: E402
import torch          # noqa: E402
import bench          # noqa: E402
from autoencoder_based_image_compression_amd import codec, device as dev, pipeline          # noqa: E402
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var          # noqa: E402
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats          # noqa: E402

torch.cuda.set_device(0)
sys.setswitchinterval(1e-4)
v = bench.synthetic_model(1.0)
images = torch.from_numpy(bench.synthetic_images(1000, 1, 512, 768)).cuda()
y = pipeline.DeviceEncoder(v, False, 'cuda')(images)
mean = dev.map_means(y).cpu().numpy()
prob = lossless_stats.compute_binary_probabilities(y.cpu().numpy(), v[var.BIN_WIDTHS_NAME], mean, 10)
stamps = {}
orig_wait = codec._Worker._wait_sequence


def wait_sequence(self, words, expected):
    stamps['worker_has_job'] = time.perf_counter()
    orig_wait(self, words, expected)
    stamps['worker_sees_counters'] = time.perf_counter()


codec._Worker._wait_sequence = wait_sequence
orig_set = codec.threading.Event.set
rows = []
with codec.BatchCodec(v, False, v[var.BIN_WIDTHS_NAME], mean, prob, 67, 1, 512, 768, nb_in_flight=1, nb_transform_streams=1, use_graphs=True) as c:
    for _ in range(20):
        c.submit(images).result()
    for _ in range(200):
        t0 = time.perf_counter()
        ticket = c.submit(images)
        t1 = time.perf_counter()
        ticket.result()
        t2 = time.perf_counter()
        rows.append((t1 - t0, stamps['worker_has_job'] - t0, stamps['worker_sees_counters'] - t0, t2 - t0))
a = numpy.array(rows)*1e6
med = numpy.median(a, axis=0)
print('median over 200 images, us from the start of submit(): submit returns {0:.0f}; the worker has the job {1:.0f}; the worker sees the step counters '
      '{2:.0f}; result() returns {3:.0f}'.format(*med))
print('=> after the device is through: worker post-processing + hand-over to the waiting thread {0:.0f} us'.format(med[3] - med[2]))
