"""BatchCodec on the bench's images in several modes: which mode breaks the round trip, and on which maps."""
import os, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
import bench
from autoencoder_based_image_compression_amd import codec, device as dev, pipeline
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 24
variables = bench.synthetic_model(1.0)
images = torch.from_numpy(bench.synthetic_images(1000, batch, 512, 768)).cuda()
bin_widths = variables[var.BIN_WIDTHS_NAME]
y = pipeline.DeviceEncoder(variables, False)(images)
map_mean = dev.map_means(y).cpu().numpy()
probabilities = lossless_stats.compute_binary_probabilities(y.cpu().numpy(), bin_widths, map_mean, 10)
for (name, kw) in (('1 stream, launches, depth 1', dict(nb_in_flight=1)), ('1 stream, launches, depth 3', dict(nb_in_flight=3)),
                   ('2 streams, launches', dict(nb_in_flight=3, nb_transform_streams=2)),
                   ('1 stream, graphs', dict(nb_in_flight=3, use_graphs=True)),
                   ('2 streams, graphs', dict(nb_in_flight=3, nb_transform_streams=2, use_graphs=True))):
    c = codec.BatchCodec(variables, False, bin_widths, map_mean, probabilities, 67, batch, 512, 768, **kw)
    failures = 0
    first = None
    tickets = [c.submit(images) for _ in range(12)]
    for (k, t) in enumerate(tickets):
        try:
            t.result()
        except Exception as exc:
            failures += 1
            if first is None:
                first = (k, type(exc).__name__, str(exc)[:80].replace('\n', ' '))
    # statuses of the last slots
    bad = []
    for slot in range(c.nb_slots):
        res = c._views(c._pinned_out[slot])[0].numpy()
        w = numpy.flatnonzero(res[2])
        if w.size:
            bad.append((slot, w[:6].tolist(), res[2][w[:6]].tolist(), (w[:6] % 128).tolist()))
    print(name, '-> failures', failures, 'of 12', first, 'bad maps per slot', bad)
    c.close()
