"""A few fused round trips of one image's maps, for rocprofv3 --kernel-trace --stats (which kernels take the time)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch          # noqa: E402
import bench          # noqa: E402
from autoencoder_based_image_compression_amd import device as dev, pipeline          # noqa: E402
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var          # noqa: E402
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats          # noqa: E402

torch.cuda.set_device(0)
width = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
v = bench.synthetic_model(width)
images = torch.from_numpy(bench.synthetic_images(1000, batch, 512, 768)).cuda()
y = pipeline.DeviceEncoder(v, False, 'cuda')(images)
mean = dev.map_means(y)
bw = v[var.BIN_WIDTHS_NAME]
prob = torch.from_numpy(lossless_stats.compute_binary_probabilities(y.cpu().numpy(), bw, mean.cpu().numpy(), 10)).cuda()
q = dev.quantize_maps(y, torch.from_numpy(bw).cuda(), mean, want_symbols=True)
symbols = q['symbols'].reshape(batch*128, -1)
rows = torch.arange(128, dtype=torch.int32).repeat(batch)
rows[67::128] = -1
rows = rows.cuda()
streams = dev.CoderStreams(batch*128, symbols.shape[1], 10, 'cuda')
ws = dev.coder_trailing_workspace(batch*128, symbols.shape[1], 10, 'cuda')
for _ in range(20):
    dev.coder_roundtrip_fused(symbols, prob, rows, 10, out=streams, workspace=ws)
torch.cuda.synchronize()
for _ in range(20):
    dev.coder_encode_batch(symbols, prob, rows, 10, out=streams, workspace=ws)
    dev.coder_decode_batch(streams, prob, rows, expected=symbols, workspace=ws)
torch.cuda.synchronize()
print('status any', bool(streams.status.cpu().numpy().any()))
