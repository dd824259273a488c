"""Cycles of the three wavefronts of block 0 of coder_pipe_kernel (built with -DEAE_PIPE_PROBE into a private library):
SRC=coder_simd EXTRA=-DEAE_PIPE_PROBE SCRIPT=r05/pipe_probe.py bash scratch/variant.sh [bin width]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch          # noqa: E402
import bench          # noqa: E402
from autoencoder_based_image_compression_amd import _native, device as dev, pipeline          # noqa: E402
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var          # noqa: E402
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats          # noqa: E402

torch.cuda.set_device(0)
width = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
v = bench.synthetic_model(width)
images = torch.from_numpy(bench.synthetic_images(1000, 1, 512, 768)).cuda()
y = pipeline.DeviceEncoder(v, False, 'cuda')(images)
mean = dev.map_means(y)
bw = v[var.BIN_WIDTHS_NAME]
prob = torch.from_numpy(lossless_stats.compute_binary_probabilities(y.cpu().numpy(), bw, mean.cpu().numpy(), 10)).cuda()
q = dev.quantize_maps(y, torch.from_numpy(bw).cuda(), mean, want_symbols=True)
symbols = q['symbols'].reshape(128, -1)
rows = torch.arange(128, dtype=torch.int32)
rows[67] = -1
rows = rows.cuda()
streams = dev.CoderStreams(128, symbols.shape[1], 10, 'cuda')
ws = dev.coder_trailing_workspace(128, symbols.shape[1], 10, 'cuda')
for _ in range(3):
    dev.coder_roundtrip_fused(symbols, prob, rows, 10, out=streams, workspace=ws)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ['EAE_HIP_LIB'])
out = (ctypes.c_ulonglong*12)()
assert lib.eae_hip_debug_pipe_probe(out) == 0
for (r, name) in enumerate(('encoder core', 'bit writer', 'decoder core')):
    (total, wait, rounds, hwid) = out[4*r:4*r + 4]
    print('{0:13s} {1:9d} cycles in all (clock64 ticks, about the shader clock){2:.0f}, {3:9d} waiting, {4:5d} rounds; HW_ID {5:#x}: SIMD {6}, CU {7}, SE {8}'.format(
        name, total, 0, wait, rounds, hwid, (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 13) & 7))
out2 = (ctypes.c_ulonglong*8)()
assert lib.eae_hip_debug_pipe_probe2(out2) == 0
print('decoder: every stream complete at round {0}, cycle {1}; fast rounds {2}, lost bets {3}'.format(*out2[:4]))
print('bits per map', float(streams.bac_bits.float().mean()), 'status any', bool(streams.status.cpu().numpy().any()))
