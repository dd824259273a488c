"""conv2+GDN2 on 24 Kodak-sized images as ONE launch vs two back-to-back launches on the same stream (a images, then 24 - a):
16 images are exactly one round of three waves per SIMD; the remaining 8 take the half-channel path (NT = 2 + GDN pass)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autoencoder_based_image_compression_amd import device as dev, pipeline

variables = bench.synthetic_model(1.)
enc = pipeline.DeviceEncoder(variables, False, 'cuda')
v = enc.v
n = 24
x = torch.randn((n, 128, 192, 128), device='cuda')*0.5
out = torch.empty((n, 64, 96, 128), device='cuda')

def conv(lo, hi):
    dev.conv5x5s2(x[lo:hi], enc.w2, v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], v['encoder/beta_2'], out=out[lo:hi])

def run(parts, reps=30):
    times = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for (lo, hi) in parts:
            conv(lo, hi)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    times.sort()
    return times[len(times)//2]

ref = None
for parts in ([(0, 24)], [(0, 16), (16, 24)], [(0, 8), (8, 24)], [(0, 16), (16, 20), (20, 24)], [(0, 12), (12, 24)], [(0, 20), (20, 24)], [(0, 16)], [(16, 24)]):
    ms = run(parts)
    print(parts, 'ms', round(ms, 4))
    if ref is None:
        ref = out.clone()
    elif parts[-1][1] == 24 and parts[0][0] == 0:
        assert torch.equal(out, ref)
