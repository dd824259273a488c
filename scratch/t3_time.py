"""Times transpose_conv_3 (+ BT.601 cast + squared error) alone: bursts of 12 launches, median of 7. Usage: t3_time.py [batch [h w]]"""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from autoencoder_based_image_compression_amd import device as dev, pipeline

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 24
(H, W) = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 768)
variables = bench.synthetic_model(1.)
dec = pipeline.DeviceDecoder(variables, False)
x = torch.randn((batch, H//4, W//4, 128), device='cuda')
ref = torch.randint(16, 236, (batch, H, W), dtype=torch.uint8, device='cuda')
sse = torch.zeros(batch, dtype=torch.int64, device='cuda')
ts = []
for rnd in range(8):
    (a, b) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    dev.tconv9x9s4_luma(x, dec.w6, want_f32=False, want_u8=True, ref_u8=ref, sse=sse)
    a.record()
    for _ in range(12):
        dev.tconv9x9s4_luma(x, dec.w6, want_f32=False, want_u8=True, ref_u8=ref, sse=sse)
    b.record()
    torch.cuda.synchronize()
    if rnd:
        ts.append(a.elapsed_time(b)/12)
t = statistics.median(ts)
print('tconv3 batch %d %dx%d: %.4f ms  (%.3f of 157.3 TF on 1,296 FLOP/px)' % (batch, H, W, t, 1296.*batch*H*W/(t*1e-3)/157.3e12))
