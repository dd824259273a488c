"""Phase breakdown of tconv3_kernel from its tracing build (scratch/t3_trace.sh): cycles per wave-tile spent before each mark."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from autoencoder_based_image_compression_amd import device as dev, pipeline

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 24
(H, W) = (512, 768)
variables = bench.synthetic_model(1.)
dec = pipeline.DeviceDecoder(variables, False)
x = torch.randn((batch, H//4, W//4, 128), device='cuda')
ref = torch.randint(16, 236, (batch, H, W), dtype=torch.uint8, device='cuda')
sse = torch.zeros(128, dtype=torch.int64, device='cuda')
for _ in range(3):
    dev.tconv9x9s4_luma(x, dec.w6, want_f32=False, want_u8=True, ref_u8=ref, sse=sse)
torch.cuda.synchronize()
sse.zero_()
(a, b) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
a.record()
dev.tconv9x9s4_luma(x, dec.w6, want_f32=False, want_u8=True, ref_u8=ref, sse=sse)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b)
acc = sse[64:72].cpu().numpy().astype('float64')
wave_tiles = batch*(H//16)*(W//64)*2
names = ['barrier A wait', 'LDS write', 'barrier B wait', 'fetch issue', 'MFMA phase', 'barrier C wait', 'epilogue', 'loop top']
print('launch %.4f ms; clock ticks per wave-tile (sum over both passes), total %.0f' % (ms, acc.sum()/wave_tiles))
for (n, v) in zip(names, acc):
    print('  %-16s %9.1f  (%.1f %%)' % (n, v/wave_tiles, 100.*v/acc.sum()))
waves = min(1024, wave_tiles//2)*2
print('ticks per wave over the launch: %.0f -> %.1f MHz counter' % (acc.sum()/waves, acc.sum()/waves/(ms*1e3)))
