"""Where a block of transpose_conv_3 spends its shader-clock ticks (tracing build: SRC=tconv3 EXTRA=-DEAE_T3_TRACE SCRIPT=r06/t3_trace.py
bash scratch/variant.sh [N H W]). Wave 0 of every block adds its sums behind the per-image squared errors."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from autoencoder_based_image_compression_amd import device as dev

(n, H, W) = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (24, 512, 768)
x = torch.randn((n, H//4, W//4, 128), device='cuda') + 1.5
w6 = torch.rand((9, 9, 1, 128), device='cuda')*0.1
wph = dev.pack_tconv9x9s4_weights(w6)
ref = torch.randint(16, 236, (n, H, W), dtype=torch.uint8, device='cuda')
sse = torch.zeros(max(128, n + 16), dtype=torch.int64, device='cuda')
assert n <= 64
for _ in range(3):
    dev.tconv9x9s4_luma(x, wph, want_f32=False, want_u8=True, ref_u8=ref, sse=sse)
torch.cuda.synchronize()
sse.zero_()
(a, b) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
a.record()
dev.tconv9x9s4_luma(x, wph, want_f32=False, want_u8=True, ref_u8=ref, sse=sse)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b)
acc = sse[64:98].cpu().numpy().astype('float64')
rows = n*H//4
blocks = min(256, rows)
chunks = acc[33]
(a0, a1) = (acc[0:8]/4, acc[16:24]/4)                    # per wave of each half
print('launch %.4f ms; %d blocks, %.1f chunks per block; longest wave %.0f ticks -> %.0f MHz' % (ms, blocks, chunks/blocks, acc[32], acc[32]/(ms*1e3)))
print('  setup (filter, LDS zero, first sites): %.0f / %.0f ticks per block' % (a0[0]/blocks, a1[0]/blocks))
print('  waves 0-3: barrier %.0f, col2im etc. %.0f, MFMAs + parts + next sites %.0f ticks per chunk' % (a0[1]/chunks, a0[2]/chunks, a0[3]/chunks))
print('  waves 4-7: barrier %.0f, MFMAs + parts + next sites %.0f ticks per chunk' % (a1[1]/chunks, a1[4]/chunks))
print('  per chunk %.0f / %.0f; a chunk of a body row is 192 MFMAs = 6144 ticks on its SIMD' % (a0[1:4].sum()/chunks, (a1[1] + a1[4])/chunks))
