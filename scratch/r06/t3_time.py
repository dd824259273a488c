"""transpose_conv_3 alone on the GPU: ms per launch in bursts, fraction of the f32 MFMA peak (1,296 FLOP per output pixel).
usage: python scratch/r06/t3_time.py [N H W] (site rows / columns = H/4, W/4)"""
import os
import sys
import numpy
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from autoencoder_based_image_compression_amd import device as dev

(n, H, W) = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (24, 512, 768)
(h, w) = (H//4, W//4)
torch.manual_seed(0)
x = torch.randn((n, h, w, 128), device='cuda', dtype=torch.float32) + 1.5
w6 = (torch.rand((9, 9, 1, 128), device='cuda', dtype=torch.float32)*0.1).contiguous()
wph = dev.pack_tconv9x9s4_weights(w6)
ref = torch.randint(16, 236, (n, H, W), device='cuda', dtype=torch.uint8)
sse = torch.zeros(n, dtype=torch.int64, device='cuda')
out = None
for rep in range(3):
    ts = []
    for burst in range(5):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            out = dev.tconv9x9s4_luma(x, wph, want_f32=False, want_u8=True, ref_u8=ref, sse=sse)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)/20)
    ms = float(numpy.median(ts))
    flop = 1296.*n*H*W
    print('tconv3 {0}x{1}x{2}: {3:.4f} ms per launch (bursts of 20: {4}), {5:.1f} TFLOP/s = {6:.3f} of the f32 MFMA peak, strips env {7}'.format(
        n, H, W, ms, ' '.join('%.4f' % t for t in ts), flop/ms/1e9, flop/ms/1e9/157.3, os.environ.get('EAE_HIP_T3_STRIPS')), flush=True)
