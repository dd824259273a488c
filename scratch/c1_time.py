"""Times conv_1 + gdn_1 alone: bursts of 12 launches, median of 7. Usage: c1_time.py [batch [h w]]"""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from autoencoder_based_image_compression_amd import device as dev, pipeline
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 24
(H, W) = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 768)
variables = bench.synthetic_model(1.)
enc = pipeline.DeviceEncoder(variables, False)
v = enc.v
images = torch.from_numpy(bench.synthetic_images(5, batch, H, W)).cuda()
ts = []
for rnd in range(8):
    (a, b) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    dev.conv9x9s4_u8(images, enc.w1, v['encoder/biases_1'], enc.g[1], v['encoder/beta_1'])
    a.record()
    for _ in range(12):
        dev.conv9x9s4_u8(images, enc.w1, v['encoder/biases_1'], enc.g[1], v['encoder/beta_1'])
    b.record()
    torch.cuda.synchronize()
    if rnd:
        ts.append(a.elapsed_time(b)/12)
t = statistics.median(ts)
print('conv1 batch %d %dx%d: %.4f ms  (%.3f of 157.3 TF on %.0f FLOP/px)' % (batch, H, W, t, pipeline.FLOP_PER_PIXEL['conv1_gdn1']*batch*H*W/(t*1e-3)/157.3e12, pipeline.FLOP_PER_PIXEL['conv1_gdn1']))
ts = []
for rnd in range(8):
    (a, b) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    dev.conv9x9s4_u8(images, enc.w1, v['encoder/biases_1'], None, None)
    a.record()
    for _ in range(12):
        dev.conv9x9s4_u8(images, enc.w1, v['encoder/biases_1'], None, None)
    b.record()
    torch.cuda.synchronize()
    if rnd:
        ts.append(a.elapsed_time(b)/12)
t = statistics.median(ts)
print('conv1 without gdn_1: %.4f ms  (%.3f of 157.3 TF on 1296 FLOP/px; output write %.2f TB/s)' % (t, 1296.*batch*H*W/(t*1e-3)/157.3e12, batch*H*W/16*512/(t*1e-3)/1e12))
