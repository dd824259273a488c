"""Per-wave phase timestamps of conv_gemm_split_kernel (SRC=conv_gemm_split EXTRA=-DEAE_TRACE SCRIPT=gemm_trace.py scratch/variant.sh [layer])."""
import os, sys, ctypes
import numpy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from autoencoder_based_image_compression_amd import device as dev, pipeline, _native
layer = sys.argv[1] if len(sys.argv) > 1 else 'conv2'
batch, H, W = 24, 512, 768
variables = bench.synthetic_model(1.)
enc = pipeline.DeviceEncoder(variables, False)
dec = pipeline.DeviceDecoder(variables, False)
(v, d) = (enc.v, dec.v)
images = torch.from_numpy(bench.synthetic_images(5, batch, H, W)).cuda()
gdn_1 = dev.conv9x9s4_u8(images, enc.w1, v['encoder/biases_1'], enc.g[1], v['encoder/beta_1'])
gdn_2 = dev.conv5x5s2(gdn_1, enc.w2, v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], v['encoder/beta_2'], workspace=False)
conv_3 = dev.conv5x5s2(gdn_2, enc.w3, v['encoder/biases_3'], dev.NORM_NONE, workspace=False)
t1 = dev.tconv5x5s2(conv_3, dec.w4, d['decoder/biases_4'], dev.NORM_IGDN, dec.g[5], d['decoder/beta_5'], workspace=False)
ws = dev.conv_workspace('cuda')
runs = {
    'conv2': lambda: dev.conv5x5s2(gdn_1, enc.w2, v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], v['encoder/beta_2'], out=gdn_2, workspace=ws),
    'conv3': lambda: dev.conv5x5s2(gdn_2, enc.w3, v['encoder/biases_3'], dev.NORM_NONE, out=conv_3, workspace=ws),
    'tconv2': lambda: dev.tconv5x5s2(t1, dec.w5, d['decoder/biases_5'], dev.NORM_IGDN, dec.g[6], d['decoder/beta_6'], workspace=ws),
}
run = runs[layer]
for _ in range(4):
    run()
torch.cuda.synchronize()
(a, b) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
a.record(); run(); b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b)
n = 65536
buf = numpy.zeros(n*8, dtype=numpy.int64)
lib = _native.hip()
lib.eae_hip_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.eae_hip_trace_read(buf.ctypes.data, buf.size) == 0
t = buf.reshape(n, 8)
live = t[:, 4] > 0
t = t[live]
steps = t[:, 5]
k = (t[:, 2] - t[:, 1])
full = steps == steps.max()
print('%s: launch %.1f us; %d waves with an item, %d whole tiles of %d K-steps' % (layer, ms*1e3, len(t), int(full.sum()), int(steps.max())))
print('  prologue (start -> K loop)   mean %8.0f cycles' % (t[:, 1] - t[:, 0]).mean())
print('  K loop per step              mean %8.1f  median %8.1f  p90 %8.1f   (64 MFMAs = 4096 cycles; x3 waves sharing = 12288)' % ((k[full]/steps[full]).mean(), numpy.median(k[full]/steps[full]), numpy.percentile(k[full]/steps[full], 90)))
print('  epilogue / hand-off          mean %8.0f cycles' % (t[:, 4] - t[:, 2]).mean())
span = t[:, 4].max() - t[:, 0].min()
print('  span %d ticks = %.1f MHz x launch; sum of wave lifetimes / span = %.1f waves resident' % (span, span/(ms*1e3), (t[:, 4] - t[:, 0]).sum()/span))
