"""r04: where a quarter-tile conv GEMM wave (one image: conv_gemm_wave_kernel<1, NONE, 1>) spends its time: s_memtime stamps per wave.
N=1 python scratch/r04/stamps_small.py"""
import os, sys
import numpy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from autoencoder_based_image_compression_amd import _native, device as dev
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
v = var.random_variables(1., False, seed=0, bias_std=0.01)
cu = lambda a: torch.from_numpy(numpy.ascontiguousarray(a)).cuda()
N = int(os.environ.get('N', '1'))
for (name, hh, ww) in (('conv3', 64, 96), ('conv2', 128, 192)):
    x = torch.randn(N, hh, ww, 128, device='cuda')
    key = 'encoder/weights_3' if name == 'conv3' else 'encoder/weights_2'
    w = dev.pack_conv_weights(cu(v[key]))
    bb = cu(v['encoder/biases_3' if name == 'conv3' else 'encoder/biases_2'])
    fn = lambda: dev.conv5x5s2(x, w, bb, 0, None, None)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    (a, e) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    a.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    print(name, 'N', N, 'back to back: %.1f us per launch' % (a.elapsed_time(e)/10*1e3))
    grid = 4096
    stamps = torch.zeros(grid*4*8, dtype=torch.int64, device='cuda')
    _native.hip().eae_hip_debug_set_stamp_buffer(stamps.data_ptr())
    fn(); torch.cuda.synchronize()
    _native.hip().eae_hip_debug_set_stamp_buffer(None)
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] != 0]
    t0 = s[:, 0].min()
    pro = s[:, 1] - s[:, 0]; loop = s[:, 2] - s[:, 1]; fin = s[:, 4] - s[:, 2]
    steps = numpy.maximum(s[:, 5], 1)
    print('  waves', len(s), 'kernel span (s_memtime ticks of 10 ns)', s[:, 4].max() - t0, ' start spread', s[:, 0].max() - t0)
    print('  per wave medians: prologue', numpy.median(pro), 'loop', numpy.median(loop), 'steps', numpy.median(steps), 'loop/step', numpy.median(loop/steps), 'epilogue', numpy.median(fin))
