"""Diagnostic: where does a conv GEMM block spend its cycles? (s_memtime stamps per wave)"""
import os, sys
import numpy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoencoder_based_image_compression_amd import _native, device as dev
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
v = var.random_variables(1., False, seed=0, bias_std=0.01)
cu = lambda a: torch.from_numpy(numpy.ascontiguousarray(a)).cuda()
N = int(os.environ.get('N', '24'))
for name in ('tconv2', 'conv2'):
    if name == 'tconv2':
        x = torch.randn(N, 64, 96, 128, device='cuda')
        w = dev.pack_tconv_weights(cu(v['decoder/weights_5'])); g = dev.pack_gamma(cu(v['decoder/gamma_6']))
        bb = cu(v['decoder/biases_5']); be = cu(v['decoder/beta_6'])
        fn = lambda: dev.tconv5x5s2(x, w, bb, 2, g, be)
        grid = N*48*4
    else:
        x = torch.randn(N, 128, 192, 128, device='cuda')
        w = dev.pack_conv_weights(cu(v['encoder/weights_2'])); g = dev.pack_gamma(cu(v['encoder/gamma_2']))
        bb = cu(v['encoder/biases_2']); be = cu(v['encoder/beta_2'])
        fn = lambda: dev.conv5x5s2(x, w, bb, 1, g, be)
        grid = N*48
    fn(); torch.cuda.synchronize()
    stamps = torch.zeros(grid*4*8, dtype=torch.int64, device='cuda')
    _native.hip().eae_hip_debug_set_stamp_buffer(stamps.data_ptr())
    fn(); torch.cuda.synchronize()
    _native.hip().eae_hip_debug_set_stamp_buffer(None)
    s = stamps.cpu().numpy().reshape(grid, 4, 8)
    t0 = s[..., 0].min()
    pro = s[..., 1] - s[..., 0]; loop = s[..., 2] - s[..., 1]; gdn = s[..., 3] - s[..., 2]; fin = s[..., 4] - s[..., 3]
    steps = s[..., 5]
    print(name, 'kernel span (s_memtime ticks)', s[..., 4].max() - t0)
    print('  per wave medians: prologue', numpy.median(pro), 'loop', numpy.median(loop), 'loop/step', numpy.median(loop/steps),
          'gdn', numpy.median(gdn), 'final', numpy.median(fin), 'total', numpy.median(s[..., 4] - s[..., 0]))
    for k in sorted(set(steps.reshape(-1))):
        m = steps == k
        print('   steps', k, 'n', m.sum(), 'loop/step', numpy.median((loop/steps)[m]), 'pro', numpy.median(pro[m]), 'gdn', numpy.median(gdn[m]), 'fin', numpy.median(fin[m]))
    starts = numpy.sort(s[..., 0].reshape(-1)); ends = numpy.sort(s[..., 4].reshape(-1))
    ts = numpy.linspace(t0, s[..., 4].max(), 24)
    alive = [int((starts <= t).sum() - (ends <= t).sum()) for t in ts]
    print('  waves alive over time:', alive)
