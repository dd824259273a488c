"""Times the four conv GEMM launches of the benchmark's step (24 x 512x768) in every form, interleaved in one process:
bursts of back-to-back launches of one layer (sustained duty cycle), HIP events around the burst,
median over rounds. Usage: python scratch/gemm_forms.py [batch [height width [form ...]]]"""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from autoencoder_based_image_compression_amd import _native, device as dev, pipeline

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 24
(H, W) = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 768)
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
outs = {}
layers = {
    'conv2': lambda w: dev.conv5x5s2(gdn_1, enc.w2, v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], v['encoder/beta_2'], out=outs.setdefault('conv2', torch.empty_like(gdn_2)), workspace=w),
    'conv3': lambda w: dev.conv5x5s2(gdn_2, enc.w3, v['encoder/biases_3'], dev.NORM_NONE, out=outs.setdefault('conv3', torch.empty_like(conv_3)), workspace=w),
    'tconv1': lambda w: dev.tconv5x5s2(conv_3, dec.w4, d['decoder/biases_4'], dev.NORM_IGDN, dec.g[5], d['decoder/beta_5'], out=outs.setdefault('tconv1', torch.empty_like(t1)), workspace=w),
    'tconv2': lambda w: dev.tconv5x5s2(t1, dec.w5, d['decoder/biases_5'], dev.NORM_IGDN, dec.g[6], d['decoder/beta_6'], out=outs.setdefault('tconv2', torch.empty((batch, H//4, W//4, 128), device='cuda')), workspace=w),
}
forms = {
    'wave': ({'EAE_HIP_GEMM': 'w'}, False),
    'wave128': ({'EAE_HIP_GEMM': 'w', 'EAE_HIP_FORCE_TILE': '128'}, False),
    'wave32_nt4': ({'EAE_HIP_GEMM': 'w', 'EAE_HIP_FORCE_TILE': '32', 'EAE_HIP_FORCE_NT': '4'}, False),
    'wave32_nt2': ({'EAE_HIP_GEMM': 'w', 'EAE_HIP_FORCE_TILE': '32', 'EAE_HIP_FORCE_NT': '2'}, False),
    'wave32_nt1': ({'EAE_HIP_GEMM': 'w', 'EAE_HIP_FORCE_TILE': '32', 'EAE_HIP_FORCE_NT': '1'}, False),
    'wave64_nt4': ({'EAE_HIP_GEMM': 'w', 'EAE_HIP_FORCE_TILE': '64', 'EAE_HIP_FORCE_NT': '4'}, False),
    'pack_nt2': ({'EAE_HIP_GEMM': 'w', 'EAE_HIP_FORCE_NT': '2', 'EAE_HIP_PACK': '1'}, False),
    'pack_nt1': ({'EAE_HIP_GEMM': 'w', 'EAE_HIP_FORCE_NT': '1', 'EAE_HIP_PACK': '1'}, False),
    'default': ({}, ws),
    'whole': ({'EAE_HIP_GEMM': 'u'}, ws),
    'cut1': ({'EAE_HIP_GEMM': 's', 'EAE_HIP_SPLIT_WAVES': '1'}, ws),
    'cut2': ({'EAE_HIP_GEMM': 's', 'EAE_HIP_SPLIT_WAVES': '2'}, ws),
    'cut3': ({'EAE_HIP_GEMM': 's', 'EAE_HIP_SPLIT_WAVES': '3'}, ws),
}
extra = [a for a in sys.argv[4:]]
if extra:
    forms = {k: forms[k] for k in extra}
KEYS = ('EAE_HIP_GEMM', 'EAE_HIP_FORCE_TILE', 'EAE_HIP_SPLIT_WAVES', 'EAE_HIP_FORCE_NT', 'EAE_HIP_PACK')
BURST, ROUNDS = 12, 7
flops = {'conv2': pipeline.FLOP_PER_PIXEL['conv2_gdn2'], 'conv3': 3200, 'tconv1': pipeline.FLOP_PER_PIXEL['tconv1_igdn5'], 'tconv2': pipeline.FLOP_PER_PIXEL['tconv2_igdn6']}
times = {}
for rnd in range(ROUNDS + 1):
    for (lname, fn) in layers.items():
        for (fname, (env, w)) in forms.items():
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            _native.hip().eae_hip_debug_reload_launch_options()      # the forms are read when the library loads (round 4 on)
            (a, b) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            fn(w)
            a.record()
            for _ in range(BURST):
                fn(w)
            b.record()
            torch.cuda.synchronize()
            if rnd:
                times.setdefault((lname, fname), []).append(a.elapsed_time(b)/BURST)
print('batch', batch, 'ms per launch (median of %d bursts of %d), fraction of 157.3 TF' % (ROUNDS, BURST))
for lname in layers:
    row = []
    for fname in forms:
        t = statistics.median(times[(lname, fname)])
        row.append('%s %.4f (%.3f)' % (fname, t, flops[lname]*batch*H*W/(t*1e-3)/157.3e12))
    print(lname.ljust(7), ' | '.join(row))
print('error word', int(ws[255].item()), 'workspace clean', int(torch.count_nonzero(ws).item()) == 0)
