import os, sys, collections
import numpy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoencoder_based_image_compression_amd import _native, device as dev
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
v = var.random_variables(1., False, seed=0, bias_std=0.01)
cu = lambda a: torch.from_numpy(numpy.ascontiguousarray(a)).cuda()
N = int(os.environ.get('N', '24'))
x = torch.randn(N, 128, 192, 128, device="cuda")
w = dev.pack_conv_weights(cu(v["encoder/weights_2"])); g = dev.pack_gamma(cu(v["encoder/gamma_2"]))
bb = cu(v["encoder/biases_2"]); be = cu(v["encoder/beta_2"])
fn = lambda: dev.conv5x5s2(x, w, bb, 1, g, be)
grid = N*96 + 8
fn(); torch.cuda.synchronize()
stamps = torch.zeros(grid*4*8, dtype=torch.int64, device='cuda')
_native.hip().eae_hip_debug_set_stamp_buffer(stamps.data_ptr())
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
_native.hip().eae_hip_debug_set_stamp_buffer(None)
print('kernel ms', e0.elapsed_time(e1))
s = stamps.cpu().numpy().reshape(grid*2, 2, 8)
print('zero starts', int((s[..., 0] == 0).sum()), 'of', s[..., 0].size)
ok = s[..., 0] != 0
t0 = s[..., 0][ok].min(); t1 = s[..., 4][ok].max()
print('span ticks', t1 - t0, '-> tick rate GHz if span==kernel:', (t1 - t0)/(e0.elapsed_time(e1)*1e6))
starts = numpy.sort(s[..., 0][ok]); ends = numpy.sort(s[..., 4][ok])
ts = numpy.linspace(t0, t1, 30)
print('waves alive:', [int(numpy.searchsorted(starts, t, 'right') - numpy.searchsorted(ends, t, 'right')) for t in ts])
xcc = s[..., 6][ok]; hw = s[..., 7][ok]
cu_id = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
key = xcc*100000 + se*1000 + sh*100 + cu_id
cnt = numpy.bincount(numpy.unique(key, return_inverse=True)[1])
print('distinct CUs seen', len(cnt), 'waves per CU min/max', cnt.min(), cnt.max())
life = (s[..., 4] - s[..., 0])[ok]
per_cu = collections.defaultdict(float)
for k, l in zip(key.tolist(), life.tolist()):
    per_cu[k] += l
vals = numpy.array(list(per_cu.values()))/(t1 - t0)
print('avg resident waves per CU: mean', vals.mean(), 'min', vals.min(), 'max', vals.max())
# per-CU analysis (s_memtime is per-XCD, so stay within a CU)
groups = collections.defaultdict(list)
for k, a, b2, st in zip(key.tolist(), s[..., 0][ok].tolist(), s[..., 4][ok].tolist(), s[..., 5][ok].tolist()):
    groups[k].append((a, b2, st))
spans = []; occ = []; mf = []
for k, lst in groups.items():
    a0 = min(x[0] for x in lst); b0 = max(x[1] for x in lst)
    spans.append(b0 - a0)
    occ.append(sum(x[1] - x[0] for x in lst)/(b0 - a0))
    mf.append(sum((x[2]*64 + 256)*64 for x in lst)/4/(b0 - a0))   # MFMA cycles per SIMD / span
spans = numpy.array(spans); occ = numpy.array(occ); mf = numpy.array(mf)
print('per-CU span ticks: median', numpy.median(spans), 'min', spans.min(), 'max', spans.max(), '-> GHz vs kernel time', numpy.median(spans)/(e0.elapsed_time(e1)*1e6))
print('per-CU avg resident waves: median', numpy.median(occ), 'min', occ.min(), 'max', occ.max())
print('per-CU MFMA pipe utilisation: median', numpy.median(mf), 'min', mf.min(), 'max', mf.max())
one = sorted(groups[list(groups)[0]])
a0 = one[0][0]
print('one CU timeline (start, end, steps) in kiloticks:', [(round((x[0]-a0)/1e3), round((x[1]-a0)/1e3), int(x[2])) for x in one[:40]])
print('waves per CU: min', cnt.min(), 'max', cnt.max(), 'hist', numpy.bincount(cnt)[cnt.min():cnt.max()+1].tolist()[:40])
work = numpy.array([sum(x[2] for x in lst) for lst in groups.values()])
print('K-steps per CU: min', work.min(), 'max', work.max(), 'mean', work.mean())
xs = numpy.array([k//100000 for k in groups.keys()])
for xc in range(8):
    m = xs == xc
    print(' xcc', xc, 'CUs', m.sum(), 'steps/CU mean', work[m].mean(), 'span median', numpy.median(spans[m]), 'start min', min(min(x[0] for x in lst) for k, lst in groups.items() if k//100000 == xc) - 0)
