"""r05: where conv_3 of 64 x 256x256 (1,024 half-tile waves on 1,024 SIMDs: 0.10 ms of MFMAs each) spends its 0.22 ms in the step:
per-wave timestamps (eae_hip_debug_set_stamp_buffer) of the launch right behind conv_2, and in a burst of its own."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy, torch
import bench
from autoencoder_based_image_compression_amd import _native, device as dev, pipeline
(N, H, W) = (int(os.environ.get('N', '64')), int(os.environ.get('H', '256')), int(os.environ.get('W', '256')))
variables = bench.synthetic_model(1.)
enc = pipeline.DeviceEncoder(variables, False)
v = enc.v
images = torch.from_numpy(bench.synthetic_images(5, N, H, W)).cuda()
gdn_1 = dev.conv9x9s4_u8(images, enc.w1, v['encoder/biases_1'], enc.g[1], v['encoder/beta_1'])
ws = dev.conv_workspace('cuda')
gdn_2 = dev.conv5x5s2(gdn_1, enc.w2, v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], v['encoder/beta_2'], workspace=ws)
out3 = dev.conv5x5s2(gdn_2, enc.w3, v['encoder/biases_3'], dev.NORM_NONE, workspace=ws)
conv2 = lambda: dev.conv5x5s2(gdn_1, enc.w2, v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], v['encoder/beta_2'], out=gdn_2, workspace=ws)
conv3 = lambda: dev.conv5x5s2(gdn_2, enc.w3, v['encoder/biases_3'], dev.NORM_NONE, out=out3, workspace=ws)
grid = 4096
stamps = torch.zeros(grid*8, dtype=torch.int64, device='cuda')
hip = _native.hip()


def report(tag, before):
    stamps.zero_()
    for _ in range(3):
        before(); conv3()
    torch.cuda.synchronize()
    before()
    hip.eae_hip_debug_set_stamp_buffer(stamps.data_ptr())
    (e0, e1) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    e0.record(); conv3(); e1.record(); torch.cuda.synchronize()
    hip.eae_hip_debug_set_stamp_buffer(None)
    ms = e0.elapsed_time(e1)
    s = stamps.cpu().numpy().reshape(grid, 8)
    s = s[s[:, 0] != 0]
    (t0, t1) = (s[:, 0].min(), s[:, 4].max())
    print('%s: conv_3 %.4f ms by events; %d waves stamped; span %d ticks (100 MHz: %.4f ms)' % (tag, ms, len(s), t1 - t0, (t1 - t0)/1e5))
    life = s[:, 4] - s[:, 0]; loop = s[:, 2] - s[:, 1]; pro = s[:, 1] - s[:, 0]
    print('   wave life ticks: median %d min %d max %d; prologue median %d; K loop median %d (%d steps)' % (numpy.median(life), life.min(), life.max(), numpy.median(pro), numpy.median(loop), s[0, 5]))
    starts = numpy.sort(s[:, 0]); ends = numpy.sort(s[:, 4])
    ts = numpy.linspace(t0, t1, 24)
    print('   waves alive over the span:', [int(numpy.searchsorted(starts, t, 'right') - numpy.searchsorted(ends, t, 'right')) for t in ts])
    print('   starts (ticks after the first): deciles', [int(x) for x in numpy.percentile(s[:, 0] - t0, [0, 10, 25, 50, 75, 90, 100])])
    (xcc, hw) = (s[:, 6], s[:, 7])
    simd = hw >> 4 & 0x3; cu = hw >> 8 & 0xF; sh = hw >> 12 & 1; se = hw >> 13 & 0x7
    key_cu = xcc*10000 + se*1000 + sh*100 + cu
    key_simd = key_cu*10 + simd
    per_cu = numpy.bincount(numpy.unique(key_cu, return_inverse=True)[1])
    per_simd = numpy.bincount(numpy.unique(key_simd, return_inverse=True)[1])
    print('   distinct CUs %d, waves per CU histogram %s; distinct SIMDs %d, waves per SIMD histogram %s' % (
        len(per_cu), numpy.bincount(per_cu).tolist(), len(per_simd), numpy.bincount(per_simd).tolist()))
    # do the waves that share a SIMD overlap in time?
    groups = collections.defaultdict(list)
    for (k, a, b) in zip(key_simd.tolist(), s[:, 0].tolist(), s[:, 4].tolist()):
        groups[k].append((a, b))
    overlap = sum(1 for lst in groups.values() if len(lst) > 1 and max(x[0] for x in lst) < min(x[1] for x in lst))
    print('   SIMDs whose waves overlap in time: %d of %d with more than one wave' % (overlap, sum(1 for lst in groups.values() if len(lst) > 1)))


report('behind conv_2', conv2)
report('behind an idle gap', lambda: torch.cuda.synchronize())
report('behind conv_3 itself (burst)', conv3)

# the same launch in a sustained sequence of the whole analysis + synthesis chain (what the one-stream leg of bench.py times)
dec = pipeline.DeviceDecoder(variables, False)
d = dec.v
t1 = dev.tconv5x5s2(out3, dec.w4, d['decoder/biases_4'], dev.NORM_IGDN, dec.g[5], d['decoder/beta_5'], workspace=ws)
t2 = dev.tconv5x5s2(t1, dec.w5, d['decoder/biases_5'], dev.NORM_IGDN, dec.g[6], d['decoder/beta_6'], workspace=ws)


def chain():
    dev.conv9x9s4_u8(images, enc.w1, v['encoder/biases_1'], enc.g[1], v['encoder/beta_1'], out=gdn_1)
    conv2(); conv3()
    dev.tconv5x5s2(out3, dec.w4, d['decoder/biases_4'], dev.NORM_IGDN, dec.g[5], d['decoder/beta_5'], out=t1, workspace=ws)
    dev.tconv5x5s2(t1, dec.w5, d['decoder/biases_5'], dev.NORM_IGDN, dec.g[6], d['decoder/beta_6'], out=t2, workspace=ws)


def sustained():
    for _ in range(60):
        chain()
    dev.conv9x9s4_u8(images, enc.w1, v['encoder/biases_1'], enc.g[1], v['encoder/beta_1'], out=gdn_1)
    conv2()


report('in a sustained sequence of the transforms (60 chains, conv_1, conv_2 in front)', sustained)
