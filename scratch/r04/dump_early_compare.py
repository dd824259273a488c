"""r04: with `a_dump_early` (scratch/r04/edit_dump_early.py): the state every decoder wavefront enters its loop with, alone and next to
other kernels -- it must be the same.   EAE_HIP_LIB=scratch/r04/libs/a_dump_early/libeae_hip.so python scratch/r04/dump_early_compare.py"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(HERE, 'decode_hunt.py')).read()
exec(compile(src[:src.index('NEIGHBOURS = [')], 'decode_hunt_setup', 'exec'))


def run(beside):
    streams.status.zero_()
    torch.cuda.synchronize()
    if beside:
        beside()
    with torch.cuda.stream(side):
        dev.coder_decode_batch(streams, prob, rows, expected=symbols, workspace=ws)
    if beside:
        beside()
    torch.cuda.synchronize()
    raw = streams.streams[:, 4096:4096 + 1024 + 128 + 88].contiguous().cpu().numpy().view(numpy.uint32)
    return dict(v=raw[:, :40].copy(), s=raw[:, 64:64 + 82].copy(), ring=raw[:, 256:288].copy(), probs=raw[:, 288:310].copy())


live = coded & (numpy.arange(n_maps) >= 0)
ref = run(None)
again = run(None)
names = {'v': ['v%d' % k for k in range(40)], 's': ['s%d' % j for j in range(82)], 'ring': ['ring[%d]' % r for r in range(32)], 'probs': ['p[%d].%s' % (k//2, 'lo' if k % 2 == 0 else 'hi') for k in range(22)]}
# registers that differ between two runs ALONE are not part of the deterministic state (dead / uninitialised at that point)
noisy = {key: (ref[key] != again[key])[live].any(axis=0) for key in ref}
print('not deterministic alone (ignored):', {key: [names[key][i] for i in numpy.flatnonzero(noisy[key])] for key in ref})
for (name, beside) in (('VALU only', synthetic(1, 2048, 8000, 3)), ('MFMA only', synthetic(0, 768, 8000, 3))):
    got = run(beside)
    print('=== next to', name)
    total = {}
    for w in range(n_maps//64):
        lanes = numpy.arange(w*64, w*64 + 64)
        lanes = lanes[live[lanes]]
        found = []
        for key in ('v', 's', 'ring', 'probs'):
            d = (got[key][lanes] != ref[key][lanes]) & ~noisy[key][None, :]
            if key == 's':
                d[:, 66:74] = False     # placement and clock
            for i in numpy.flatnonzero(d.any(axis=0)):
                l = lanes[numpy.flatnonzero(d[:, i])]
                found.append('%s (%d lanes, e.g. lane %d: 0x%08x instead of 0x%08x)' % (names[key][i], l.size, l[0] % 64, int(got[key][l[0], i]), int(ref[key][l[0], i])))
                total[names[key][i]] = total.get(names[key][i], 0) + 1
        hw = int(got['s'][w*64, 66]); gpr = int(got['s'][w*64, 67])
        print('wave %2d se %d cu %2d simd %d slot %d vgpr base %3d | %s' % (w, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15, (gpr & 0x3F)*8, '; '.join(found[:6]) or 'same as alone'))
    print('waves in which it differs:', sorted(total.items(), key=lambda kv: -kv[1]))
