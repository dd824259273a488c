"""r04: with the `a_dump` library (scratch/r04/edit_dump.py): the register state of every decoder wavefront at the end of its decode loop,
alone on the GPU and next to other kernels; which registers differ, and (for the loop invariants) what they hold instead.
   EAE_HIP_LIB=scratch/r04/libs/a_dump/libeae_hip.so python scratch/r04/dump_compare.py"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(HERE, 'decode_hunt.py')).read()
exec(compile(src[:src.index('NEIGHBOURS = [')], 'decode_hunt_setup', 'exec'))

def prefixes():
    base = (-ws.data_ptr()) % 256
    dcap = (size*(L + 1) + 7)//8*8
    r256 = lambda v: (v + 255)//256*256
    b_off = base + r256(n_maps*4) + r256(max((n_maps + 63)//64*64*dcap, n_maps*size*2))
    return ws[b_off:b_off + n_maps*size].cpu().numpy().reshape(n_maps, size)


INVARIANT_V = (0, 1, 2, 3, 4, 5, 6, 7, 9, 16, 17, 18, 22)
INVARIANT_S = tuple(range(34, 48)) + (12, 13)


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
    raw = streams.streams[:, 4096:4096 + 256 + 4*82].contiguous().cpu().numpy().view(numpy.uint32)      # [n_maps][64 + 82]
    return (raw[:, :40].copy(), raw[:, 64:64 + 82].copy(), prefixes() != want)


(v_ref, s_ref, wrong_ref) = run(None)
(v_ref2, s_ref2, _) = run(None)
print('alone twice: maps with a wrong prefix', int((wrong_ref.any(axis=1) & coded).sum()), '; VGPR words differing between the two runs', int((v_ref != v_ref2)[:, [k for k in range(40) if k != 8]].sum()),
      '; SGPR words differing (s0..s59)', int((s_ref != s_ref2)[:, :60].sum()))
for (name, beside) in (('VALU only', synthetic(1, 2048, 8000, 3)), ('MFMA only', synthetic(0, 768, 8000, 3))):
    (v_got, s_got, wrong) = run(beside)
    bad_maps = wrong.any(axis=1) & coded
    print('=== next to', name, ': maps with a wrong prefix', int(bad_maps.sum()))
    for w in range(0, n_maps//64):
        lanes = slice(w*64, w*64 + 64)
        nbad = int(bad_maps[lanes].sum())
        dv = [(k, int((v_got[lanes, k] != v_ref[lanes, k]).sum())) for k in range(40) if k != 8]
        ds = [j for j in range(60) if s_got[w*64, j] != s_ref[w*64, j]]
        inv_v = [(k, n) for (k, n) in dv if k in INVARIANT_V and n]
        inv_s = [j for j in ds if j in INVARIANT_S]
        hw = int(s_got[w*64, 66]); gpr = int(s_got[w*64, 67]); lds = int(s_got[w*64, 68])
        t0 = int(s_got[w*64, 70]) | (int(s_got[w*64, 71]) << 32); t1 = int(s_got[w*64, 72]) | (int(s_got[w*64, 73]) << 32)
        line = 'wave %2d se %d cu %2d simd %d slot %d gpr_alloc 0x%08x lds_alloc 0x%08x ticks %9d..%9d | wrong maps %2d | invariant VGPRs changed %s | invariant SGPRs changed %s' % (
            w, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15, gpr, lds, t0 % 10**9, t1 % 10**9, nbad, inv_v, inv_s)
        print(line)
        if inv_v and w < 48:
            (k, _) = inv_v[0]
            lanes_changed = numpy.flatnonzero(v_got[lanes, k] != v_ref[lanes, k])[:6]
            print('      e.g. v%d: lanes %s hold %s instead of %s' % (k, lanes_changed.tolist(), [hex(int(x)) for x in v_got[lanes, k][lanes_changed]], [hex(int(x)) for x in v_ref[lanes, k][lanes_changed]]))
        if inv_s:
            print('      SGPRs: ' + ', '.join('s%d = 0x%08x (alone 0x%08x)' % (j, int(s_got[w*64, j]), int(s_ref[w*64, j])) for j in inv_s[:8]))
    # a census over all waves: how often each register differs in waves with wrong maps
    census = {}
    for w in range(n_maps//64):
        lanes = slice(w*64, w*64 + 64)
        if not bad_maps[lanes].any():
            continue
        for k in range(40):
            if k != 8 and (v_got[lanes, k] != v_ref[lanes, k]).any():
                census['v%d' % k] = census.get('v%d' % k, 0) + 1
        for j in range(60):
            if s_got[w*64, j] != s_ref[w*64, j]:
                census['s%d' % j] = census.get('s%d' % j, 0) + 1
    print('census (failing waves in which the register differs from the run alone):', sorted(census.items(), key=lambda kv: -kv[1]))
