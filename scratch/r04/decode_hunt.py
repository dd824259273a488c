"""r04: the decoder-core fault of DESIGN.md section 5, characterised. The batch decoder (verify mode) on one stream, static inputs
encoded once without any load, next to ONE kind of neighbour at a time on another stream (or behind a kernel that leaves LDS /
registers dirty on the same stream):   EAE_HIP_LIB=scratch/r04/libs/<variant>/libeae_hip.so python scratch/r04/decode_hunt.py [bw] [rounds]
Prints one line per (neighbour, LDS size) with the failing rounds, the failing wavefronts and where in its chain a map first went wrong.
The "LDS 65536 / 163840" rows and EAE_HUNT_PLACEMENT need the probe code that lived in coder_simd.hip during the hunt (-DEAE_DECODE_HUNT,
-DEAE_DECODE_HUNT_LDS): `git apply scratch/r04/edits/decode_hunt_probes.patch` puts it back."""
import ctypes, os, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy, torch
import bench
from autoencoder_based_image_compression_amd import device as dev, pipeline
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats

bw = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
only = sys.argv[3].split(',') if len(sys.argv) > 3 else None
lk = ctypes.CDLL(os.path.join(ROOT, 'scratch', 'r04', 'libs', 'liblk.so'))
lk.lk_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
batch, L = 24, 10
variables = bench.synthetic_model(bw)
images = torch.from_numpy(bench.synthetic_images(1000, batch, 512, 768)).cuda()
bin_widths = variables[var.BIN_WIDTHS_NAME]
enc = pipeline.DeviceEncoder(variables, False)
y = enc(images)
map_mean = dev.map_means(y)
probabilities = lossless_stats.compute_binary_probabilities(y.cpu().numpy(), bin_widths, map_mean.cpu().numpy(), L)
q = dev.quantize_maps(y, torch.from_numpy(bin_widths).cuda(), map_mean, want_symbols=True)
symbols = q['symbols'].reshape(batch*128, -1).contiguous()
n_maps, size = symbols.shape
rows = torch.arange(128, dtype=torch.int32).repeat(batch)
rows[67::128] = -1
coded = (rows >= 0).numpy()
rows = rows.cuda()
prob = torch.from_numpy(probabilities).cuda()
streams = dev.CoderStreams(n_maps, size, L, 'cuda')
ws = dev.coder_workspace(n_maps, size, L, 'cuda')
dev.coder_encode_batch(symbols, prob, rows, L, out=streams, workspace=ws)
torch.cuda.synchronize()
assert int(streams.status.abs().sum()) == 0
sym = symbols.cpu().numpy()
want = numpy.minimum(numpy.abs(sym.astype(numpy.int32)), L).astype(numpy.uint8)
decisions_before = numpy.cumsum(want.astype(numpy.int64) + (want < L), axis=1) - (want.astype(numpy.int64) + (want < L))
bac_bits = streams.bac_bits.cpu().numpy()
gdn_1 = dev.conv9x9s4_u8(images, enc.w1, enc.v['encoder/biases_1'], enc.g[1], enc.v['encoder/beta_1'])
out = torch.empty((batch, 64, 96, 128), device='cuda')
big = torch.zeros(1 << 28, dtype=torch.float32, device='cuda')       # 1 GB for the streaming copy
side = torch.cuda.Stream()
torch.cuda.synchronize()
print('variant', os.environ.get('EAE_HIP_LIB', 'shipped'), 'bin width', bw, 'max stream bits', int(bac_bits.max()), 'median', int(numpy.median(bac_bits[coded])))


def cur():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def gemm():
    for _ in range(6):
        dev.conv5x5s2(gdn_1, enc.w2, enc.v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], enc.v['encoder/beta_2'], out=out)


def synthetic(kind, blocks, iters, times):
    def go():
        for _ in range(times):
            lk.lk_launch(kind, blocks, iters, ctypes.c_void_p(big.data_ptr()), big.numel()*4, cur())
    return go


NEIGHBOURS = [
    ('none', None, None),
    ('conv GEMM', gemm, None),
    ('MFMA only', synthetic(0, 768, 8000, 3), None),
    ('VALU only', synthetic(1, 2048, 8000, 3), None),
    ('LDS only', synthetic(2, 1024, 3000, 4), None),
    ('memory copy', synthetic(3, 2048, 0, 12), None),
    ('after LDS dirtied (same stream)', None, synthetic(4, 1024, 0, 1)),
    ('after VGPRs dirtied (same stream)', None, synthetic(5, 4096, 0, 1)),
    ('conv GEMM, LDS 65536', gemm, None),
    ('conv GEMM, LDS 163840', gemm, None),
    ('VALU only, LDS 163840', synthetic(1, 2048, 8000, 3), None),
    ('MFMA only, LDS 163840', synthetic(0, 768, 8000, 3), None),
]


def placement(wrong):
    """Where and when every decoder wavefront ran (the hunt builds leave HW_ID / LDS_ALLOC / GPR_ALLOC / s_memtime in the stage words
    of a wavefront's first lanes), next to how it fared."""
    sg = streams.stage.cpu().numpy().astype(numpy.int64) & 0xFFFFFFFF
    rows_ = []
    for w in range((n_maps + 63)//64):
        g = sg[w*64:w*64 + 9]
        if g.size < 9:
            continue
        (hw0, hw1, lds, gpr, xcc) = (int(v) for v in g[:5])
        t0 = int(g[5]) | (int(g[6]) << 32)
        t1 = int(g[7]) | (int(g[8]) << 32)
        lanes = numpy.arange(w*64, min(n_maps, w*64 + 64))
        bad = [int(m) for m in lanes if coded[m] and wrong[m].any()]
        steps = sorted(int(decisions_before[m, numpy.flatnonzero(wrong[m])[0]]) for m in bad)
        rows_.append(dict(w=w, xcc=xcc & 15, se=(hw0 >> 13) & 7, cu=(hw0 >> 8) & 15, simd=(hw0 >> 4) & 3, slot=hw0 & 15, moved=int((hw0 ^ hw1) & 0xFFFF != 0),
                          lds_base=lds & 0xFF, lds_size=(lds >> 12) & 0x1FF, vgpr_base=gpr & 0x3F, vgpr_size=(gpr >> 8) & 0x3F, t0=t0, t1=t1, bad=len(bad), steps=steps))
    tmin = min(r['t0'] for r in rows_)
    rows_.sort(key=lambda r: (r['xcc'], r['se'], r['cu'], r['t0']))
    print('     wave xcc se cu simd slot | lds base size | vgpr base size | start end (x100 ticks) | failing lanes | first wrong decision of the failing lanes (sorted, first 10)')
    for r in rows_:
        print('     %4d %3d %2d %2d %4d %4d | %8d %4d | %9d %4d | %7d %7d | %3d | %s%s' % (
            r['w'], r['xcc'], r['se'], r['cu'], r['simd'], r['slot'], r['lds_base'], r['lds_size'], r['vgpr_base'], r['vgpr_size'],
            (r['t0'] - tmin)//100, (r['t1'] - tmin)//100, r['bad'], r['steps'][:10], ' MOVED' if r['moved'] else ''))



def prefixes():
    base = (-ws.data_ptr()) % 256
    dcap = (size*(L + 1) + 7)//8*8
    r256 = lambda v: (v + 255)//256*256
    b_off = base + r256(n_maps*4) + r256(max((n_maps + 63)//64*64*dcap, n_maps*size*2))
    return ws[b_off:b_off + n_maps*size].cpu().numpy().reshape(n_maps, size)


for (name, beside, before) in NEIGHBOURS:
    if only and not any(o in name for o in only):
        continue
    os.environ.pop('EAE_HUNT_DECODE_LDS', None)
    if 'LDS 65536' in name:
        os.environ['EAE_HUNT_DECODE_LDS'] = '65536'
    if 'LDS 163840' in name:
        os.environ['EAE_HUNT_DECODE_LDS'] = '163840'
    bad_rounds, detail = 0, None
    if os.environ.get('EAE_HUNT_SELF'):
        # kernels edited by hand (scratch/r04/edit_ops.py) no longer decode: their prefix bytes are compared with their OWN output alone on the GPU
        differing = []
        for r in range(rounds):
            ws.zero_()
            streams.status.zero_()
            torch.cuda.synchronize()
            if beside:
                beside()
            with torch.cuda.stream(side):
                if before:
                    before()
                dev.coder_decode_batch(streams, prob, rows, expected=symbols, workspace=ws)
            if beside:
                beside()
            torch.cuda.synchronize()
            got = prefixes()
            if name == 'none':
                if r == 0:
                    self_reference = got.copy()
                differing.append(int(((got != self_reference).any(axis=1) & coded).sum()))
            else:
                d = (got != self_reference).any(axis=1) & coded
                differing.append(int(d.sum()))
                if d.any() and detail is None:
                    detail = 'wavefronts with a differing map: %d of %d' % (len(set((numpy.flatnonzero(d)//64).tolist())), (n_maps + 63)//64)
        print('%-36s maps whose prefix bytes differ from the run alone, per round: %s   %s' % (name, differing, detail or ''), flush=True)
        continue
    for r in range(rounds):
        streams.status.zero_()
        torch.cuda.synchronize()
        if beside:
            beside()
        with torch.cuda.stream(side):
            if before:
                before()
            dev.coder_decode_batch(streams, prob, rows, expected=symbols, workspace=ws)
        if beside:
            beside()
        torch.cuda.synchronize()
        st = streams.status.cpu().numpy()
        if st.any():
            bad_rounds += 1
            if detail is None:
                pref = prefixes()
                wrong = (pref != want)
                bad = numpy.flatnonzero(wrong.any(axis=1) & coded)
                waves = sorted(set((bad//64).tolist()))
                firsts = numpy.array([int(numpy.flatnonzero(wrong[m])[0]) for m in bad])
                dec_at = numpy.array([int(decisions_before[m, f]) for (m, f) in zip(bad, firsts)])
                per_wave = {w: int(((bad//64) == w).sum()) for w in waves[:6]}
                coded_per_wave = {w: int(coded[w*64:(w + 1)*64].sum()) for w in waves[:6]}
                long_per_wave = {w: int((bac_bits[w*64:(w + 1)*64] > 256).sum()) for w in waves[:6]}
                detail = ('round %d: %d maps in %d wavefronts %s; failing lanes per wavefront %s (coded %s, streams > 256 bits %s); '
                          'first wrong symbol min/med/max %d/%d/%d, decisions before it %d/%d/%d (checkpoint %d/%d/%d), stream bits of the failing maps %d..%d, '
                          'status codes %s'
                          % (r, bad.size, len(waves), waves[:12], per_wave, coded_per_wave, long_per_wave, firsts.min(), numpy.median(firsts), firsts.max(),
                             dec_at.min(), numpy.median(dec_at), dec_at.max(), dec_at.min()//8, numpy.median(dec_at)//8, dec_at.max()//8,
                             bac_bits[bad].min(), bac_bits[bad].max(), numpy.unique(st[st != 0]).tolist()) if bad.size else 'status only: %s' % numpy.unique(st[st != 0]).tolist())
    print('%-36s failed verifies: %d of %d   %s' % (name, bad_rounds, rounds, detail or ''), flush=True)
    if os.environ.get('EAE_HUNT_PLACEMENT') and (name in ('none', 'memory copy') or bad_rounds):
        placement(prefixes() != want)
