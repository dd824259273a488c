"""The batch decoder alone on one stream while conv GEMM launches keep the GPU busy on another: does the verify still pass?
(The codec's failure isolated from BatchCodec.)  usage: r03_decode_under_load.py [rounds]"""
import os, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy, torch
import bench
import test_coder_device as T
from autoencoder_based_image_compression_amd import device as dev, pipeline
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bw = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
batch = 24
variables = bench.synthetic_model(bw)
images = torch.from_numpy(bench.synthetic_images(1000, batch, 512, 768)).cuda()
bin_widths = variables[var.BIN_WIDTHS_NAME]
enc = pipeline.DeviceEncoder(variables, False)
y = enc(images)
map_mean = dev.map_means(y)
probabilities = lossless_stats.compute_binary_probabilities(y.cpu().numpy(), bin_widths, map_mean.cpu().numpy(), 10)
q = dev.quantize_maps(y, torch.from_numpy(bin_widths).cuda(), map_mean, want_symbols=True)
symbols = q['symbols'].reshape(batch*128, -1).contiguous()
rows = torch.arange(128, dtype=torch.int32).repeat(batch)
rows[67::128] = -1
rows = rows.cuda()
prob = torch.from_numpy(probabilities).cuda()
streams = dev.CoderStreams(batch*128, symbols.shape[1], 10, 'cuda')
ws = dev.coder_workspace(batch*128, symbols.shape[1], 10, 'cuda')
dev.coder_encode_batch(symbols, prob, rows, 10, out=streams, workspace=ws)
torch.cuda.synchronize()
assert int(streams.status.abs().sum()) == 0
gdn_1 = dev.conv9x9s4_u8(images, enc.w1, enc.v['encoder/biases_1'], enc.g[1], enc.v['encoder/beta_1'])
side = torch.cuda.Stream()
out = torch.empty((batch, 64, 96, 128), device='cuda')
for load in (False, True):
    bad_rounds = 0
    detail = None
    for r in range(rounds):
        if load:
            for _ in range(6):
                dev.conv5x5s2(gdn_1, enc.w2, enc.v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], enc.v['encoder/beta_2'], out=out)
        with torch.cuda.stream(side):
            dev.coder_decode_batch(streams, prob, rows, expected=symbols, workspace=ws)
        torch.cuda.synchronize()
        if r == 0 and os.environ.get('EAE_DUMP'):
            n_maps = batch*128; size = symbols.shape[1]
            base = (-ws.data_ptr()) % 256
            dcap = (size*11 + 7)//8*8
            r256 = lambda v: (v + 255)//256*256
            a_off = base + r256(n_maps*4)
            b_off = a_off + r256(max((n_maps + 63)//64*64*dcap, n_maps*size*2))
            pref = ws[b_off:b_off + n_maps*size].cpu().numpy().reshape(n_maps, size)
            dec = ws[a_off:a_off + n_maps*size*2].view(torch.int16).cpu().numpy().reshape(n_maps, size)
            sym = symbols.cpu().numpy()
            want = numpy.minimum(numpy.abs(sym.astype(numpy.int32)), 10).astype(numpy.uint8)
            coded = rows.cpu().numpy() >= 0
            bad_pref = numpy.flatnonzero((pref != want).any(axis=1) & coded)
            bad_dec = numpy.flatnonzero((dec != sym).any(axis=1) & coded)
            print('   load', load, ': maps with wrong prefix bytes', bad_pref.size, 'maps with wrong decoded symbols', bad_dec.size,
                  'both', numpy.intersect1d(bad_pref, bad_dec).size)
            only_pref = numpy.setdiff1d(bad_pref, bad_dec)
            for m in only_pref[:4]:
                w = numpy.flatnonzero(pref[m] != want[m])
                print('     map', int(m), 'PREFIX wrong only: at', w[:8], 'count', w.size, 'got', pref[m][w[:8]], 'want', want[m][w[:8]], 'nonzero symbols', int((sym[m] != 0).sum()),
                      'bac bits', int(streams.bac_bits[m]))
            firsts = [int(numpy.flatnonzero(pref[m] != want[m])[0]) for m in bad_pref] or [-1]
            if bad_pref.size: print('     first wrong prefix index: min/median/max', min(firsts), int(numpy.median(firsts)), max(firsts), ' values seen in wrong bytes', numpy.unique(pref[bad_pref][pref[bad_pref] != want[bad_pref]])[:20])
            for m in bad_dec[:2]:
                print('     map', int(m), 'got ', ''.join('%x' % v for v in pref[m][:120]))
                print('     map', int(m), 'want', ''.join('%x' % v for v in want[m][:120]))
            for m in bad_dec[:3]:
                w = numpy.flatnonzero(dec[m] != sym[m])
                print('     map', int(m), 'decoded wrong at', w[:6], 'got', dec[m][w[:6]], 'want', sym[m][w[:6]], 'prefix there', pref[m][w[:6]], 'want', want[m][w[:6]],
                      'bypass bits', int(streams.bypass_bits[m]))
        if os.environ.get('EAE_LDSCHECK') and r == 0:
            sg = streams.stage.cpu().numpy()
            coded_ = rows.cpu().numpy() >= 0
            print('   load', load, ': lanes with a changed ring word', int(((sg & 0xff) != 0)[coded_].sum()), ' changed probability', int((((sg >> 8) & 0xff) != 0)[coded_].sum()),
                  ' symbols decoded != size', int(((sg >> 16) != symbols.shape[1])[coded_].sum()), 'examples', [hex(int(v)) for v in sg[coded_][:6]])
        st = streams.status.cpu().numpy()
        if st.any():
            bad_rounds += 1
            if detail is None:
                w = numpy.flatnonzero(st)
                detail = (r, w.size, sorted(set((w//64).tolist())), w[:8].tolist())
            streams.status.zero_()
    if os.environ.get('EAE_TRACE'):
        n_maps = batch*128; size = symbols.shape[1]
        base = (-ws.data_ptr()) % 256
        dcap = (size*11 + 7)//8*8
        r256 = lambda v: (v + 255)//256*256
        off = base + r256(n_maps*4) + r256(max((n_maps + 63)//64*64*dcap, n_maps*size*2)) + (n_maps*size + 3)//4*4
        tr = ws[off:off + 512*64*8*4].view(torch.int32).cpu().numpy().reshape(512, 64, 8).copy()
        if not load:
            ref = tr
        else:
            diff = (tr != ref)
            steps = numpy.flatnonzero(diff.any(axis=(1, 2)))
            print('trace: first differing step', steps[:5], 'of', steps.size)
            if steps.size:
                t = int(steps[0])
                lanes = numpy.flatnonzero(diff[t].any(axis=1))
                for l in lanes[:6]:
                    print('  step', t, 'lane', int(l), 'fields differing', numpy.flatnonzero(diff[t, l]).tolist(),
                          'ref', [hex(int(v) & 0xffffffff) for v in ref[t, l]], 'got', [hex(int(v) & 0xffffffff) for v in tr[t, l]])
                    print('     previous step ref', [hex(int(v) & 0xffffffff) for v in ref[t - 1, l]], 'got', [hex(int(v) & 0xffffffff) for v in tr[t - 1, l]])
    print('bin width', bw, 'max stream bits', int(streams.bac_bits.max()), 'GEMM load on the other stream:', load, '-> rounds with a failed verify:', bad_rounds, 'of', rounds, detail)
