"""On a failed batch of BatchCodec: are the STREAMS wrong (encode side) or only the decode? Compares the bad maps' streams with the
host library's and decodes the device's streams on the host."""
import os, sys, ctypes
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy, torch
import bench
import test_coder_device as T
from autoencoder_based_image_compression_amd import codec, device as dev, pipeline, _native
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats

batch = 24
variables = bench.synthetic_model(1.0)
images = torch.from_numpy(bench.synthetic_images(1000, batch, 512, 768)).cuda()
bin_widths = variables[var.BIN_WIDTHS_NAME]
y = pipeline.DeviceEncoder(variables, False)(images)
map_mean = dev.map_means(y).cpu().numpy()
probabilities = lossless_stats.compute_binary_probabilities(y.cpu().numpy(), bin_widths, map_mean, 10)
c = codec.BatchCodec(variables, False, bin_widths, map_mean, probabilities, 67, batch, 512, 768, nb_in_flight=1)
rows = numpy.tile(numpy.arange(128, dtype=numpy.int32), batch)
rows[67::128] = -1
found = False
for step in range(60):
    first_slot = c._index % c.nb_slots
    tickets = [c.submit(images) for _ in range(c.nb_slots)]
    failed = None
    for (k, t) in enumerate(tickets):
        try:
            t.result()
        except AssertionError:
            if failed is None:
                failed = k
    if failed is None:
        continue
    slot = (first_slot + failed) % c.nb_slots
    found = True
    torch.cuda.synchronize()
    res = c._views(c._pinned_out[slot])[0].numpy()
    bad = numpy.flatnonzero(res[2])
    print('step', step, 'slot', slot, 'bad maps', bad[:10], '... count', bad.size, 'block(s)', sorted(set((bad//64).tolist())))
    streams = c._coder_streams[slot]
    planar = c._symbols[slot].view(batch*128, -1).cpu().numpy()
    (h_streams, h_bac, h_byp, h_status, h_stage) = T.host_encode_maps(planar, probabilities, rows)[:5] if False else (None,)*5
    raw = streams.streams.cpu().numpy()
    half = streams.stride//2
    lib = _native.coder()
    for m in bad[:6]:
        m = int(m)
        pp = numpy.ascontiguousarray(probabilities[rows[m]])
        size = planar.shape[1]
        cap = size*32//8 + 32
        (bac, byp) = (numpy.zeros(cap, dtype=numpy.uint8), numpy.zeros(cap, dtype=numpy.uint8))
        (bb, yb, stage) = (ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_int(0))
        sym = numpy.ascontiguousarray(planar[m])
        rc = lib.eae_coder_encode(size, _native.ptr(sym, _native.c_i16p), 10, _native.ptr(pp, _native.c_f64p), _native.ptr(bac, _native.c_u8p),
                                  ctypes.byref(bb), _native.ptr(byp, _native.c_u8p), ctypes.byref(yb), ctypes.byref(stage))
        d_bac_bits = int(streams.bac_bits[m]); d_byp_bits = int(streams.bypass_bits[m])
        nb = (bb.value + 7)//8
        same_bac = d_bac_bits == bb.value and numpy.array_equal(raw[m, :nb], bac[:nb])
        ny = (yb.value + 7)//8
        same_byp = d_byp_bits == yb.value and numpy.array_equal(raw[m, half:half + ny], byp[:ny])
        first = None
        if not same_bac:
            w = numpy.flatnonzero(raw[m, :nb] != bac[:nb])
            first = (int(w[0]) if w.size else None, w.size)
        print('  map', m, 'lane', m % 64, 'host bits', bb.value, yb.value, 'device bits', d_bac_bits, d_byp_bits, 'bac same', same_bac, 'first diff byte/count', first, 'byp same', same_byp)
    # the decoder core's prefix bytes of the bad maps (workspace piece B) against min(|s|, L)
    ws = c._workspaces[slot]
    base = (-ws.data_ptr()) % 256
    n_maps = batch*128
    size = planar.shape[1]
    dcap = (size*11 + 7)//8*8
    r256 = lambda v: (v + 255)//256*256
    piece_a = r256(max((n_maps + 63)//64*64*dcap, n_maps*size*2))
    off = base + r256(n_maps*4) + piece_a
    pref = ws[off:off + n_maps*size].cpu().numpy().reshape(n_maps, size)
    dec = ws[base + r256(n_maps*4):base + r256(n_maps*4) + n_maps*size*2].view(torch.int16).cpu().numpy().reshape(n_maps, size)
    dbg_off = off + (n_maps*size + 3)//4*4
    dbg = ws[dbg_off:dbg_off + n_maps*8].view(torch.int32).cpu().numpy().reshape(n_maps, 2)
    blk = int(bad[0])//64
    print('  first-fly snapshot of block', blk, ': (lane, symbol index, rword, flying)',
          [(l, int(dbg[blk*64 + l, 0]), int(dbg[blk*64 + l, 1] & 0x7fffffff), bool(dbg[blk*64 + l, 1] < 0)) for l in range(64) if dbg[blk*64 + l, 0] > 0][:64])
    for m in bad[:12]:
        m = int(m)
        want = numpy.minimum(numpy.abs(planar[m].astype(numpy.int32)), 10).astype(numpy.uint8)
        w = numpy.flatnonzero(pref[m] != want)
        w2 = numpy.flatnonzero(dec[m] != planar[m])
        print('  map', m, 'prefix wrong at', w[:8], 'count', w.size, 'got', pref[m][w[:8]], 'want', want[w[:8]], '| symbols wrong at', w2[:8], 'count', w2.size,
              'nonzero symbols', int((planar[m] != 0).sum()), 'first nonzero at', numpy.flatnonzero(planar[m])[:3])
    # and a fresh verify of the same slot, coder alone now
    st = streams.status.clone()
    dev.coder_decode_batch(streams, c.probabilities, c.prob_row, expected=c._symbols[slot].view(batch*128, -1), workspace=c._workspaces[slot])
    torch.cuda.synchronize()
    print('  re-verify alone: statuses', numpy.unique(streams.status.cpu().numpy()))
    break
if not found:
    print('no failure in 60 rounds')
c.close()
