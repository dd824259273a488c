"""The lossless coder on the GPU (eae_hip_coder_compress_maps / eae_hip_coder_decode_maps, one map per lane) against the
host C-ABI coder (itself pinned to the reference's C++ coder by tests/test_coder_host.py), the C oracle and the golden
streams produced by the real reference build: bit counts, stream BYTES, decoded symbols, error codes and stages are
identical."""
import os
import re
import sys

import numpy
import pytest
import torch

from autoencoder_based_image_compression_amd import _native
from oracle import coder as oc

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'coder_golden.npz')


@pytest.fixture(scope='module')
def gold():
    with numpy.load(GOLD, allow_pickle=False) as g:
        return {k: g[k] for k in g.files}


@pytest.fixture(scope='module')
def dev():
    from autoencoder_based_image_compression_amd import device
    return device


def host_encode_maps(planar, probs, prob_row):
    """eae_coder_encode_maps on the host: (streams [n, stride], bac_bits, bypass_bits, status, stage)."""
    lib = _native.coder()
    (n, size) = planar.shape
    L = probs.shape[1]
    stride = int(_native.hip().eae_hip_coder_stream_stride_bytes(size, L))
    streams = numpy.zeros((n, stride), dtype=numpy.uint8)
    (bac, byp) = (numpy.zeros(n, dtype=numpy.uint32), numpy.zeros(n, dtype=numpy.uint32))
    (status, stage) = (numpy.zeros(n, dtype=numpy.int32), numpy.zeros(n, dtype=numpy.int32))
    pp = numpy.ascontiguousarray(probs, dtype=numpy.float64)
    rows = numpy.ascontiguousarray(prob_row, dtype=numpy.int32)
    lib.eae_coder_encode_maps(n, size, _native.ptr(planar, _native.c_i16p), L, _native.ptr(pp, _native.c_f64p),
                              _native.ptr(rows, _native.c_i32p), _native.ptr(streams, _native.c_u8p), stride,
                              _native.ptr(bac, _native.c_u32p), _native.ptr(byp, _native.c_u32p),
                              _native.ptr(status, _native.c_i32p), _native.ptr(stage, _native.c_i32p), 4)
    return streams, bac, byp, status, stage


def device_code(dev, planar, probs, prob_row, mode, lanes=0):
    sym = torch.from_numpy(planar).cuda()
    p = torch.from_numpy(numpy.ascontiguousarray(probs, dtype=numpy.float64)).cuda()
    rows = torch.from_numpy(numpy.ascontiguousarray(prob_row, dtype=numpy.int32)).cuda()
    (streams, rec) = dev.coder_compress_maps(sym, p, rows, probs.shape[1], mode=mode, lanes_per_wave=lanes)
    torch.cuda.synchronize()
    return streams, rec, p, rows


def valid_bytes_equal(a, b, bits):
    nbytes = (int(bits) + 7)//8
    return numpy.array_equal(a[:nbytes], b[:nbytes])


@pytest.mark.parametrize('lanes', [0, 1, 8, 64])
def test_streams_bits_and_symbols_equal_the_host_coder(gold, dev, lanes):
    rng = numpy.random.RandomState(3)
    probs = gold['real_probabilities_1']
    planar = numpy.round(rng.laplace(size=(3*128, 96))*numpy.tile(rng.uniform(0.1, 4., size=128), 3)[:, None]).astype(numpy.int16)
    prob_row = numpy.tile(numpy.arange(128, dtype=numpy.int32), 3)
    prob_row[67::128] = -1
    (h_streams, h_bac, h_byp, h_status, _) = host_encode_maps(planar, probs, prob_row)
    assert not h_status.any()
    (streams, rec, p, rows) = device_code(dev, planar, probs, prob_row, dev.CODER_ROUNDTRIP, lanes)
    assert not streams.status.cpu().numpy().any()
    assert numpy.array_equal(streams.bac_bits.cpu().numpy().astype(numpy.uint32), h_bac)
    assert numpy.array_equal(streams.bypass_bits.cpu().numpy().astype(numpy.uint32), h_byp)
    assert numpy.array_equal(rec.cpu().numpy(), planar)            # skipped maps are copied verbatim
    d = streams.streams.cpu().numpy()
    half = streams.stride//2
    for m in range(planar.shape[0]):
        assert valid_bytes_equal(d[m], h_streams[m], h_bac[m]), m
        assert valid_bytes_equal(d[m, half:], h_streams[m, half:], h_byp[m]), m
    # the decoder on its own, from the device streams and from the HOST streams
    out = dev.coder_decode_maps(streams, p, rows, lanes).cpu().numpy()
    keep = prob_row >= 0
    assert numpy.array_equal(out[keep], planar[keep]) and not out[~keep].any()
    streams.streams.copy_(torch.from_numpy(h_streams))
    out = dev.coder_decode_maps(streams, p, rows, lanes).cpu().numpy()
    assert numpy.array_equal(out[keep], planar[keep])
    # one map against the C oracle directly
    ref = oc.CoderLib('oracle').compress_lossless(planar[9], probs[9], want_streams=True)
    assert (int(h_bac[9]), int(h_byp[9])) == (ref[2]['bac_bits'], ref[2]['bypass_bits'])
    assert numpy.array_equal(d[9, :ref[2]['bac_bytes'].size], ref[2]['bac_bytes'])
    assert numpy.array_equal(d[9, half:half + ref[2]['bypass_bytes'].size], ref[2]['bypass_bytes'])


def test_golden_streams_of_the_reference_build(gold, dev):
    """tests/golden/coder_golden.npz: streams dumped from the reference's own C++ classes (oracle/gen_golden.py)."""
    for i in range(int(gold['nb_cases'])):
        x = numpy.ascontiguousarray(gold['case{}_in'.format(i)]).reshape(1, -1)
        p = gold['case{}_p'.format(i)].reshape(1, -1)
        if x.size == 0:
            continue
        (streams, rec, _, _) = device_code(dev, x, p, numpy.zeros(1, dtype=numpy.int32), dev.CODER_ROUNDTRIP)
        assert int(streams.status.item()) == 0, i
        assert (int(streams.bac_bits.item()), int(streams.bypass_bits.item())) == (int(gold['case{}_bac_bits'.format(i)]),
                                                                                  int(gold['case{}_byp_bits'.format(i)])), i
        d = streams.streams.cpu().numpy()[0]
        (bac, byp) = (gold['case{}_bac'.format(i)], gold['case{}_byp'.format(i)])
        assert numpy.array_equal(d[:bac.size], bac) and numpy.array_equal(d[streams.stride//2:streams.stride//2 + byp.size], byp), i
        assert numpy.array_equal(rec.cpu().numpy(), x), i


def test_error_codes_and_stages_equal_the_host_coder(gold, dev):
    """Capacity / probability / out-of-range failures: same per-map status and stage as the host library, whose
    messages tests/test_coder_host.py pins to the reference's (compression.cpp:32-62)."""
    seen = set()
    for i in range(int(gold['nb_err_cases'])):
        x = numpy.ascontiguousarray(gold['err{}_in'.format(i)]).reshape(1, -1)
        p = gold['err{}_p'.format(i)].reshape(1, -1)
        if x.size == 0 or p.size == 0:
            continue
        (_, _, _, h_status, h_stage) = host_encode_maps(x, p, numpy.zeros(1, dtype=numpy.int32))
        (streams, _, _, _) = device_code(dev, x, p, numpy.zeros(1, dtype=numpy.int32), dev.CODER_ROUNDTRIP_VERIFY)
        if h_status[0] != 0:       # encode-side failures are what encode_maps can show
            assert (int(streams.status.item()), int(streams.stage.item())) == (int(h_status[0]), int(h_stage[0])), i
        seen.add(int(streams.status.item()))
    assert {1, 4} <= seen          # capacity and probability errors are both exercised by the golden cases
    # mixed launch: one bad map does not disturb its neighbours
    planar = numpy.zeros((4, 16), dtype=numpy.int16)
    planar[2] = 3
    probs = numpy.full((4, 5), 0.5)
    probs[2, 1] = numpy.nan
    (streams, _, _, _) = device_code(dev, planar, probs, numpy.arange(4), dev.CODER_ROUNDTRIP_VERIFY)
    assert streams.status.cpu().tolist() == [0, 0, 4, 0] and streams.stage.cpu().tolist() == [0, 0, 1, 0]


def test_fuzz_against_host_coder_and_oracle(dev):
    orc = oc.CoderLib('oracle')
    rng = numpy.random.RandomState(77)
    for t in range(120):
        n_maps = int(rng.randint(1, 9))
        size = int(rng.randint(1, 300))
        L = int(rng.randint(1, 60))
        scale = rng.choice([0.2, 1, 3, 10, 100, 5000], size=(n_maps, 1))
        planar = numpy.clip(numpy.round(rng.laplace(size=(n_maps, size))*scale), -32768, 32767).astype(numpy.int16)
        probs = numpy.clip(rng.rand(n_maps, L), 0.005, 0.995)
        rows = numpy.arange(n_maps, dtype=numpy.int32)
        (h_streams, h_bac, h_byp, h_status, h_stage) = host_encode_maps(planar, probs, rows)
        (streams, _, _, _) = device_code(dev, planar, probs, rows, dev.CODER_ROUNDTRIP_VERIFY, lanes=int(rng.choice([0, 1, 4, 64])))
        status = streams.status.cpu().numpy()
        assert numpy.array_equal(status, h_status), t
        assert numpy.array_equal(streams.stage.cpu().numpy(), h_stage), t
        ok = status == 0
        assert numpy.array_equal(streams.bac_bits.cpu().numpy()[ok], h_bac[ok].astype(numpy.int32))
        assert numpy.array_equal(streams.bypass_bits.cpu().numpy()[ok], h_byp[ok].astype(numpy.int32))
        d = streams.streams.cpu().numpy()
        half = streams.stride//2
        for m in numpy.nonzero(ok)[0]:
            assert valid_bytes_equal(d[m], h_streams[m], h_bac[m]) and valid_bytes_equal(d[m, half:], h_streams[m, half:], h_byp[m]), (t, m)
        if ok[0]:
            assert orc.compress_lossless(planar[0], probs[0])[1] == int(h_bac[0] + h_byp[0])


def test_full_kodak_batch_verifies_on_the_device(gold, dev):
    """BASELINE.json configs[1] size: 24 images x 128 maps of 32x48 symbols, roundtrip verified in registers; bit counts
    equal the threaded host coder's."""
    from autoencoder_based_image_compression_amd.kodak.lossless import compression
    rng = numpy.random.RandomState(11)
    probs = gold['real_probabilities_1']
    planar = numpy.round(rng.laplace(size=(24, 128, 1536))*rng.uniform(0.05, 6., size=(1, 128, 1))).astype(numpy.int16)
    prob_row = numpy.tile(numpy.arange(128, dtype=numpy.int32), 24)
    prob_row[67::128] = -1
    (_, nb_bits) = compression.code_planar_symbols(planar, probs, idx_map_exception=67, nb_threads=8, roundtrip=False)
    (streams, _, _, _) = device_code(dev, planar.reshape(-1, 1536), probs, prob_row, dev.CODER_ROUNDTRIP_VERIFY)
    assert not streams.status.cpu().numpy().any()
    assert numpy.array_equal(streams.nb_bits().cpu().numpy().reshape(24, 128), nb_bits.astype(numpy.int32))


def test_verify_launch_detects_a_corrupted_stream(gold, dev):
    """eae_hip_coder_verify_maps = the decode half of compress_lossless + the comparison of compression.py:146-153."""
    rng = numpy.random.RandomState(5)
    probs = gold['real_probabilities_1']
    planar = numpy.round(rng.laplace(size=(128, 1536))*rng.uniform(0.3, 4., size=(128, 1))).astype(numpy.int16)
    rows = numpy.arange(128, dtype=numpy.int32)
    rows[67] = -1
    (streams, _, p, r) = device_code(dev, planar, probs, rows, dev.CODER_ENCODE_ONLY)
    sym = torch.from_numpy(planar).cuda()
    dev.coder_verify_maps(streams, sym, p, r)
    assert not streams.status.cpu().numpy().any()
    streams.streams[5, 3] ^= 0x10            # one flipped bit in the arithmetic-coded stream of map 5
    streams.streams[67, 0] ^= 0xFF           # the skipped map's region is never read
    dev.coder_verify_maps(streams, sym, p, r)
    status = streams.status.cpu().numpy()
    assert status[5] != 0 and not numpy.delete(status, 5).any()
    # a failed encode is not overwritten by the verify launch
    streams.status[9] = 1
    dev.coder_verify_maps(streams, sym, p, r)
    assert int(streams.status[9].item()) == 1


# ---- the 64-maps-per-wavefront organisation (eae_hip_coder_encode_batch / decode_batch) ----------------------------------

def batch_code(dev, planar, probs, prob_row):
    sym = torch.from_numpy(planar).cuda()
    p = torch.from_numpy(numpy.ascontiguousarray(probs, dtype=numpy.float64)).cuda()
    rows = torch.from_numpy(numpy.ascontiguousarray(prob_row, dtype=numpy.int32)).cuda()
    streams = dev.coder_encode_batch(sym, p, rows, probs.shape[1])
    torch.cuda.synchronize()
    return streams, sym, p, rows


def assert_equals_host(streams, planar, probs, prob_row, tag=''):
    (h_streams, h_bac, h_byp, h_status, h_stage) = host_encode_maps(planar, probs, prob_row)
    status = streams.status.cpu().numpy()
    assert numpy.array_equal(status, h_status), (tag, status, h_status)
    assert numpy.array_equal(streams.stage.cpu().numpy(), h_stage), tag
    ok = status == 0
    assert numpy.array_equal(streams.bac_bits.cpu().numpy()[ok], h_bac[ok].astype(numpy.int32)), tag
    assert numpy.array_equal(streams.bypass_bits.cpu().numpy()[ok], h_byp[ok].astype(numpy.int32)), tag
    d = streams.streams.cpu().numpy()
    half = streams.stride//2
    for m in numpy.nonzero(ok)[0]:
        assert valid_bytes_equal(d[m], h_streams[m], h_bac[m]), (tag, m)
        assert valid_bytes_equal(d[m, half:], h_streams[m, half:], h_byp[m]), (tag, m)
    return ok


@pytest.mark.parametrize('scale', [0.3, 2., 40., 3000.])
def test_batch_encoder_and_decoder_equal_the_host_coder(gold, dev, scale):
    """Sparse maps, dense maps (streams longer than the decoder's LDS window -> general kernel), Exp-Golomb escapes."""
    rng = numpy.random.RandomState(int(scale*10))
    probs = gold['real_probabilities_1']
    n = 3*128 + 5                                               # a ragged last group of 64
    planar = numpy.clip(numpy.round(rng.laplace(size=(n, 96))*rng.uniform(0.1, 1., size=(n, 1))*scale), -32768, 32767).astype(numpy.int16)
    planar[7] = 0                                               # a dead map: a stream of a few bits
    prob_row = (numpy.arange(n) % 128).astype(numpy.int32)
    prob_row[67::128] = -1
    (streams, sym, p, rows) = batch_code(dev, planar, probs, prob_row)
    ok = assert_equals_host(streams, planar, probs, prob_row, scale)
    assert ok.all()
    out = dev.coder_decode_batch(streams, p, rows).cpu().numpy()
    keep = prob_row >= 0
    assert not streams.status.cpu().numpy().any()
    assert numpy.array_equal(out[keep], planar[keep]) and not out[~keep].any()
    # the same decode with a workspace: the lean core (ring-fed, one prefix byte per symbol) + the data-parallel pass
    ws = dev.coder_workspace(n, 96, probs.shape[1], 'cuda')
    out = dev.coder_decode_batch(streams, p, rows, workspace=ws).cpu().numpy()
    assert not streams.status.cpu().numpy().any()
    assert numpy.array_equal(out[keep], planar[keep]) and not out[~keep].any()
    dev.coder_decode_batch(streams, p, rows, expected=sym)
    assert not streams.status.cpu().numpy().any()
    # a flipped bit in one arithmetic-coded stream is found by the comparison, and only there
    streams.streams[5, 1] ^= 0x04
    dev.coder_decode_batch(streams, p, rows, expected=sym)
    status = streams.status.cpu().numpy()
    assert status[5] != 0 and not numpy.delete(status, 5).any()


@pytest.mark.parametrize('n', [70, 256, 257])
def test_batch_decoder_small_steps_long_bypass_streams(gold, dev, n):
    """The data-parallel pass behind the decoder core has two forms (round 6): at most 256 maps -> `debinarise_kernel<true>`, which
    stages prefix bytes and expected symbols 1,024 at a time and the first 512 words of the bypass stream in LDS (longer streams: the
    rest tile by tile from memory); more -> the plain form. Maps of 2,500 symbols (two chunk boundaries and a ragged chunk), bypass
    streams from a few bits to thousands of words in one batch, a ragged last group: symbols and statuses equal the host coder's."""
    rng = numpy.random.RandomState(n)
    probs = gold['real_probabilities_1']
    scale = rng.choice([0.05, 0.5, 5., 300., 3000.], size=(n, 1))
    planar = numpy.clip(numpy.round(rng.laplace(size=(n, 2500))*scale), -32768, 32767).astype(numpy.int16)
    prob_row = (numpy.arange(n) % 128).astype(numpy.int32)
    prob_row[67::128] = -1
    (streams, sym, p, rows) = batch_code(dev, planar, probs, prob_row)
    assert assert_equals_host(streams, planar, probs, prob_row, n).all()
    bypass_words = (streams.bypass_bits.cpu().numpy().astype(numpy.int64) + 31)//32
    assert bypass_words.max() > 1500 and bypass_words[prob_row >= 0].min() < 64       # both sides of the 512 staged words
    keep = prob_row >= 0
    ws = dev.coder_workspace(n, 2500, probs.shape[1], 'cuda')
    out = dev.coder_decode_batch(streams, p, rows, workspace=ws).cpu().numpy()
    assert not streams.status.cpu().numpy().any()
    assert numpy.array_equal(out[keep], planar[keep])
    dev.coder_decode_batch(streams, p, rows, expected=sym, workspace=ws)
    assert not streams.status.cpu().numpy().any()
    other = sym.clone()
    other[3, 2499] += 1                                         # the last symbol of the ragged chunk
    other[5, 1024] -= 1                                         # the first symbol of the second chunk
    dev.coder_decode_batch(streams, p, rows, expected=other, workspace=ws)
    status = streams.status.cpu().numpy()
    assert status[3] == 6 and status[5] == 6 and not numpy.delete(status, [3, 5]).any()


def test_batch_large_maps_long_pending_runs_and_every_window_tier(gold, dev):
    """Maps of 128 x 128 latents (a 2048 x 2048 image): near-dead maps under a very skewed p0 build runs of pending E3 bits far
    beyond 47 (emit_kernel's long-queue path and its give-up to the general kernel), and the stream lengths run from a few words
    (inside the decoder ring's first fill of 32) to thousands (refilled hundreds of times; round 2's decoder had window tiers
    here, hence the test's name). Bytes, bit counts and decoded symbols equal the host coder's."""
    rng = numpy.random.RandomState(77)
    size = 128*128
    L = 10
    n = 70
    planar = numpy.zeros((n, size), dtype=numpy.int16)
    probs = numpy.tile(gold['real_probabilities_1'][0], (n, 1))
    # 0-19: (almost) dead maps, p0 from 0.9 to 0.99999: long runs of identical decisions
    for m in range(20):
        probs[m, 0] = 1. - 10.**(-1. - 0.2*m)
        if m % 2:
            planar[m, rng.randint(0, size, size=m)] = rng.choice([-2, -1, 1, 3], size=m)
    # 20-69: densities from a few hundred bits to several times the largest window
    for m in range(20, n):
        density = 0.0005*1.25**(m - 20)
        hits = rng.rand(size) < min(density, 0.9)
        planar[m, hits] = numpy.clip(numpy.round(rng.laplace(size=int(hits.sum()))*3.), -300, 300).astype(numpy.int16)
    rows = numpy.arange(n, dtype=numpy.int32)
    (streams, sym, p, r) = batch_code(dev, planar, probs, rows)
    ok = assert_equals_host(streams, planar, probs, rows, 'large')
    assert ok.all()
    bits = streams.bac_bits.cpu().numpy()
    assert (bits < 64*32).any() and ((bits > 64*32) & (bits <= 192*32)).any() and ((bits > 192*32) & (bits <= 448*32)).any() \
        and (bits > 448*32).any(), bits
    out = dev.coder_decode_batch(streams, p, r).cpu().numpy()
    assert not streams.status.cpu().numpy().any()
    assert numpy.array_equal(out, planar)
    out = dev.coder_decode_batch(streams, p, r, workspace=dev.coder_workspace(n, size, L, 'cuda')).cpu().numpy()     # lean core + debinarise pass
    assert not streams.status.cpu().numpy().any()
    assert numpy.array_equal(out, planar)
    dev.coder_decode_batch(streams, p, r, expected=sym)
    assert not streams.status.cpu().numpy().any()
    # one corrupted stream per stream-length class (inside the ring's first fill / refilled several times) is found, and only those
    order = [int(m) for m in numpy.argsort(bits) if bits[m] >= 64]      # the flipped bit must be a coded one
    victims = [order[0], order[len(order)//2], order[-1]]
    for m in victims:
        streams.streams[m, 0] ^= 0x20
    dev.coder_decode_batch(streams, p, r, expected=sym)
    status = streams.status.cpu().numpy()
    assert all(status[m] != 0 for m in victims) and not numpy.delete(status, victims).any()


def test_batch_golden_streams_of_the_reference_build(gold, dev):
    for i in range(int(gold['nb_cases'])):
        x = numpy.ascontiguousarray(gold['case{}_in'.format(i)]).reshape(1, -1)
        p = gold['case{}_p'.format(i)].reshape(1, -1)
        if x.size == 0:
            continue
        (streams, sym, pd, rows) = batch_code(dev, x, p, numpy.zeros(1, dtype=numpy.int32))
        assert int(streams.status.item()) == 0, i
        assert (int(streams.bac_bits.item()), int(streams.bypass_bits.item())) == (int(gold['case{}_bac_bits'.format(i)]),
                                                                                  int(gold['case{}_byp_bits'.format(i)])), i
        d = streams.streams.cpu().numpy()[0]
        (bac, byp) = (gold['case{}_bac'.format(i)], gold['case{}_byp'.format(i)])
        assert numpy.array_equal(d[:bac.size], bac) and numpy.array_equal(d[streams.stride//2:streams.stride//2 + byp.size], byp), i
        assert numpy.array_equal(dev.coder_decode_batch(streams, pd, rows).cpu().numpy(), x), i


def test_batch_fuzz_against_host_coder_including_errors(gold, dev):
    """Random sizes, L (including L > 32, which the fast kernels hand to the general one), magnitudes and a few invalid
    probabilities: statuses, stages, bits and bytes equal the host library's; what encodes also decodes."""
    rng = numpy.random.RandomState(123)
    seen = set()
    for t in range(150):
        n_maps = int(rng.randint(1, 140))
        size = int(rng.randint(1, 200))
        L = int(rng.choice([1, 2, 5, 10, 31, 32, 33, 60]))
        scale = rng.choice([0.2, 1, 3, 10, 100, 5000], size=(n_maps, 1))
        planar = numpy.clip(numpy.round(rng.laplace(size=(n_maps, size))*scale), -32768, 32767).astype(numpy.int16)
        probs = numpy.clip(rng.rand(n_maps, L), 0.005, 0.995)
        if t % 5 == 0:
            probs[rng.randint(n_maps), rng.randint(L)] = rng.choice([0., 1., numpy.nan, -0.2])
        rows = numpy.arange(n_maps, dtype=numpy.int32)
        if t % 7 == 0:
            rows[rng.randint(n_maps)] = -1
        (streams, sym, p, r) = batch_code(dev, planar, probs, rows)
        ok = assert_equals_host(streams, planar, probs, rows, t)
        seen.update(int(v) for v in streams.status.cpu().numpy())
        encode_status = streams.status.clone()
        dev.coder_decode_batch(streams, p, r, expected=sym)
        assert torch.equal(streams.status, encode_status), t          # nothing new: every encoded map decodes to its input
        out = dev.coder_decode_batch(streams, p, r).cpu().numpy()     # pure decode rewrites every status
        good = ok & (rows >= 0)
        assert numpy.array_equal(out[good], planar[good]), t
        two_pass_status = streams.status.clone()
        out = dev.coder_decode_batch(streams, p, r, workspace=dev.coder_workspace(n_maps, size, L, 'cuda')).cpu().numpy()   # lean core + debinarise pass
        assert numpy.array_equal(out[good], planar[good]), t
        assert torch.equal(streams.status, two_pass_status) and torch.equal(streams.stage[streams.status != 0], streams.stage[two_pass_status != 0]), t
    assert {0, 1, 4} <= seen


def test_batch_full_kodak_batch(gold, dev):
    from autoencoder_based_image_compression_amd.kodak.lossless import compression
    rng = numpy.random.RandomState(11)
    probs = gold['real_probabilities_1']
    planar = numpy.round(rng.laplace(size=(24, 128, 1536))*rng.uniform(0.05, 6., size=(1, 128, 1))).astype(numpy.int16)
    prob_row = numpy.tile(numpy.arange(128, dtype=numpy.int32), 24)
    prob_row[67::128] = -1
    (_, nb_bits) = compression.code_planar_symbols(planar, probs, idx_map_exception=67, nb_threads=8, roundtrip=False)
    (streams, sym, p, rows) = batch_code(dev, planar.reshape(-1, 1536), probs, prob_row)
    assert not streams.status.cpu().numpy().any()
    assert numpy.array_equal(streams.nb_bits().cpu().numpy().reshape(24, 128), nb_bits.astype(numpy.int32))
    dev.coder_decode_batch(streams, p, rows, expected=sym)
    assert not streams.status.cpu().numpy().any()
    # and the per-lane kernels agree byte for byte
    (ref_streams, _, _, _) = device_code(dev, planar.reshape(-1, 1536), probs, prob_row, dev.CODER_ENCODE_ONLY)
    half = streams.stride//2
    a = streams.streams.cpu().numpy()
    b = ref_streams.streams.cpu().numpy()
    bac = streams.bac_bits.cpu().numpy()
    byp = streams.bypass_bits.cpu().numpy()
    for m in range(0, a.shape[0], 37):
        assert valid_bytes_equal(a[m], b[m], bac[m]) and valid_bytes_equal(a[m, half:], b[m, half:], byp[m]), m


# ---- the chunked round trip: emit pass and decoder trailing the encoder core (eae_hip_coder_roundtrip_trailing) ------------
# Experimental, compiled into the test build only (include/eae_hip.h, -DEAE_EXPERIMENTAL_CODER): these tests run on
# lib/libeae_hip_test.so (fixture `test_library`, tests/conftest.py).

def trailing_code(dev, planar, probs, prob_row, chunks):
    """chunks == 'fused': the three serial stages of a group as one workgroup (eae_hip_coder_roundtrip_fused)."""
    sym = torch.from_numpy(planar).cuda()
    p = torch.from_numpy(numpy.ascontiguousarray(probs, dtype=numpy.float64)).cuda()
    rows = torch.from_numpy(numpy.ascontiguousarray(prob_row, dtype=numpy.int32)).cuda()
    if chunks == 'fused':
        streams = dev.coder_roundtrip_fused(sym, p, rows, probs.shape[1])
    else:
        streams = dev.coder_roundtrip_trailing(sym, p, rows, probs.shape[1], chunks=chunks)
    torch.cuda.synchronize()
    return streams, sym, p, rows


@pytest.mark.parametrize('chunks', [1, 2, 3, 4, 8, 16, 'fused'])
@pytest.mark.parametrize('scale', [0.3, 4., 300.])
def test_trailing_round_trip_equals_the_host_coder(gold, dev, scale, chunks, test_library):
    """One Kodak image's worth of maps (and a ragged second group) at three densities: whatever the number of chunks, the streams,
    bit counts, statuses and stages are the host coder's, and the round trip finds nothing to complain about."""
    rng = numpy.random.RandomState(int(scale*10) + (99 if chunks == 'fused' else chunks))
    probs = gold['real_probabilities_1']
    n = 128 + 37
    planar = numpy.clip(numpy.round(rng.laplace(size=(n, 1536))*rng.uniform(0.05, 1., size=(n, 1))*scale), -32768, 32767).astype(numpy.int16)
    planar[7] = 0
    planar[130, :40] = 0
    prob_row = (numpy.arange(n) % 128).astype(numpy.int32)
    prob_row[67::128] = -1
    (streams, sym, p, rows) = trailing_code(dev, planar, probs, prob_row, chunks)
    ok = assert_equals_host(streams, planar, probs, prob_row, (scale, chunks))
    assert ok.all()
    # the streams it wrote decode with the ordinary decoder too
    out = dev.coder_decode_batch(streams, p, rows).cpu().numpy()
    keep = prob_row >= 0
    assert numpy.array_equal(out[keep], planar[keep])


def test_trailing_round_trip_large_maps_and_long_pending_runs(gold, dev, test_library):
    """The maps of test_batch_large_maps_long_pending_runs_and_every_window_tier (128 x 128 latents; nearly dead maps whose pending
    E3 runs make the emit pass give up, streams from a few words to thousands): a decoder chunk that finds too few bits in memory
    parks, the general kernel recodes what the fast encoder hands over, and the result is the host coder's."""
    rng = numpy.random.RandomState(77)
    size = 128*128
    n = 70
    planar = numpy.zeros((n, size), dtype=numpy.int16)
    probs = numpy.tile(gold['real_probabilities_1'][0], (n, 1))
    for m in range(20):
        probs[m, 0] = 1. - 10.**(-1. - 0.2*m)
        if m % 2:
            planar[m, rng.randint(0, size, size=m)] = rng.choice([-2, -1, 1, 3], size=m)
    for m in range(20, n):
        hits = rng.rand(size) < min(0.0005*1.25**(m - 20), 0.9)
        planar[m, hits] = numpy.clip(numpy.round(rng.laplace(size=int(hits.sum()))*3.), -300, 300).astype(numpy.int16)
    rows = numpy.arange(n, dtype=numpy.int32)
    for chunks in (4, 7, 'fused'):
        (streams, sym, p, r) = trailing_code(dev, planar, probs, rows, chunks)
        assert assert_equals_host(streams, planar, probs, rows, ('large', chunks)).all()


def test_trailing_round_trip_fuzz_including_errors(gold, dev, test_library):
    """Random sizes, L, magnitudes, a few invalid probabilities and skipped maps: the statuses and stages after the chunked round
    trip are those after encode_batch + decode_batch(expected), the bytes and bit counts the host coder's."""
    rng = numpy.random.RandomState(321)
    seen = set()
    for t in range(120):
        n_maps = int(rng.randint(1, 200))
        size = int(rng.randint(1, 400))
        L = int(rng.choice([1, 2, 5, 10, 31, 32, 33]))
        chunks = 'fused' if t % 2 else int(rng.choice([2, 3, 4, 5, 9]))
        scale = rng.choice([0.2, 1, 3, 10, 100, 5000], size=(n_maps, 1))
        planar = numpy.clip(numpy.round(rng.laplace(size=(n_maps, size))*scale), -32768, 32767).astype(numpy.int16)
        probs = numpy.clip(rng.rand(n_maps, L), 0.005, 0.995)
        if t % 5 == 0:
            probs[rng.randint(n_maps), rng.randint(L)] = rng.choice([0., 1., numpy.nan, -0.2])
        rows = numpy.arange(n_maps, dtype=numpy.int32)
        if t % 7 == 0:
            rows[rng.randint(n_maps)] = -1
        (streams, sym, p, r) = trailing_code(dev, planar, probs, rows, chunks)
        assert_equals_host(streams, planar, probs, rows, (t, chunks))
        (two, _, _, _) = batch_code(dev, planar, probs, rows)
        dev.coder_decode_batch(two, p, r, expected=sym)
        assert torch.equal(streams.status, two.status) and torch.equal(streams.stage, two.stage), t
        seen.update(int(v) for v in streams.status.cpu().numpy())
    assert {0, 1, 4} <= seen


def test_trailing_round_trip_finds_a_difference(gold, dev, test_library):
    """The comparison at the end of the round trip is a real one: symbols that change between the encoder's read and the
    decoder's comparison (here: the caller's buffer rewritten behind the call) give status 6 for that map only."""
    rng = numpy.random.RandomState(5)
    probs = gold['real_probabilities_1']
    planar = numpy.round(rng.laplace(size=(128, 1536))*2.).astype(numpy.int16)
    rows = numpy.arange(128, dtype=numpy.int32)
    sym = torch.from_numpy(planar).cuda()
    p = torch.from_numpy(probs).cuda()
    r = torch.from_numpy(rows).cuda()
    streams = dev.coder_roundtrip_trailing(sym, p, r, 10, chunks=4)
    torch.cuda.synchronize()
    assert not streams.status.cpu().numpy().any()
    other = sym.clone()
    other[9, 100] += 1
    dev.coder_decode_batch(streams, p, r, expected=other)
    status = streams.status.cpu().numpy()
    assert status[9] == 6 and not numpy.delete(status, 9).any()


def test_argument_checks(dev):
    lib = _native.hip()
    assert lib.eae_hip_coder_compress_maps(1, 4, None, None, 3, None, None, None, 64, None, None, None, None, 1, 0, None) == -1
    sym = torch.zeros((1, 64), dtype=torch.int16, device='cuda')
    p = torch.full((1, 10), 0.5, dtype=torch.float64, device='cuda')
    s = dev.CoderStreams(1, 64, 10, sym.device)
    # a stride below capacity + slack is refused before any work, like eae_coder_encode_maps
    rc = lib.eae_hip_coder_compress_maps(1, 64, sym.data_ptr(), None, 10, p.data_ptr(), None, s.streams.data_ptr(), 64,
                                         s.bac_bits.data_ptr(), s.bypass_bits.data_ptr(), s.status.data_ptr(), None, 1, 0, None)
    assert rc == 1
    with pytest.raises(TypeError):
        dev.coder_compress_maps(sym.int(), p, None, 10)


@pytest.mark.parametrize('bin_width', [1.0, 0.125, 0.0125])
def test_batch_coder_next_to_mfma_kernels(dev, bin_width):
    """The coder's wavefronts share their SIMDs with the transforms' MFMA waves in the product (codec.BatchCodec): every kernel of
    the batch coder must give the same bytes and symbols while conv GEMM launches saturate the GPU from another stream. (Round 3:
    a first form of the decoder core passed every stand-alone test and derailed exactly there; this test is the guard.) The
    benchmark's own latents, 24 Kodak-sized images, from 0.2 to 3 bits per pixel: streams of up to ~150 words, so the ring refill
    runs too."""
    import bench
    from autoencoder_based_image_compression_amd import pipeline
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats
    batch = 24
    variables = bench.synthetic_model(bin_width)
    images = torch.from_numpy(bench.synthetic_images(1000, batch, 512, 768)).cuda()
    bin_widths = variables[var.BIN_WIDTHS_NAME]
    enc = pipeline.DeviceEncoder(variables, False)
    y = enc(images)
    map_mean = dev.map_means(y)
    probabilities = lossless_stats.compute_binary_probabilities(y.cpu().numpy(), bin_widths, map_mean.cpu().numpy(), 10)
    q = dev.quantize_maps(y, torch.from_numpy(bin_widths).cuda(), map_mean, want_symbols=True)
    symbols = q['symbols'].reshape(batch*128, -1).contiguous()
    planar = symbols.cpu().numpy()
    rows_host = numpy.tile(numpy.arange(128, dtype=numpy.int32), batch)
    rows_host[67::128] = -1
    rows = torch.from_numpy(rows_host).cuda()
    prob = torch.from_numpy(probabilities).cuda()
    streams = dev.CoderStreams(batch*128, symbols.shape[1], 10, 'cuda')
    ws = dev.coder_workspace(batch*128, symbols.shape[1], 10, 'cuda')
    gdn_1 = dev.conv9x9s4_u8(images, enc.w1, enc.v['encoder/biases_1'], enc.g[1], enc.v['encoder/beta_1'])
    out = torch.empty((batch, 64, 96, 128), device='cuda')
    side = torch.cuda.Stream()
    torch.cuda.synchronize()

    def load():
        for _ in range(6):
            dev.conv5x5s2(gdn_1, enc.w2, enc.v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], enc.v['encoder/beta_2'], out=out)

    for round_ in range(4):
        load()
        with torch.cuda.stream(side):
            dev.coder_encode_batch(symbols, prob, rows, 10, out=streams, workspace=ws)
        load()
        with torch.cuda.stream(side):
            dev.coder_decode_batch(streams, prob, rows, expected=symbols, workspace=ws)
        torch.cuda.synchronize()
        assert int(streams.status.abs().sum().item()) == 0, (round_, numpy.flatnonzero(streams.status.cpu().numpy())[:8])
        if round_ == 0:
            assert_equals_host(streams, planar, probabilities, rows_host, ('under load', bin_width))
    load()
    with torch.cuda.stream(side):
        decoded = dev.coder_decode_batch(streams, prob, rows)
    torch.cuda.synchronize()
    coded = rows_host >= 0
    assert numpy.array_equal(decoded.cpu().numpy()[coded], planar[coded])


def host_decode_maps(streams, bac, byp, probs, prob_row, size):
    """eae_coder_decode_maps on the host over the given bytes: (symbols [n, size], status, stage)."""
    lib = _native.coder()
    n = streams.shape[0]
    out = numpy.zeros((n, size), dtype=numpy.int16)
    (status, stage) = (numpy.zeros(n, dtype=numpy.int32), numpy.zeros(n, dtype=numpy.int32))
    pp = numpy.ascontiguousarray(probs, dtype=numpy.float64)
    rows = numpy.ascontiguousarray(prob_row, dtype=numpy.int32)
    streams = numpy.ascontiguousarray(streams)
    (bac, byp) = (numpy.ascontiguousarray(bac, dtype=numpy.uint32), numpy.ascontiguousarray(byp, dtype=numpy.uint32))
    lib.eae_coder_decode_maps(n, size, _native.ptr(out, _native.c_i16p), probs.shape[1], _native.ptr(pp, _native.c_f64p),
                              _native.ptr(rows, _native.c_i32p), _native.ptr(streams, _native.c_u8p), streams.shape[1],
                              _native.ptr(bac, _native.c_u32p), _native.ptr(byp, _native.c_u32p),
                              _native.ptr(status, _native.c_i32p), _native.ptr(stage, _native.c_i32p), 4)
    return out, status, stage


@pytest.mark.parametrize('scale', [0.5, 3.])
def test_batch_decoder_on_streams_it_did_not_write_equals_the_host_decoder(gold, dev, scale):
    """Corrupted, truncated and foreign streams: flipped bits at the head, in the middle and at the end of the
    arithmetic-coded bytes, bit counts cut short or stretched, streams swapped between maps. A pure decode by the
    64-maps-per-wavefront kernels must return, map for map, the symbols, status and stage of the host library (which follows the
    reference's decode_bit, including its "leave the interval alone when the code is outside it" -- which no stream can bring about:
    coder/lean_step.h: code_inside)."""
    rng = numpy.random.RandomState(int(scale*100) + 1)
    probs = gold['real_probabilities_1']
    n = 2*128
    size = 384
    planar = numpy.clip(numpy.round(rng.laplace(size=(n, size))*rng.uniform(0.2, 1., size=(n, 1))*scale), -32768, 32767).astype(numpy.int16)
    prob_row = (numpy.arange(n) % 128).astype(numpy.int32)
    prob_row[67::128] = -1
    (streams, sym, p, rows) = batch_code(dev, planar, probs, prob_row)
    assert not streams.status.cpu().numpy().any()
    raw = streams.streams.cpu().numpy().copy()
    bac = streams.bac_bits.cpu().numpy().astype(numpy.uint32).copy()
    byp = streams.bypass_bits.cpu().numpy().astype(numpy.uint32).copy()
    coded = numpy.flatnonzero((prob_row >= 0) & (bac >= 48))
    kinds = ('head', 'middle', 'end', 'burst', 'short', 'long', 'swap')
    for (j, m) in enumerate(coded[:14*len(kinds)]):
        kind = kinds[j % len(kinds)]
        nbytes = int(bac[m] + 7)//8
        if kind == 'head':
            raw[m, rng.randint(0, 2)] ^= 1 << rng.randint(0, 8)
        elif kind == 'middle':
            raw[m, nbytes//2] ^= 1 << rng.randint(0, 8)
        elif kind == 'end':
            raw[m, nbytes - 1] ^= 1 << rng.randint(0, 8)
        elif kind == 'burst':
            raw[m, 2:2 + min(6, nbytes - 2)] = rng.randint(0, 256, size=min(6, nbytes - 2)).astype(numpy.uint8)
        elif kind == 'short':
            bac[m] = bac[m]//2
        elif kind == 'long':
            bac[m] = bac[m] + 40                 # bytes behind the stream: whatever an earlier encode left there (zeros here)
        else:
            other = coded[(j + 5) % coded.size]
            raw[m, :nbytes + 8] = raw[other, :nbytes + 8]
    (h_sym, h_status, h_stage) = host_decode_maps(raw, bac, byp, probs, prob_row, size)
    streams.streams.copy_(torch.from_numpy(raw).cuda())
    streams.bac_bits.copy_(torch.from_numpy(bac.astype(numpy.int32)).cuda())
    ws = dev.coder_workspace(n, size, probs.shape[1], 'cuda')
    out = dev.coder_decode_batch(streams, p, rows, workspace=ws).cpu().numpy()
    torch.cuda.synchronize()
    status = streams.status.cpu().numpy()
    assert numpy.array_equal(status, h_status), numpy.flatnonzero(status != h_status)[:8]
    assert numpy.array_equal(streams.stage.cpu().numpy()[status != 0], h_stage[status != 0])
    keep = (prob_row >= 0) & (status == 0)
    differing = numpy.flatnonzero((out != h_sym).any(axis=1) & keep)
    assert differing.size == 0, differing[:8]
    # the streams WERE damaged: most of the touched maps decode to other symbols than the encoder's, a few end in an error
    touched = coded[:14*len(kinds)]
    assert ((out[touched] != planar[touched]).any(axis=1) | (status[touched] != 0)).sum() > touched.size//2


def _private_library(tmp_path, flags):
    """libeae_hip.so with coder_simd.hip rebuilt with extra flags (never the shipped library): the other objects are the build's."""
    import glob
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, 'autoencoder_based_image_compression_amd', 'csrc')
    objects = [o for o in glob.glob(os.path.join(root, 'build', 'hip', '*.o')) if not o.endswith('/coder_simd.o')]
    hipcc = '/opt/rocm/bin/hipcc'
    if not os.path.isfile(hipcc) or len(objects) < 10:
        pytest.skip('needs hipcc and the objects of the build (build/hip/*.o)')
    obj = str(tmp_path / 'coder_simd.o')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt',
                    '-fno-fast-math', '-I' + os.path.join(root, 'include'), '-I' + os.path.join(csrc, 'hip')] + flags +
                   ['-c', '-o', obj, os.path.join(csrc, 'hip', 'coder_simd.hip')], check=True)
    lib = str(tmp_path / 'libeae_hip.so')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objects + [obj], check=True)
    return lib


def test_the_guard_test_is_red_on_the_first_decoder_core(tmp_path):
    """Sensitivity of `test_batch_coder_next_to_mfma_kernels`, re-demonstrated on whatever toolchain builds this tree: round 3's first
    decoder core (-DEAE_DECODE_TOPUP_ZEROS: 40 of 40 VGPRs, its 64-bit window shift fed from v39, the last register of the
    allocation -- csrc/isa_guard.py rule 2) in a PRIVATE library, at the priority of its neighbours, must fail
    that test at every rate; and the ISA guard must name the instruction. If the compiler ever stops producing the pattern from
    this source, the guard stays the judge: the test then only checks that guard and run agree."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'autoencoder_based_image_compression_amd', 'csrc'))
    import isa_guard
    lib = _private_library(tmp_path, ['-DEAE_DECODE_TOPUP_ZEROS', '-DEAE_SIMD_PRIO=0'])
    findings = isa_guard.check([lib])
    env = dict(os.environ, EAE_HIP_LIB=lib)
    run = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-k', 'test_batch_coder_next_to_mfma_kernels',
                          '-p', 'no:cacheprovider'], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)
    if findings:
        assert all('bac_decode_core_kernel' in f and 'rule 2' in f for f in findings), findings
        if run.returncode == 0:
            # the ISA pattern is there (the guard's verdict stands) but this box did not miscompute it: nothing to assert about
            # hardware behaviour that a firmware may change; say so instead of failing the suite
            pytest.skip('the guard rejects the first decoder core, but the guard test passed on it on this box')
        assert 'test_batch_coder_next_to_mfma_kernels' in run.stdout, run.stdout[-2000:]
    else:
        assert run.returncode == 0, run.stdout[-2000:]


def test_the_64_bit_shift_hazard_itself(tmp_path, capsys):
    """The fault in its minimal form (scratch/r04/probe_shift64.hip, generated by gen_probe_shift.py): `v_lshlrev_b64 v[18:19], vK,
    v[16:17]` in a wave whose allocation is N registers, against the same shift done with 32-bit instructions. Asserted: alone on the
    GPU every result is right for every (N, K); next to other waves every result is right when K + 1 < N. Reported, not asserted
    (a later firmware may cure it, and then rule 2 of the guard can go): the wrong results with K = N - 1 next to other waves."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    source = os.path.join(root, 'scratch', 'r04', 'probe_shift64.hip')
    hipcc = '/opt/rocm/bin/hipcc'
    if not os.path.isfile(hipcc) or not os.path.isfile(source):
        pytest.skip('needs hipcc and scratch/r04/probe_shift64.hip')
    exe = str(tmp_path / 'probe_shift64')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O2', '-o', exe, source], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = subprocess.run([exe, '48', '20000', '2'], check=True, stdout=subprocess.PIPE, universal_newlines=True, timeout=600).stdout
    rows = re.findall(r'^(alone|VALU neighbours|MFMA neighbours)\s+N = +(\d+), K = +(\d+).*?wrong results (\d+) of', out, flags=re.M)
    assert len(rows) == 27, out
    hazard = 0
    for (where, n, k, wrong) in rows:
        (n, k, wrong) = (int(n), int(k), int(wrong))
        if where == 'alone' or k + 1 < n:
            assert wrong == 0, (where, n, k, wrong)
        else:
            hazard += wrong
    # said out loud and kept (gpurun_out/ travels back from the GPU box; a copy goes to profiles/): a green run of this test on
    # another box then says whether the mechanism was seen there, not only that the guarded cases were right
    import json
    summary = {'probe': 'scratch/r04/probe_shift64.hip 48 20000 2', 'reproduced': bool(hazard > 0),
               'wrong_results_with_the_shift_amount_in_the_last_register_next_to_other_waves': hazard,
               'cases': [{'where': where, 'N': int(n), 'K': int(k), 'wrong': int(wrong)} for (where, n, k, wrong) in rows]}
    try:
        os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(root, 'gpurun_out', 'shift64_hazard_probe.json'), 'w') as f:
            json.dump(summary, f, indent=1)
    except OSError:
        pass
    with capsys.disabled():
        print('\n[64-bit shift hazard probe] reproduced: {0} ({1} wrong results with the shift amount in the last register of the '
              'allocation, next to other waves; 0 in every other case)'.format('true' if hazard > 0 else 'false', hazard))
    if hazard == 0:
        pytest.skip('the hazard did not reproduce on this box: rule 2 of csrc/isa_guard.py may have become unnecessary')
