"""GPU parity: every gfx950 kernel, called through the C ABI (include/eae_hip.h), against the CPU oracle on the same
seeded inputs. Floating point results are compared with `numpy.array_equal` (exact equality; -0 == +0): the kernels
reproduce the oracle's f32 FMA-chain order, so the stated tolerance is ZERO. Integer results are bit-exact."""
import os

import numpy
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def T():
    import torch
    return torch


@pytest.fixture(scope='module')
def dev():
    from autoencoder_based_image_compression_amd import device
    return device


@pytest.fixture(scope='module')
def orc():
    from oracle import transforms
    return transforms


def _vars(seed=0, learned=False):
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables
    return variables.random_variables(1., learned, seed=seed, bias_std=0.01)


def _cuda(T, a):
    return T.from_numpy(numpy.ascontiguousarray(a)).cuda()


def _image(rng, n, h, w):
    x = rng.randint(16, 236, size=(n, h, w)).astype(numpy.float64)
    for _ in range(3):   # 3x box blur, SURVEY.md 8(d)
        x = (x + numpy.roll(x, 1, 1) + numpy.roll(x, -1, 1) + numpy.roll(x, 1, 2) + numpy.roll(x, -1, 2))/5.
    return numpy.round(x).astype(numpy.uint8)


@pytest.mark.parametrize('shape', [(2, 64, 96), (1, 48, 80), (3, 16, 16), (1, 32, 160)])
@pytest.mark.parametrize('with_gdn', [True, False])
def test_conv1_gdn1(T, dev, orc, shape, with_gdn):
    v = _vars(1)
    x = _image(numpy.random.RandomState(2), *shape)
    ref = orc.conv2d_same(x.astype(numpy.float32)[..., None], v['encoder/weights_1'], 4, v['encoder/biases_1'])
    if with_gdn:
        ref = orc.gdn(ref, v['encoder/gamma_1'], v['encoder/beta_1'])
    got = dev.conv9x9s4_u8(_cuda(T, x), dev.pack_conv9x9s4_weights(_cuda(T, v['encoder/weights_1'])), _cuda(T, v['encoder/biases_1']),
                           dev.pack_gamma(_cuda(T, v['encoder/gamma_1'])) if with_gdn else None,
                           _cuda(T, v['encoder/beta_1']) if with_gdn else None).cpu().numpy()
    assert got.shape == ref.shape
    assert numpy.array_equal(got, ref)


def _select_form(launch_options, form, tile):
    """Every form of the conv GEMM must give the same bits (EAE_HIP_GEMM, csrc/hip/conv_gemm.hip: launch):
    'wave' = conv_gemm_wave_kernel, one tile per wave (32 / 64 / 128-position blocks); 'nt1' / 'nt2' = the same with a
    wave's output channels spread over 4 / 2 waves (small layers): as one-wave blocks, 'nt1p' / 'nt2p' as four-wave blocks holding the
    channel parts of 1 / 2 position tiles (EAE_HIP_PACK: what a launch picks by itself when that gives every CU a block) where the
    layer's width is a multiple of 8 / 16 positions (the (1, 32, 64) and (1, 16, 32) cases below; one-wave blocks otherwise); 'lds' = the block-cooperative LDS form (64- or
    128-position blocks); 'whole' = conv_gemm_split_kernel with whole tiles; 'cut1..3' = conv_gemm_split_kernel with the cut
    forced onto these small shapes (every tile is then cut in two and the tails wait for their heads); 'one2' / 'one3' = the same
    as blocks of one wave (EAE_HIP_SPLIT_WPB=1)."""
    launch_options.clear()
    if form.startswith('one'):            # 'one2': the cut forced, one-wave blocks (EAE_HIP_SPLIT_WPB=1)
        launch_options.setenv('EAE_HIP_GEMM', 's')
        launch_options.setenv('EAE_HIP_SPLIT_WAVES', form[3:])
        launch_options.setenv('EAE_HIP_SPLIT_WPB', '1')
        return
    if form.startswith('cut'):
        launch_options.setenv('EAE_HIP_GEMM', 's')
        launch_options.setenv('EAE_HIP_SPLIT_WAVES', form[3:])
        return
    if form == 'whole':
        launch_options.setenv('EAE_HIP_GEMM', 'u')
        return
    launch_options.setenv('EAE_HIP_GEMM', 'l' if form == 'lds' else 'w')
    launch_options.setenv('EAE_HIP_FORCE_TILE', tile)
    if form.startswith('nt'):
        launch_options.setenv('EAE_HIP_FORCE_NT', form[2])
        launch_options.setenv('EAE_HIP_PACK', '1' if form.endswith('p') else '0')


def _assert_handed_over(T, dev, ws):
    """No tail of a cut launch gave up (eae_hip_conv_workspace_collect reports 0) and the workspace is all zero again."""
    if ws is None:
        return
    word = T.zeros(1, dtype=T.int32, device='cuda')
    dev.conv_workspace_collect(ws, word)
    assert int(word.item()) == 0 and int(T.count_nonzero(ws).item()) == 0


FORMS = [('wave', '32'), ('wave', '64'), ('wave', '128'), ('lds', '64'), ('lds', '128'), ('nt1', '32'), ('nt2', '32'), ('nt1p', '32'), ('nt2p', '32'),
         ('whole', ''), ('cut1', ''), ('cut2', ''), ('cut3', ''), ('one2', ''), ('one3', '')]


def test_conv1_reads_words(T, dev, orc):
    """conv_1 reads the luminance as aligned 32-bit words: an image at an odd byte offset is refused (include/eae_hip.h), one
    at a multiple of 4 inside a larger buffer gives the oracle's bits, and so does a width that leaves the last tile ragged."""
    from autoencoder_based_image_compression_amd import device
    v = _vars(1)
    (wp, b) = (dev.pack_conv9x9s4_weights(_cuda(T, v['encoder/weights_1'])), _cuda(T, v['encoder/biases_1']))
    x = _image(numpy.random.RandomState(5), 1, 36, 100)
    ref = orc.conv2d_same(x.astype(numpy.float32)[..., None], v['encoder/weights_1'], 4, v['encoder/biases_1'])
    raw = T.zeros(x.size + 16, dtype=T.uint8, device='cuda')
    for offset in (4, 12):
        view = raw[offset:offset + x.size].view(1, 36, 100)
        view.copy_(_cuda(T, x))
        assert numpy.array_equal(dev.conv9x9s4_u8(view, wp, b).cpu().numpy(), ref)
    odd = raw[1:1 + x.size].view(1, 36, 100)
    with pytest.raises(device.HipError):
        dev.conv9x9s4_u8(odd, wp, b)


@pytest.mark.parametrize('form,tile', FORMS)
@pytest.mark.parametrize('shape', [(2, 16, 24), (1, 32, 64), (1, 6, 10), (2, 2, 2), (1, 20, 36)])
@pytest.mark.parametrize('norm', [0, 1])
def test_conv5x5s2(T, dev, orc, shape, norm, form, tile, launch_options):
    _select_form(launch_options, form, tile)
    v = _vars(3)
    rng = numpy.random.RandomState(4)
    x = rng.standard_normal(size=shape + (128,)).astype(numpy.float32)
    ref = orc.conv2d_same(x, v['encoder/weights_2'], 2, v['encoder/biases_2'])
    if norm:
        ref = orc.gdn(ref, v['encoder/gamma_2'], v['encoder/beta_2'])
    wp = dev.pack_conv_weights(_cuda(T, v['encoder/weights_2']))
    perm = dev.packed_channel_order()
    expect_w = numpy.empty((25, 128, 128), dtype=numpy.float32)
    expect_w[:, :, perm] = v['encoder/weights_2'].reshape(25, 128, 128)
    assert numpy.array_equal(wp.cpu().numpy(), expect_w)
    ws = dev.conv_workspace('cuda') if form.startswith(('cut', 'one')) else None      # a cut launch needs a workspace, and its owner collects
    got = dev.conv5x5s2(_cuda(T, x), wp, _cuda(T, v['encoder/biases_2']), norm,
                        dev.pack_gamma(_cuda(T, v['encoder/gamma_2'])), _cuda(T, v['encoder/beta_2']), workspace=ws).cpu().numpy()
    assert numpy.array_equal(got, ref)
    _assert_handed_over(T, dev, ws)


@pytest.mark.parametrize('form,tile', FORMS)
@pytest.mark.parametrize('shape', [(2, 8, 12), (1, 16, 32), (1, 3, 5), (2, 1, 1), (1, 10, 18)])
@pytest.mark.parametrize('norm', [0, 2])
def test_tconv5x5s2(T, dev, orc, shape, norm, form, tile, launch_options):
    _select_form(launch_options, form, tile)
    v = _vars(5)
    rng = numpy.random.RandomState(6)
    x = rng.standard_normal(size=shape + (128,)).astype(numpy.float32)
    ref = orc.conv2d_transpose_same(x, v['decoder/weights_4'], 2, v['decoder/biases_4'])
    if norm:
        ref = orc.gdn(ref, v['decoder/gamma_5'], v['decoder/beta_5'], inverse=True)
    wp = dev.pack_tconv_weights(_cuda(T, v['decoder/weights_4']))
    perm = dev.packed_channel_order()
    expect_w = numpy.empty((25, 128, 128), dtype=numpy.float32)
    expect_w[:, :, perm] = v['decoder/weights_4'].transpose(0, 1, 3, 2).reshape(25, 128, 128)
    assert numpy.array_equal(wp.cpu().numpy(), expect_w)
    ws = dev.conv_workspace('cuda') if form.startswith(('cut', 'one')) else None
    got = dev.tconv5x5s2(_cuda(T, x), wp, _cuda(T, v['decoder/biases_4']), norm,
                         dev.pack_gamma(_cuda(T, v['decoder/gamma_5'])), _cuda(T, v['decoder/beta_5']), workspace=ws).cpu().numpy()
    assert numpy.array_equal(got, ref)
    _assert_handed_over(T, dev, ws)


@pytest.mark.parametrize('shape', [(2, 16, 24), (1, 4, 16), (1, 5, 7), (3, 1, 1), (1, 12, 40)])
def test_tconv9x9s4_luma(T, dev, orc, shape):
    v = _vars(7)
    rng = numpy.random.RandomState(8)
    # positive weights and inputs so that the reconstruction spans the BT.601 range instead of sitting on the clip floor
    w6 = (numpy.absolute(v['decoder/weights_6'])*numpy.float32(8.)).astype(numpy.float32)
    x = (rng.standard_normal(size=shape + (128,)) + 1.5).astype(numpy.float32)
    ref = orc.conv2d_transpose_same(x, w6, 4, None)[..., 0]
    ref_u8 = numpy.round(ref.clip(min=16., max=235.)).astype(numpy.uint8)
    target = _image(numpy.random.RandomState(9), shape[0], 4*shape[1], 4*shape[2])
    wph = dev.pack_tconv9x9s4_weights(_cuda(T, w6))
    f32, u8, sse = dev.tconv9x9s4_luma(_cuda(T, x), wph, want_f32=True, want_u8=True, ref_u8=_cuda(T, target))
    assert numpy.array_equal(f32.cpu().numpy(), ref)
    assert numpy.array_equal(u8.cpu().numpy(), ref_u8)
    expected = ((target.astype(numpy.int64) - ref_u8.astype(numpy.int64))**2).reshape(shape[0], -1).sum(axis=1)
    assert numpy.array_equal(sse.cpu().numpy(), expected)
    if shape[1]*shape[2] >= 16:
        assert len(numpy.unique(ref_u8)) > 4   # the data exercises the cast, not only the clip floor


@pytest.mark.parametrize('rows', [1, 127, 128, 1000, 32767, 32801])      # below 128 rows per CU: 32-row blocks (a 32-channel tile per wave); from there on 128-row blocks
@pytest.mark.parametrize('inverse', [False, True])
def test_gdn(T, dev, orc, rows, inverse):
    v = _vars(10)
    x = numpy.random.RandomState(11).standard_normal(size=(rows, 128)).astype(numpy.float32)*3
    ref = orc.gdn(x, v['encoder/gamma_3'], v['encoder/beta_3'], inverse=inverse)
    got = dev.gdn(_cuda(T, x), dev.pack_gamma(_cuda(T, v['encoder/gamma_3'])), _cuda(T, v['encoder/beta_3']), inverse=inverse).cpu().numpy()
    assert numpy.array_equal(got, ref)


def test_gdn_reference_known_answer(T, dev):
    """test_tfutils.py:398-423, 493-518: gamma = 0, beta = 4 -> x/2 and x*2."""
    x = numpy.random.RandomState(12).standard_normal(size=(2, 4, 6, 128)).astype(numpy.float32)
    g = numpy.zeros((128, 128), dtype=numpy.float32)
    b = numpy.full(128, 4., dtype=numpy.float32)
    assert numpy.array_equal(dev.gdn(_cuda(T, x), _cuda(T, g), _cuda(T, b)).cpu().numpy(), x/2)
    assert numpy.array_equal(dev.gdn(_cuda(T, x), _cuda(T, g), _cuda(T, b), inverse=True).cpu().numpy(), x*2)


@pytest.mark.parametrize('shape', [(2, 8, 12), (1, 32, 48), (3, 2, 2), (1, 5, 13)])
def test_quantize_maps_and_histograms(T, dev, shape):
    rng = numpy.random.RandomState(13)
    y = (rng.laplace(size=shape + (128,))*3).astype(numpy.float32)
    y[..., 5] = 0.01   # a dead map (after centring by a mean of 0.01 it is exactly 0)
    y[0, 0, 0, 7] = 2.5
    y[0, 0, 1, 7] = -2.5
    y[0, 1, 0, 7] = 3.5   # half-way cases with bw = 1
    bw = rng.uniform(0.5, 2., size=128).astype(numpy.float32)
    bw[7] = 1.
    mean = (rng.standard_normal(size=128)*0.1).astype(numpy.float32)
    mean[5] = numpy.float32(0.01)
    mean[7] = 0.
    res = dev.quantize_maps(_cuda(T, y), _cuda(T, bw), _cuda(T, mean), want_cq=True, want_shifted=True, want_symbols=True,
                            want_flags=True)
    # numpy restatement of reconstructing_eae_kodak.py:178-192 + tools.py:927-929 + compression.py:142
    centered = y - numpy.tile(mean, shape + (1,))
    tiled = numpy.tile(bw.reshape(1, 1, 1, 128), shape + (1,))
    cq = tiled*numpy.round(centered/tiled)
    sym = numpy.round(cq/tiled).astype(numpy.int16)
    assert numpy.array_equal(res['cq'].cpu().numpy(), cq)
    assert numpy.array_equal(res['shifted'].cpu().numpy(), cq + numpy.tile(mean, shape + (1,)))
    planar = numpy.ascontiguousarray(sym.reshape(shape[0], -1, 128).transpose(0, 2, 1))
    assert numpy.array_equal(res['symbols'].cpu().numpy(), planar)
    assert res['checks'].cpu().tolist() == [0, 0, 0] or res['checks'].cpu().tolist()[0] == 0
    dead = numpy.sum(numpy.sum(numpy.absolute(cq), axis=(1, 2)) == 0, axis=1)     # tools.py:318-320
    assert numpy.array_equal((res['nonzero_flags'].cpu().numpy() == 0).sum(axis=1), dead)
    assert dead.min() >= 1
    assert tuple(sym[0, 0, :2, 7]) == (2, -2) and sym[0, 1, 0, 7] == 4            # round half to even
    for radius in (3, 40, 5000):
        hist, overflow = dev.symbol_histograms(res['symbols'], radius)
        hist = hist.cpu().numpy()
        overflow = overflow.cpu().numpy()
        flat = planar.reshape(-1, planar.shape[-1]).astype(numpy.int64)
        for m in range(0, flat.shape[0], 17):
            inside = flat[m][numpy.absolute(flat[m]) <= radius]
            assert numpy.array_equal(hist[m], numpy.bincount(inside + radius, minlength=2*radius + 1))
            assert overflow[m] == flat[m].size - inside.size


def test_quantize_range_error(T, dev):
    y = numpy.zeros((1, 2, 2, 128), dtype=numpy.float32)
    y[0, 0, 0, 3] = 40000.
    y[0, 1, 1, 4] = -32768.
    y[0, 1, 0, 4] = 32767.
    res = dev.quantize_maps(_cuda(T, y), _cuda(T, numpy.ones(128, dtype=numpy.float32)), None, want_symbols=True)
    assert int(res['checks'][0].item()) == 2


def test_cast_bt601_and_sse(T, dev):
    """tools.py:61-93 incl. the reference's own example (test_tools.py:56-71) and half-way cases."""
    x = numpy.array([[15.431, -0.001, 0.], [235.678, 143.18, 1.111]], dtype=numpy.float32)
    assert numpy.array_equal(dev.cast_bt601(_cuda(T, x)).cpu().numpy(), numpy.array([[16, 16, 16], [235, 143, 16]], dtype=numpy.uint8))
    rng = numpy.random.RandomState(14)
    big = numpy.concatenate([rng.uniform(-10, 260, size=100000), numpy.arange(16, 236) + 0.5]).astype(numpy.float32)
    assert numpy.array_equal(dev.cast_bt601(_cuda(T, big)).cpu().numpy(), numpy.round(big.clip(min=16., max=235.)).astype(numpy.uint8))
    a = rng.randint(0, 256, size=(3, 40, 56)).astype(numpy.uint8)
    b = rng.randint(0, 256, size=(3, 40, 56)).astype(numpy.uint8)
    expected = ((a.astype(numpy.int64) - b.astype(numpy.int64))**2).reshape(3, -1).sum(axis=1)
    assert numpy.array_equal(dev.sse_u8(_cuda(T, a), _cuda(T, b)).cpu().numpy(), expected)


@pytest.mark.parametrize('learned', [False, True])
def test_full_chain_bitwise(T, dev, orc, learned):
    """encoder -> quantiser -> decoder through the kernels == the oracle composition, exactly (64x96 image)."""
    v = _vars(20, learned)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    x = _image(numpy.random.RandomState(21), 2, 64, 96)
    y_ref = orc.encoder(x.astype(numpy.float32)[..., None], v, learned)
    from autoencoder_based_image_compression_amd import pipeline
    y = pipeline.DeviceEncoder(v, learned)(_cuda(T, x))
    assert numpy.array_equal(y.cpu().numpy(), y_ref)
    bw = numpy.full(128, 0.5, dtype=numpy.float32)
    q = dev.quantize_maps(y, _cuda(T, bw), None, want_shifted=True)['shifted']
    tiled = numpy.tile(bw.reshape(1, 1, 1, 128), y_ref.shape[:3] + (1,))
    q_ref = tiled*numpy.round(y_ref/tiled)
    assert numpy.array_equal(q.cpu().numpy(), q_ref)
    rec_ref = orc.decoder(q_ref, v, learned)[..., 0]
    (f32, u8, _) = pipeline.DeviceDecoder(v, learned)(q, want_float=True, want_uint8=True)
    assert numpy.array_equal(f32.cpu().numpy(), rec_ref)
    assert numpy.array_equal(u8.cpu().numpy(), numpy.round(rec_ref.clip(min=16., max=235.)).astype(numpy.uint8))


@pytest.mark.parametrize('learned', [False, True])
@pytest.mark.parametrize('size', ['64x96', '256x256'])
def test_transforms_equal_the_committed_fixture(learned, size):
    """tests/golden/transforms_golden.npz (written by the CPU restatement, oracle/gen_transforms_golden.py): the HIP path
    reproduces the latents, the reconstruction and the per-layer checksums without running the oracle."""
    import os
    import zlib
    import torch
    from autoencoder_based_image_compression_amd import device as dev
    from autoencoder_based_image_compression_amd import pipeline
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    v = var.random_variables(1., learned, seed=0, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    tag = '{0}_{1}'.format('learned' if learned else 'fixed', size)
    with numpy.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'transforms_golden.npz')) as g:
        (x, y_ref, q_ref, rec_ref) = (g[tag + '_x'], g[tag + '_y'], g[tag + '_q'], g[tag + '_rec_u8'])
        crc = {name: g[tag + '_sum_crc_' + name] for name in ('gdn_1', 'gdn_2', 'conv_3', 'igdn_3')}
    enc = pipeline.DeviceEncoder(v, learned)
    dec = pipeline.DeviceDecoder(v, learned)
    xd = torch.from_numpy(x).cuda()
    y = enc(xd)
    assert numpy.array_equal(y.cpu().numpy(), y_ref)
    (_, rec, _) = dec(torch.from_numpy(q_ref).cuda())
    assert numpy.array_equal(rec.cpu().numpy(), rec_ref)
    gdn_1 = dev.conv9x9s4_u8(xd, enc.w1, enc.v['encoder/biases_1'], enc.g[1], enc.v['encoder/beta_1'])
    gdn_2 = dev.conv5x5s2(gdn_1, enc.w2, enc.v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], enc.v['encoder/beta_2'])
    conv_3 = dev.conv5x5s2(gdn_2, enc.w3, enc.v['encoder/biases_3'], dev.NORM_NONE)
    for (name, a) in (('gdn_1', gdn_1), ('gdn_2', gdn_2), ('conv_3', conv_3)):
        a = a.cpu().numpy()
        # the float64 sum is blind to the one permitted difference, -0.0 against +0.0; the CRC is not
        assert numpy.sum(a, dtype=numpy.float64) == crc[name][0], (tag, name)
        if float(zlib.crc32(a.tobytes())) != crc[name][1]:
            assert numpy.count_nonzero(a == 0.) > 0, (tag, name)


# ---- the normalisations' mid-range forms (csrc/hip/common.h: sqrt_mid, div_mid, gdn_tile) ------------------
def _mid_forms(mode, first, count, seed=0):
    import ctypes
    from autoencoder_based_image_compression_amd import _native
    import torch
    out = torch.zeros(2, dtype=torch.int64, device='cuda')
    rc = _native.hip().eae_hip_debug_check_mid_forms(mode, first, count, seed, ctypes.c_void_p(out.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    return (int(out[0].item()), int(out[1].item()))


def test_sqrt_mid_equals_sqrtf_on_every_float_of_its_range(test_library):
    """v_sqrt_f32 + the two neighbours + two FMA residuals + two selects (hipcc's own correctly rounded sequence without the scaling
    of tiny inputs and the class check) against sqrtf, EXHAUSTIVELY over everything `gdn_tile`'s guard lets through: every float
    from 2^-96 to +inf and every NaN (fminf skips NaNs, so they reach the short form): 0 results differ. Below 2^-96 and for
    negative denormals -- where the guard sends the wavefront to sqrtf -- the short form does go wrong."""
    two_m96 = 0x0F800000
    (bad, example) = _mid_forms(0, two_m96, (1 << 31) - two_m96)            # 2^-96 .. +inf, positive NaNs
    assert bad == 0, hex(example)
    (bad, example) = _mid_forms(0, 0xFF800001, (1 << 32) - 0xFF800001)      # negative NaNs
    assert bad == 0, hex(example)
    (bad_below, _) = _mid_forms(0, 1, two_m96 - 1)
    (bad_negative, _) = _mid_forms(0, 0x80000001, 0x7F800000)
    print('sqrt_mid outside its range: {0} of {1} positive floats below 2^-96 and {2} negative floats differ from sqrtf'.format(
        bad_below, two_m96 - 1, bad_negative))
    assert bad_below > 0                      # the guard is not decoration


def test_div_mid_equals_the_division_on_its_range(test_library):
    """v_rcp_f32 + one Newton step + quotient + two FMA corrections + the final FMA (hipcc's own correctly rounded division without
    v_div_scale / v_div_fixup) against `/` on 2^33 operand pairs of the guarded range (2^-60 <= |x| <= 2^60, 2^-20 <= s <= 2^40):
    every exponent pair, random mantissas and the extreme ones (0, all ones, equal), both signs."""
    for seed in (1, 2):
        (bad, example) = _mid_forms(1, 0, 1 << 32, seed << 40)
        assert bad == 0, hex(example)


@pytest.mark.parametrize('inverse', [False, True])
def test_normalisation_outside_the_mid_range_takes_the_general_path(T, dev, orc, inverse):
    """Tiles whose operands leave the guarded range (tiny beta, tiny, huge and zero activations) are normalised by sqrtf and `/` as
    before: the conv GEMM epilogue against the oracle on such inputs, next to ordinary ones in the same launch."""
    rng = numpy.random.RandomState(77)
    v = _vars(9)
    x = rng.standard_normal(size=(2, 16, 24, 128)).astype(numpy.float32)
    x[0, :4] *= numpy.float32(1e-25)                                     # x^2 underflows: d + beta = beta
    x[1, 4:8] *= numpy.float32(1e22)                                     # beyond 2^60 after the convolution
    x[1, 8:] = 0.                                                        # exact zeros
    beta = numpy.full(128, 1e-30, dtype=numpy.float32)                   # below 2^-96
    beta[::3] = 1.
    zero_bias = numpy.zeros(128, dtype=numpy.float32)
    if inverse:
        ref = orc.gdn(orc.conv2d_transpose_same(x, v['decoder/weights_4'], 2, zero_bias), v['decoder/gamma_5'], beta, inverse=True)
        got = dev.tconv5x5s2(_cuda(T, x), dev.pack_tconv_weights(_cuda(T, v['decoder/weights_4'])), _cuda(T, zero_bias), 2,
                             dev.pack_gamma(_cuda(T, v['decoder/gamma_5'])), _cuda(T, beta)).cpu().numpy()
    else:
        ref = orc.gdn(orc.conv2d_same(x, v['encoder/weights_2'], 2, zero_bias), v['encoder/gamma_2'], beta)
        got = dev.conv5x5s2(_cuda(T, x), dev.pack_conv_weights(_cuda(T, v['encoder/weights_2'])), _cuda(T, zero_bias), 1,
                            dev.pack_gamma(_cuda(T, v['encoder/gamma_2'])), _cuda(T, beta)).cpu().numpy()
    both_nan = numpy.isnan(got) & numpy.isnan(ref)
    assert numpy.array_equal(got[~both_nan], ref[~both_nan])
    assert numpy.isfinite(ref).sum() > ref.size//2
