"""Pins oracle/transforms_oracle.c: (1) against what the reference's own tests pin at the TF boundary (GDN/IGDN with
gamma = 0, beta = 4: test_tfutils.py:398-423, 493-518; the x16 shape law: test_eae.py:71-139, 178-247; zero latents
decode to a constant image: test_eae.py:141-176); (2) against an INDEPENDENT evaluation of TensorFlow's published op
definitions in float64 (torch on CPU: padded cross-correlation; conv2d_transpose as the autograd gradient of the
forward conv). Tolerance for (2): |oracle - float64| <= 1e-6 * (|x| conv |w|) + 1e-6 elementwise, i.e. a few float32
roundings relative to the sum of the magnitudes of the K <= 3200 products (a float32 FMA chain measures
0.75-1.5e-7 * sum|a*b|); TF itself is not installable here, its conv OUTPUT VALUES are 'parity unpinned'."""
import numpy
import pytest
import torch
import torch.nn.functional as F

from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
from oracle import transforms as T

REL = 1e-6
ABS = 1e-6


def _nchw64(x):
    return torch.from_numpy(x).permute(0, 3, 1, 2).double()


@pytest.mark.parametrize('k,s,cin,shape', [(9, 4, 1, (2, 32, 48)), (5, 2, 128, (1, 16, 24)), (5, 2, 128, (2, 6, 10))])
def test_conv2d_same_against_float64_definition(k, s, cin, shape):
    rng = numpy.random.RandomState(k + s)
    x = rng.standard_normal(size=shape + (cin,)).astype(numpy.float32)
    w = (rng.standard_normal(size=(k, k, cin, 128))*0.05).astype(numpy.float32)
    b = (rng.standard_normal(size=128)*0.1).astype(numpy.float32)
    got = T.conv2d_same(x, w, s, b)
    (h, wd) = shape[1:]
    pad_h = max((-(-h//s) - 1)*s + k - h, 0)
    pad_w = max((-(-wd//s) - 1)*s + k - wd, 0)
    ref = F.conv2d(F.pad(_nchw64(x), (pad_w//2, pad_w - pad_w//2, pad_h//2, pad_h - pad_h//2)),
                   torch.from_numpy(w).permute(3, 2, 0, 1).double(), stride=s) + torch.from_numpy(b).double().view(1, -1, 1, 1)
    bound = F.conv2d(F.pad(_nchw64(numpy.abs(x)), (pad_w//2, pad_w - pad_w//2, pad_h//2, pad_h - pad_h//2)),
                     torch.from_numpy(numpy.abs(w)).permute(3, 2, 0, 1).double(), stride=s).permute(0, 2, 3, 1).numpy()
    assert got.shape == (shape[0], h//s, wd//s, 128)
    assert numpy.all(numpy.abs(got - ref.permute(0, 2, 3, 1).numpy()) <= REL*bound + ABS)


@pytest.mark.parametrize('col2im', [False, True])
@pytest.mark.parametrize('k,s,cout,shape', [(5, 2, 128, (2, 8, 12)), (9, 4, 1, (1, 8, 12)), (5, 2, 128, (1, 3, 5)), (9, 4, 1, (2, 1, 1)),
                                            (9, 4, 1, (1, 3, 17))])
def test_conv2d_transpose_is_the_gradient_of_the_forward_conv(k, s, cout, shape, col2im):
    """Both summation orders of the restatement (one chain per output element; GEMM + col2im, the order of transpose_conv_3)
    against the float64 definition."""
    rng = numpy.random.RandomState(k*s)
    x = rng.standard_normal(size=shape + (128,)).astype(numpy.float32)
    w = (rng.standard_normal(size=(k, k, cout, 128))*0.05).astype(numpy.float32)      # [k, k, out, in]
    got = T.conv2d_transpose_same(x, w, s, None, col2im=col2im)
    (h, wd) = (shape[1]*s, shape[2]*s)
    inp = torch.zeros(shape[0], cout, h, wd, dtype=torch.double, requires_grad=True)
    pad = k - s   # (out-1)*s + k - s*out
    fw = F.conv2d(F.pad(inp, (pad//2, pad - pad//2, pad//2, pad - pad//2)), torch.from_numpy(w).permute(3, 2, 0, 1).double(), stride=s)
    (grad,) = torch.autograd.grad(fw, inp, grad_outputs=_nchw64(x))
    inp2 = torch.zeros(shape[0], cout, h, wd, dtype=torch.double, requires_grad=True)
    fw2 = F.conv2d(F.pad(inp2, (pad//2, pad - pad//2, pad//2, pad - pad//2)), torch.from_numpy(numpy.abs(w)).permute(3, 2, 0, 1).double(), stride=s)
    (bound,) = torch.autograd.grad(fw2, inp2, grad_outputs=_nchw64(numpy.abs(x)))
    assert got.shape == (shape[0], h, wd, cout)
    assert numpy.all(numpy.abs(got - grad.permute(0, 2, 3, 1).numpy()) <= REL*bound.permute(0, 2, 3, 1).numpy() + ABS)


def test_gdn_reference_known_answers():
    """test_tfutils.py:398-423 / 493-518: gamma = 0, beta = 4 gives x/2 and 2x exactly."""
    x = numpy.random.RandomState(0).standard_normal(size=(2, 4, 6, 128)).astype(numpy.float32)
    g = numpy.zeros((128, 128), dtype=numpy.float32)
    b = numpy.full(128, 4., dtype=numpy.float32)
    assert numpy.array_equal(T.gdn(x, g, b), x/2)
    assert numpy.array_equal(T.gdn(x, g, b, inverse=True), x*2)


def test_gdn_against_float64_definition():
    rng = numpy.random.RandomState(1)
    v = var.random_variables(1., False, seed=2)
    x = (rng.standard_normal(size=(50, 128))*4).astype(numpy.float32)
    (g, b) = (v['encoder/gamma_1'], v['encoder/beta_1'])
    d = numpy.sqrt((x.astype(numpy.float64)**2) @ g.astype(numpy.float64) + b.astype(numpy.float64))
    assert numpy.abs(T.gdn(x, g, b) - x/d).max() < 1e-6
    assert numpy.abs(T.gdn(x, g, b, inverse=True) - x*d).max() < 1e-4
    # IGDN(GDN(x)) is close to x only for small gamma; what must hold exactly is GDN's definition above


def test_shape_law_times_16_and_zero_latents_give_a_constant_image():
    """test_eae.py:71-139, 178-247 (shapes) and 141-176 (all-zero latents decode to a constant image)."""
    for learned in (False, True):
        v = var.random_variables(1., learned, seed=4, bias_std=0.01)
        x = numpy.random.RandomState(5).randint(0, 256, size=(2, 32, 48, 1)).astype(numpy.float32)
        y = T.encoder(x, v, learned)
        assert y.shape == (2, 2, 3, 128)
        assert T.decoder(y, v, learned).shape == (2, 32, 48, 1)
        rec = T.decoder(numpy.zeros((1, 8, 10, 128), dtype=numpy.float32), v, learned)
        assert rec.shape == (1, 128, 160, 1)
        # zero latents -> layer 1 is the bias everywhere; away from the borders (zero padding of the two later layers
        # reaches < 32 pixels) every 16x16 block sees the same taps: the image is periodic there, i.e. "constant"
        # up to the sub-pixel phase pattern of the transposed convolutions.
        interior = rec[0, 32:96, 32:128, 0]
        assert numpy.array_equal(interior[:, :16], interior[:, 16:32])
        assert numpy.array_equal(interior[:16, :], interior[16:32, :])
        assert numpy.array_equal(interior[:48, :80], interior[16:, 16:])


def test_oracle_order_is_one_fma_chain():
    """The documented accumulation order: 32-channel block (outer), kernel row, kernel column, channel inside the block;
    one float32 FMA chain from +0, bias after."""
    rng = numpy.random.RandomState(6)
    x = rng.standard_normal(size=(1, 4, 4, 128)).astype(numpy.float32)
    w = (rng.standard_normal(size=(5, 5, 128, 128))*0.05).astype(numpy.float32)
    b = rng.standard_normal(size=128).astype(numpy.float32)
    got = T.conv2d_same(x, w, 2, b)
    import math
    (i, j, co) = (1, 0, 7)
    acc = numpy.float32(0.)
    for c0 in range(0, 128, 32):
        for u in range(5):
            for v in range(5):
                (r, c) = (2*i + u - 1, 2*j + v - 1)
                if not (0 <= r < 4 and 0 <= c < 4):
                    continue
                for ci in range(c0, c0 + 32):
                    # float64 product of two float32 is exact; one rounding on the sum == fmaf
                    acc = numpy.float32(numpy.float64(x[0, r, c, ci])*numpy.float64(w[u, v, ci, co]) + numpy.float64(acc))
    assert got[0, i, j, co] == numpy.float32(acc + b[co])
    assert math.isfinite(float(acc))


def test_oracle_reproduces_the_committed_transform_fixture():
    """tests/golden/transforms_golden.npz (oracle/gen_transforms_golden.py) freezes the restatement:
    same latents, quantised latents, reconstruction and per-layer checksums on this host as when it was generated."""
    import os
    import zlib
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    from oracle import transforms as T
    with numpy.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'transforms_golden.npz')) as g:
        for learned in (False, True):
            v = var.random_variables(1., learned, seed=0, bias_std=0.01)
            v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
            tag = '{}_64x96'.format('learned' if learned else 'fixed')
            x = g[tag + '_x']
            (y, enc) = T.encoder(x.astype(numpy.float32)[..., None], v, learned, return_intermediates=True)
            assert numpy.array_equal(y, g[tag + '_y'])
            (rec, dec) = T.decoder(numpy.round(y), v, learned, return_intermediates=True)
            assert numpy.array_equal(numpy.round(rec[..., 0].clip(min=16., max=235.)).astype(numpy.uint8), g[tag + '_rec_u8'])
            for (name, a) in (('gdn_1', enc['gdn_1']), ('conv_3', enc['conv_3']), ('igdn_3', dec['igdn_3'])):
                a = numpy.ascontiguousarray(a, dtype=numpy.float32)
                assert float(zlib.crc32(a.tobytes())) == g[tag + '_sum_crc_' + name][1], (tag, name)


@pytest.mark.parametrize('learned', [False, True])
def test_the_torch_cpu_stand_in_of_the_cpu_baseline_follows_the_oracle(learned):
    """oracle/transforms_torch.py (what `bench.py: cpu_baseline` times as the stand-in for TF-CPU's kernels) computes the same
    graph as the C oracle: equal to float32 rounding of a different summation order (never used as a checker)."""
    from oracle import transforms as orc, transforms_torch
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    v = var.random_variables(1., learned, seed=21, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    x = numpy.random.RandomState(22).randint(16, 236, size=(2, 48, 80, 1)).astype(numpy.float32)
    cpu = transforms_torch.CpuTransforms(v, learned)
    y_ref = orc.encoder(x, v, learned)
    y = cpu.encoder(x)
    assert y.shape == y_ref.shape and numpy.allclose(y, y_ref, rtol=1e-4, atol=1e-4*float(numpy.abs(y_ref).max()))
    q = numpy.round(y_ref)
    rec_ref = orc.decoder(q, v, learned)
    rec = cpu.decoder(q)
    assert rec.shape == rec_ref.shape and numpy.allclose(rec, rec_ref, rtol=1e-4, atol=1e-4*float(numpy.abs(rec_ref).max()))
