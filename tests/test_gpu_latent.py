"""eae_hip_latent_stage (gdn_3 -> quantiser -> inverse_gdn_4 in one kernel) against the separate kernels it fuses
(eae_hip_gdn, eae_hip_quantize_maps, eae_hip_gdn), which are themselves pinned to the CPU oracle: identical bits."""
import numpy
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape', [(2, 32, 48), (3, 5, 7), (1, 1, 1), (5, 16, 16), (40, 1, 2)])
@pytest.mark.parametrize('learned', [False, True])
@pytest.mark.parametrize('form', ['quarter', 'wave', 'lds'])
def test_latent_stage_equals_the_separate_kernels(shape, learned, form, launch_options):
    """form (EAE_HIP_LATENT, csrc/hip/latent.hip): four waves per 32-position tile, one 32-channel tile each (the default);
    one register-resident wave per tile; the block-cooperative LDS kernel. (3, 5, 7): tiles that straddle images."""
    from autoencoder_based_image_compression_amd import device as dev
    launch_options.delenv('EAE_HIP_LATENT_LDS', raising=False)
    launch_options.setenv('EAE_HIP_LATENT', form[0])
    rng = numpy.random.RandomState(shape[1]*7 + int(learned))
    (n, h, w) = shape
    x = torch.from_numpy((rng.laplace(size=(n, h, w, 128))*rng.uniform(0.1, 6., size=128)).astype(numpy.float32)).cuda()
    x[0, 0, 0, 5] = 1.e6                         # one symbol outside int16: checks[0]
    x[:, :, :, 9] = 1e-3                         # a dead map after quantisation
    gamma = rng.uniform(2e-5, 0.01, size=(128, 128)).astype(numpy.float32)
    gamma = torch.from_numpy(0.5*(gamma + gamma.T)).cuda()
    (g3, g4) = (dev.pack_gamma(gamma), dev.pack_gamma(gamma*2.))
    (b3, b4) = (torch.from_numpy(rng.uniform(0.5, 2., size=128).astype(numpy.float32)).cuda(),
                torch.from_numpy(rng.uniform(0.5, 2., size=128).astype(numpy.float32)).cuda())
    bw = torch.from_numpy(rng.uniform(0.4, 2., size=128).astype(numpy.float32)).cuda()
    mean = torch.from_numpy(rng.normal(scale=0.2, size=128).astype(numpy.float32)).cuda()
    # the separate kernels
    y = x if learned else dev.gdn(x, g3, b3, inverse=False)
    q = dev.quantize_maps(y, bw, mean, want_shifted=True, want_symbols=True, want_flags=True)
    t = q['shifted'] if learned else dev.gdn(q['shifted'], g4, b4, inverse=True)
    # the fused kernel
    f = dev.latent_stage(x, bw, mean, gdn_in=None if learned else (g3, b3), igdn_out=None if learned else (g4, b4),
                         want_y=True, want_shifted=True, want_flags=True)
    assert torch.equal(f['y'], y)
    assert torch.equal(f['shifted'], q['shifted'])
    assert torch.equal(f['symbols'], q['symbols'])
    assert torch.equal(f['nonzero_flags'], q['nonzero_flags']) and int(f['nonzero_flags'][:, 9].sum()) == 0
    assert torch.equal(f['checks'], q['checks'])
    if learned:
        assert int(f['checks'][0]) >= 1          # 40000 / bw is outside int16 (gdn_3 would squash it)
    if learned:
        assert f['t'] is None
    else:
        assert torch.equal(f['t'], t)


def test_latent_stage_argument_checks():
    from autoencoder_based_image_compression_amd import _native
    lib = _native.hip()
    x = torch.zeros((1, 2, 2, 128), device='cuda')
    bw = torch.ones(128, device='cuda')
    assert lib.eae_hip_latent_stage(None, None, None, None, bw.data_ptr(), None, None, None, None, None, None, None, None, 1, 4, None) == -1
    # gamma without beta
    assert lib.eae_hip_latent_stage(x.data_ptr(), bw.data_ptr(), None, None, bw.data_ptr(), None, None, None, None, None, None, None, None,
                                    1, 4, None) == -1


@pytest.mark.parametrize('learned', [False, True])
@pytest.mark.parametrize('shape,form', [((2, 16, 24), ''), ((1, 6, 10), ''), ((2, 16, 24), 'u'), ((2, 16, 24), 's'), ((1, 6, 10), 's'),
                                        ((6, 128, 192), ''), ((6, 128, 192), 'u'), ((24, 64, 96), '')])
def test_conv3_with_the_latent_stage_as_its_epilogue(shape, form, learned, launch_options):
    """eae_hip_conv5x5s2_latent == eae_hip_conv5x5s2 (no normalisation) followed by eae_hip_latent_stage, every output, bit for
    bit: small layers (two launches inside the entry point), the fused kernel with whole tiles ('u'), with every tile cut
    ('s') and as the launch decides ('' : 6 x 128x192 = 1152 tiles is cut, 24 x 64x96 = Kodak batch)."""
    from autoencoder_based_image_compression_amd import device as dev
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    launch_options.clear()
    v = var.random_variables(1., learned, seed=61, bias_std=0.01)
    rng = numpy.random.RandomState(62 + shape[1])
    x = torch.from_numpy(rng.standard_normal(size=shape + (128,)).astype(numpy.float32)).cuda()
    w3 = dev.pack_conv_weights(torch.from_numpy(v['encoder/weights_3']).cuda())
    b3 = torch.from_numpy(v['encoder/biases_3']).cuda()
    bw = torch.from_numpy(rng.uniform(0.05, 0.5, size=128).astype(numpy.float32)).cuda()
    mean = torch.from_numpy(rng.normal(scale=0.05, size=128).astype(numpy.float32)).cuda()
    gdn_in = igdn_out = None
    if not learned:
        gdn_in = (dev.pack_gamma(torch.from_numpy(v['encoder/gamma_3']).cuda()), torch.from_numpy(v['encoder/beta_3']).cuda())
        igdn_out = (dev.pack_gamma(torch.from_numpy(v['decoder/gamma_4']).cuda()), torch.from_numpy(v['decoder/beta_4']).cuda())
    # the two separate launches (the convolution in its one-tile-per-wave form)
    launch_options.setenv('EAE_HIP_GEMM', 'w')
    raw = dev.conv5x5s2(x, w3, b3, dev.NORM_NONE, workspace=False)
    launch_options.delenv('EAE_HIP_GEMM')
    ref = dev.latent_stage(raw, bw, mean, gdn_in=gdn_in, igdn_out=igdn_out, want_y=True, want_shifted=True, want_flags=True)
    if form:
        launch_options.setenv('EAE_HIP_GEMM', form)
    ws = dev.conv_workspace('cuda')
    # small layers run the stage IN PLACE on the convolution's output inside the entry point: every form of the stage kernel
    # ('' = four waves per tile, 'w' = one wave per tile, 'l' = block-cooperative LDS form) must cope with x == its output
    for stage_form in ('', 'w', 'l') if shape[1] <= 16 else ('',):
        if stage_form:
            launch_options.setenv('EAE_HIP_LATENT', stage_form)
        for _ in range(2):
            got = dev.conv5x5s2_latent(x, w3, b3, bw, mean, gdn_in=gdn_in, igdn_out=igdn_out, want_y=True, want_shifted=True,
                                       want_flags=True, workspace=ws)
            for key in ('y', 'shifted', 'symbols', 'nonzero_flags', 'checks') + (() if learned else ('t',)):
                assert torch.equal(got[key].reshape(ref[key].shape), ref[key]), (key, stage_form)
            assert int(torch.count_nonzero(ws).item()) == 0
        launch_options.delenv('EAE_HIP_LATENT', raising=False)
    assert int(ref['symbols'].abs().max().item()) > 2            # the quantiser is exercised, not a field of zeros
