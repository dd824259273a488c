"""conv_gemm_split_kernel (csrc/hip/conv_gemm_split.hip): tiles interrupted at a K-step boundary and finished by another
wave must give the bits of the uninterrupted chain. Small shapes with the cut forced are in tests/test_gpu_kernels.py
(every tile cut, tails waiting on their heads); here: shapes large enough for the launch to decide by itself, against the
CPU oracle and against the other kernels, the workspace contract (zero again after every launch, no timeout word), reuse
of one workspace, and concurrent launches on two streams."""
import numpy
import pytest
import torch

pytestmark = pytest.mark.gpu



def _vars(seed):
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables
    return variables.random_variables(1., False, seed=seed, bias_std=0.01)


def _workspace_is_clean(ws):
    return int(torch.count_nonzero(ws).item()) == 0


def _forms(launch_options, call, ws, reference):
    """The same launch in every other form: bits equal `reference`, workspace left zeroed."""
    for (form, waves) in (('u', None), ('w', None), ('s', '1'), ('s', '2'), ('s', '3')):
        launch_options.setenv('EAE_HIP_GEMM', form)
        if waves:
            launch_options.setenv('EAE_HIP_SPLIT_WAVES', waves)
        assert torch.equal(call(ws), reference), (form, waves)
        assert _workspace_is_clean(ws)
    launch_options.clear()


@pytest.mark.parametrize('norm', [0, 1])
def test_conv_against_the_oracle(norm, launch_options):
    """6 x 128x192 inputs = 1152 tiles of 32 positions on 1024 SIMDs: the launch cuts its last tiles by itself."""
    from autoencoder_based_image_compression_amd import device as dev
    from oracle import transforms as orc
    launch_options.clear()
    v = _vars(31)
    x = numpy.random.RandomState(32).standard_normal(size=(6, 128, 192, 128)).astype(numpy.float32)
    ref = orc.conv2d_same(x, v['encoder/weights_2'], 2, v['encoder/biases_2'])
    if norm:
        ref = orc.gdn(ref, v['encoder/gamma_2'], v['encoder/beta_2'])
    ws = dev.conv_workspace('cuda')
    args = (torch.from_numpy(x).cuda(), dev.pack_conv_weights(torch.from_numpy(v['encoder/weights_2']).cuda()),
            torch.from_numpy(v['encoder/biases_2']).cuda(), norm, dev.pack_gamma(torch.from_numpy(v['encoder/gamma_2']).cuda()),
            torch.from_numpy(v['encoder/beta_2']).cuda())
    got = dev.conv5x5s2(*args, workspace=ws)
    assert numpy.array_equal(got.cpu().numpy(), ref)
    assert _workspace_is_clean(ws)
    assert torch.equal(dev.conv5x5s2(*args, workspace=False), got)          # no workspace: whole tiles
    _forms(launch_options, lambda w: dev.conv5x5s2(*args, workspace=w), ws, got)


def test_the_cut_launches_are_for_one_whole_mi355x(launch_options):
    """`blockIdx.x & 7` is the XCD and an XCD holds CUs / 8 x 4 x k waves only when the logical device is one whole MI355X (compute
    partition SPX: 256 CUs, 8 XCDs). The library reads the CU count of the logical device; on anything else (here: pretended with
    EAE_HIP_ASSUME_PARTITIONED=1) a launch that would cut its last tiles keeps them whole -- the workspace is never touched -- and
    gives the same bits."""
    from autoencoder_based_image_compression_amd import device as dev
    launch_options.clear()
    info = dev.partition_info()
    assert info == {'compute_units': 256, 'xcds': 8, 'whole_device': True}, info      # the GPU boxes of this pool run SPX
    v = _vars(35)
    x = torch.from_numpy(numpy.random.RandomState(36).standard_normal(size=(6, 128, 192, 128)).astype(numpy.float32)).cuda()
    args = (x, dev.pack_conv_weights(torch.from_numpy(v['encoder/weights_2']).cuda()), torch.from_numpy(v['encoder/biases_2']).cuda(), 1,
            dev.pack_gamma(torch.from_numpy(v['encoder/gamma_2']).cuda()), torch.from_numpy(v['encoder/beta_2']).cuda())
    ws = dev.conv_workspace('cuda')
    cut = dev.conv5x5s2(*args, workspace=ws)
    launch_options.setenv('EAE_HIP_ASSUME_PARTITIONED', '1')
    assert dev.partition_info() == {'compute_units': 256, 'xcds': 8, 'whole_device': False}
    poisoned = torch.full_like(ws, 0x5a5a5a5a)          # a cut launch would read flags out of this and time out or go wrong
    whole = dev.conv5x5s2(*args, workspace=poisoned)
    assert torch.equal(whole, cut) and bool((poisoned == 0x5a5a5a5a).all())
    launch_options.clear()
    assert dev.partition_info()['whole_device']


@pytest.mark.parametrize('norm', [0, 2])
def test_tconv_against_the_oracle(norm, launch_options):
    """3 x 48x72 sites x 4 output phases = 1296 tiles of four lengths (36 / 24 / 24 / 16 K-steps)."""
    from autoencoder_based_image_compression_amd import device as dev
    from oracle import transforms as orc
    launch_options.clear()
    v = _vars(33)
    x = numpy.random.RandomState(34).standard_normal(size=(3, 48, 72, 128)).astype(numpy.float32)
    ref = orc.conv2d_transpose_same(x, v['decoder/weights_4'], 2, v['decoder/biases_4'])
    if norm:
        ref = orc.gdn(ref, v['decoder/gamma_5'], v['decoder/beta_5'], inverse=True)
    ws = dev.conv_workspace('cuda')
    args = (torch.from_numpy(x).cuda(), dev.pack_tconv_weights(torch.from_numpy(v['decoder/weights_4']).cuda()),
            torch.from_numpy(v['decoder/biases_4']).cuda(), norm, dev.pack_gamma(torch.from_numpy(v['decoder/gamma_5']).cuda()),
            torch.from_numpy(v['decoder/beta_5']).cuda())
    got = dev.tconv5x5s2(*args, workspace=ws)
    assert numpy.array_equal(got.cpu().numpy(), ref)
    assert _workspace_is_clean(ws)
    assert torch.equal(dev.tconv5x5s2(*args, workspace=False), got)
    _forms(launch_options, lambda w: dev.tconv5x5s2(*args, workspace=w), ws, got)


def test_kodak_batch_layers_in_every_form(launch_options):
    """The four launches of the benchmark's step (24 x 512x768), GPU against GPU: ragged shares (4608 tiles / 8 XCDs), one
    workspace reused by all four launches, then the same again on two streams at once with a workspace each."""
    import bench
    from autoencoder_based_image_compression_amd import device as dev, pipeline
    launch_options.clear()
    variables = bench.synthetic_model(1.)
    enc = pipeline.DeviceEncoder(variables, False)
    dec = pipeline.DeviceDecoder(variables, False)
    (v, d) = (enc.v, dec.v)
    images = torch.from_numpy(bench.synthetic_images(5, 24, 512, 768)).cuda()
    gdn_1 = dev.conv9x9s4_u8(images, enc.w1, v['encoder/biases_1'], enc.g[1], v['encoder/beta_1'])

    def chain(workspace):
        gdn_2 = dev.conv5x5s2(gdn_1, enc.w2, v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], v['encoder/beta_2'], workspace=workspace)
        conv_3 = dev.conv5x5s2(gdn_2, enc.w3, v['encoder/biases_3'], dev.NORM_NONE, workspace=workspace)
        t1 = dev.tconv5x5s2(conv_3, dec.w4, d['decoder/biases_4'], dev.NORM_IGDN, dec.g[5], d['decoder/beta_5'], workspace=workspace)
        t2 = dev.tconv5x5s2(t1, dec.w5, d['decoder/biases_5'], dev.NORM_IGDN, dec.g[6], d['decoder/beta_6'], workspace=workspace)
        return (gdn_2, conv_3, t1, t2)

    launch_options.setenv('EAE_HIP_GEMM', 'w')
    plain = chain(False)                         # conv_gemm_wave_kernel, pinned to the oracle at this size by test_gpu_full_size
    launch_options.delenv('EAE_HIP_GEMM')
    ws = dev.conv_workspace('cuda')
    for _ in range(3):
        for (a, b) in zip(plain, chain(ws)):
            assert torch.equal(a, b)
        assert _workspace_is_clean(ws)
    launch_options.setenv('EAE_HIP_GEMM', 's')      # every layer cut, the transposed convolutions too
    for (a, b) in zip(plain, chain(ws)):
        assert torch.equal(a, b)
    assert _workspace_is_clean(ws)
    launch_options.delenv('EAE_HIP_GEMM')
    # two chains at once: their waves compete for the SIMDs, every tile is still one uninterrupted-equivalent chain
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    spaces = [dev.conv_workspace('cuda') for _ in streams]
    torch.cuda.synchronize()
    results = []
    for (stream, space) in zip(streams, spaces):
        with torch.cuda.stream(stream):
            results.append(chain(space))
    torch.cuda.synchronize()
    for queued in results:
        for (a, b) in zip(plain, queued):
            assert torch.equal(a, b)
    assert all(_workspace_is_clean(space) for space in spaces)


def test_workspace_entry_points_check_their_arguments(launch_options):
    from autoencoder_based_image_compression_amd import _native
    lib = _native.hip()
    assert int(lib.eae_hip_conv_workspace_bytes()) >= 4*(256 + 8*128)
    x = torch.zeros((1, 4, 4, 128), device='cuda')
    out = torch.zeros((1, 2, 2, 128), device='cuda')
    w = torch.zeros((25, 128, 128), device='cuda')
    assert lib.eae_hip_conv5x5s2_ws(x.data_ptr(), w.data_ptr(), None, 0, None, None, out.data_ptr(), 1, 4, 4, None, None) == -1
    assert lib.eae_hip_tconv5x5s2_ws(x.data_ptr(), w.data_ptr(), None, 0, None, None, out.data_ptr(), 1, 4, 4, None, None) == -1
    ws = torch.zeros(int(lib.eae_hip_conv_workspace_bytes())//4, dtype=torch.int32, device='cuda')
    assert lib.eae_hip_conv5x5s2_ws(x.data_ptr(), w.data_ptr(), None, 1, None, None, out.data_ptr(), 1, 4, 4, ws.data_ptr(), None) == -1
    assert lib.eae_hip_conv5x5s2_ws(x.data_ptr(), w.data_ptr(), None, 0, None, None, out.data_ptr(), 1, 3, 4, ws.data_ptr(), None) == -2
    launch_options.setenv('EAE_HIP_GEMM', 's')      # a forced cut needs the workspace
    assert lib.eae_hip_conv5x5s2(x.data_ptr(), w.data_ptr(), None, 0, None, None, out.data_ptr(), 1, 4, 4, None) == -1


def _collect(dev, ws):
    word = torch.zeros(1, dtype=torch.int32, device='cuda')
    dev.conv_workspace_collect(ws, word)
    return int(word.item())


@pytest.mark.parametrize('forced', [False, True])
def test_a_hand_off_that_never_happens_is_loud_and_leaves_no_trace(forced, launch_options):
    """eae_hip_debug_set_split_mute(1) (fault injection, a debug entry point: no environment variable reaches it): the heads of the cut tiles park their
    accumulators but never publish. Every tail must give up (after ~1 ms with the hook), write NOTHING into its tile, and be
    counted in the workspace's error word; eae_hip_conv_workspace_collect hands the count over and zeroes the workspace, so
    that the next launch -- hook off -- gives the oracle's bits with the same workspace."""
    from autoencoder_based_image_compression_amd import device as dev
    launch_options.clear()
    v = _vars(41)
    x = torch.from_numpy(numpy.random.RandomState(42).standard_normal(size=(6, 128, 192, 128)).astype(numpy.float32)).cuda()
    args = (x, dev.pack_conv_weights(torch.from_numpy(v['encoder/weights_2']).cuda()), torch.from_numpy(v['encoder/biases_2']).cuda(),
            1, dev.pack_gamma(torch.from_numpy(v['encoder/gamma_2']).cuda()), torch.from_numpy(v['encoder/beta_2']).cuda())
    good = dev.conv5x5s2(*args, workspace=False)
    ws = dev.conv_workspace('cuda')
    if forced:
        launch_options.setenv('EAE_HIP_GEMM', 's')      # every XCD share cut as deep as it goes
    launch_options.split_mute(True)
    sentinel = torch.full_like(good, 12345.0)
    out = dev.conv5x5s2(*args, out=sentinel.clone(), workspace=ws)
    torch.cuda.synchronize()
    unfinished = _collect(dev, ws)
    assert unfinished > 0
    assert _workspace_is_clean(ws)
    # pixels are either finished tiles (the oracle's bits), or untouched / parked accumulators of an abandoned tile: at
    # least `unfinished` tiles of 32 positions differ, and no abandoned tile carries a finished result
    differs = (out != good).any(dim=-1)
    assert int(differs.sum().item()) > 0
    launch_options.split_mute(False)
    again = dev.conv5x5s2(*args, workspace=ws)       # the same workspace, no memset by the caller
    assert torch.equal(again, good)
    assert _collect(dev, ws) == 0 and _workspace_is_clean(ws)


def test_the_whole_path_entry_points_report_the_failure(launch_options):
    """eae_hip_encode with the hook on: eae_hip_transform_status (C ABI) and device.Model.check() (what the reference-shaped
    `sess.run` nodes call after their copy to the host) both report it; the next call on the same model is clean."""
    import ctypes
    from autoencoder_based_image_compression_amd import _native, device as dev, pipeline
    launch_options.clear()
    v = _vars(43)
    images = torch.from_numpy(numpy.random.RandomState(44).randint(16, 236, size=(2, 128, 192)).astype(numpy.uint8)).cuda()
    enc = pipeline.DeviceEncoder(v, False)
    good = enc(images)
    enc.check()
    launch_options.setenv('EAE_HIP_GEMM', 's')
    launch_options.split_mute(True)
    enc(images)
    with pytest.raises(dev.SplitHandOffTimeout):
        enc.check()
    # straight through the C ABI
    lib = _native.hip()
    nbytes = int(lib.eae_hip_encode_scratch_bytes(2, 128, 192))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    latents = torch.empty_like(good)
    assert lib.eae_hip_encode(enc.model._handle, images.data_ptr(), 2, 128, 192, latents.data_ptr(), scratch.data_ptr(), nbytes, None) == 0
    count = ctypes.c_uint32(0)
    assert lib.eae_hip_transform_status(scratch.data_ptr(), ctypes.byref(count), None) == 0 and count.value > 0
    launch_options.split_mute(False)
    assert lib.eae_hip_encode(enc.model._handle, images.data_ptr(), 2, 128, 192, latents.data_ptr(), scratch.data_ptr(), nbytes, None) == 0
    assert lib.eae_hip_transform_status(scratch.data_ptr(), ctypes.byref(count), None) == 0 and count.value == 0
    assert torch.equal(latents, good)
    assert torch.equal(enc(images), good)
    enc.check()
