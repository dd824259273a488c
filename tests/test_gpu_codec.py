"""`codec.BatchCodec` (the fused, asynchronous hot path that bench.py times) against the reference-shaped functions of
`kodak/` called image by image like `fix_gamma` does (reconstructing_eae_kodak.py:170-225): bits of the lossless code,
squared error / PSNR, dead maps and reconstruction are identical."""
import os

import numpy
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'coder_golden.npz')


def reference_shaped_path(variables, learned, images, bin_widths, map_mean, path_probabilities, idx_map_exception):
    from autoencoder_based_image_compression_amd import pipeline
    from autoencoder_based_image_compression_amd.kodak.lossless import compression
    from autoencoder_based_image_compression_amd.kodak.tools import tools as tls
    encoder = pipeline.DeviceEncoder(variables, learned)
    decoder = pipeline.DeviceDecoder(variables, learned)
    y = encoder(torch.from_numpy(images).cuda()).cpu().numpy()
    centered = y - numpy.tile(map_mean, y.shape[:3] + (1,))
    cq = tls.quantize_per_map(centered, bin_widths)
    nb_bits = numpy.array([compression.rescale_compress_lossless_maps(cq[j], bin_widths, path_probabilities, idx_map_exception)
                           for j in range(images.shape[0])], dtype=numpy.int64)
    nb_deads = tls.count_nb_deads(cq)
    shifted = cq + numpy.tile(map_mean, y.shape[:3] + (1,))
    (_, rec, _) = decoder(torch.from_numpy(shifted).cuda())
    rec = rec.cpu().numpy()
    psnr = numpy.array([tls.psnr_2d(images[j], rec[j]) for j in range(images.shape[0])])
    return nb_bits, nb_deads, rec, psnr


@pytest.mark.parametrize('learned', [False, True])
@pytest.mark.parametrize('idx_map_exception', [67, -1])
def test_batch_codec_equals_the_image_by_image_path(tmp_path, learned, idx_map_exception):
    from autoencoder_based_image_compression_amd import codec
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    from autoencoder_based_image_compression_amd.kodak.tools import tools as tls
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    path = str(tmp_path/'binary_probabilities.npy')
    numpy.save(path, probabilities)
    rng = numpy.random.RandomState(17 + int(learned))
    v = var.random_variables(1., learned, seed=5, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    images = rng.randint(16, 236, size=(6, 64, 96)).astype(numpy.uint8)
    bin_widths = rng.uniform(0.6, 1.4, size=128).astype(numpy.float32)
    map_mean = rng.normal(scale=0.05, size=128).astype(numpy.float32)
    (nb_bits, nb_deads, rec, psnr) = reference_shaped_path(v, learned, images, bin_widths, map_mean, path, idx_map_exception)
    c = codec.BatchCodec(v, learned, bin_widths, map_mean, probabilities, idx_map_exception, 3, 64, 96, keep_reconstruction=True)
    device_images = torch.from_numpy(images).cuda()
    tickets = [c.submit(device_images[0:3]), c.submit(device_images[3:6]), c.submit(device_images[0:3])]   # three in flight
    results = [t.result() for t in tickets]
    for (k, lo) in enumerate((0, 3, 0)):
        r = results[k]
        assert numpy.array_equal(r['nb_bits'], nb_bits[lo:lo + 3]), k
        assert numpy.array_equal(r['nb_deads'], nb_deads[lo:lo + 3]), k
        assert numpy.array_equal(tickets[k].reconstruction_uint8.cpu().numpy(), rec[lo:lo + 3]), k
        for j in range(3):
            assert float(tls.psnr_from_sse(int(r['sse'][j]), 64*96)) == psnr[lo + j], (k, j)
        if idx_map_exception < 0:
            assert not r['exception_bits'].any()
    c.close()


def test_batch_codec_argument_checks_and_coder_errors(tmp_path):
    from autoencoder_based_image_compression_amd import codec
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    v = var.random_variables(1., False, seed=5)
    probabilities = numpy.full((128, 10), 0.5)
    ones = numpy.ones(128, dtype=numpy.float32)
    with pytest.raises(ValueError):
        codec.BatchCodec(v, False, ones, 0*ones, probabilities, 67, 1, 60, 96)           # 60 % 16 != 0
    with pytest.raises(ValueError):
        codec.BatchCodec(v, False, ones, 0*ones, probabilities[:5], 67, 1, 64, 96)
    c = codec.BatchCodec(v, False, ones, 0*ones, probabilities, 67, 1, 64, 96)
    with pytest.raises(TypeError):
        c.submit(torch.zeros((1, 64, 96), device='cuda'))
    with pytest.raises(ValueError):
        c.submit(torch.zeros((2, 64, 96), dtype=torch.uint8, device='cuda'))
    c.close()
    # an invalid probability that is actually used surfaces as the reference's RuntimeError from Ticket.result()
    bad = probabilities.copy()
    bad[:, 0] = numpy.nan
    c = codec.BatchCodec(v, False, ones, 0*ones, bad, 67, 1, 64, 96)
    ticket = c.submit(torch.full((1, 64, 96), 100, dtype=torch.uint8, device='cuda'))
    with pytest.raises(RuntimeError) as info:
        ticket.result()
    assert str(info.value) == 'Error of type 4 during the encoding.'
    c.close()


@pytest.mark.parametrize('shape', [(5, 16, 16), (2, 16, 48), (1, 32, 16)])
def test_batch_codec_on_tiny_images(tmp_path, shape):
    """Maps of 1, 3 and 2 symbols (the smallest sizes the x16 shape law allows): ragged tiles everywhere."""
    from autoencoder_based_image_compression_amd import codec
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    path = str(tmp_path/'binary_probabilities.npy')
    numpy.save(path, probabilities)
    rng = numpy.random.RandomState(shape[0])
    v = var.random_variables(1., False, seed=6, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    images = rng.randint(16, 236, size=shape).astype(numpy.uint8)
    bin_widths = numpy.full(128, 0.25, dtype=numpy.float32)
    map_mean = numpy.zeros(128, dtype=numpy.float32)
    (nb_bits, nb_deads, rec, psnr) = reference_shaped_path(v, False, images, bin_widths, map_mean, path, 67)
    c = codec.BatchCodec(v, False, bin_widths, map_mean, probabilities, 67, shape[0], shape[1], shape[2], keep_reconstruction=True)
    ticket = c.submit(torch.from_numpy(images).cuda())
    r = ticket.result()
    assert numpy.array_equal(r['nb_bits'], nb_bits) and numpy.array_equal(r['nb_deads'], nb_deads)
    assert numpy.array_equal(ticket.reconstruction_uint8.cpu().numpy(), rec)
    c.close()


@pytest.mark.parametrize('streams', [1, 2, 3])
@pytest.mark.parametrize('learned', [False, True])
def test_batch_codec_graph_replay_equals_the_launch_by_launch_path(learned, streams):
    """use_graphs: a step captured into one hipGraph per slot and replayed (with the coder as a forked branch) gives the same
    bits, squared errors, dead maps and reconstructions as the same codec launching kernel by kernel; replays of a slot with
    different images do not leak into each other."""
    from autoencoder_based_image_compression_amd import codec
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    rng = numpy.random.RandomState(23)
    v = var.random_variables(1., learned, seed=6, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    images = torch.from_numpy(rng.randint(16, 236, size=(20, 64, 96)).astype(numpy.uint8)).cuda()
    bin_widths = rng.uniform(0.6, 1.4, size=128).astype(numpy.float32)
    map_mean = rng.normal(scale=0.05, size=128).astype(numpy.float32)
    plain = codec.BatchCodec(v, learned, bin_widths, map_mean, probabilities, 67, 2, 64, 96, keep_reconstruction=True)
    expected = []
    for k in range(10):
        t = plain.submit(images[2*k:2*k + 2])
        expected.append((t.result(), t.reconstruction_uint8.cpu().numpy()))
    plain.close()
    graphed = codec.BatchCodec(v, learned, bin_widths, map_mean, probabilities, 67, 2, 64, 96, keep_reconstruction=True,
                               use_graphs=True, nb_transform_streams=streams)
    order = list(range(10)) + [3, 9, 0, 5, 5, 1]                  # 16 steps over 4 slots: every graph is replayed several times
    for start in range(0, len(order), 3):
        chunk = order[start:start + 3]
        tickets = [graphed.submit(images[2*k:2*k + 2]) for k in chunk]           # up to three steps in flight
        for (k, t) in zip(chunk, tickets):
            r = t.result()
            for key in ('nb_bits', 'coder_bits', 'exception_bits', 'sse', 'nb_deads'):
                assert numpy.array_equal(r[key], expected[k][0][key]), (k, key)
    # the reconstruction of a replayed slot is that step's until the slot comes up again
    t = graphed.submit(images[8:10])
    t.result()
    assert numpy.array_equal(t.reconstruction_uint8.cpu().numpy(), expected[4][1])
    assert all(g is not None for g in graphed._graphs)
    graphed.close()
    with pytest.raises(ValueError):
        codec.BatchCodec(v, learned, bin_widths, map_mean, probabilities, 67, 2, 64, 96, coder='host', use_graphs=True)


def test_a_second_graph_codec_captures_while_the_first_one_is_busy():
    """codecs of one process share the device's side streams, and their slices move with `nb_in_flight`; a second
    graph-mode codec whose first submit (= its capture) comes while the first codec has batches in flight must neither invalidate
    its own capture nor disturb the other's results: `_capture_all` waits until the other live codecs of the device are idle."""
    from autoencoder_based_image_compression_amd import codec
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    rng = numpy.random.RandomState(31)
    v = var.random_variables(1., False, seed=6, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    images = torch.from_numpy(rng.randint(16, 236, size=(12, 64, 96)).astype(numpy.uint8)).cuda()
    ones = numpy.ones(128, dtype=numpy.float32)
    with codec.BatchCodec(v, False, ones, 0*ones, probabilities, 67, 2, 64, 96) as plain:
        expected = [plain.submit(images[2*k:2*k + 2]).result() for k in range(6)]
    first = codec.BatchCodec(v, False, ones, 0*ones, probabilities, 67, 2, 64, 96, use_graphs=True, nb_transform_streams=3, nb_in_flight=3)
    second = codec.BatchCodec(v, False, ones, 0*ones, probabilities, 67, 2, 64, 96, use_graphs=True, nb_transform_streams=2, nb_in_flight=5)
    try:
        first.submit(images[0:2]).result()                           # first's graphs exist
        busy = [first.submit(images[2*(k % 6):2*(k % 6) + 2]) for k in range(8)]      # in flight while the second one captures
        mine = [second.submit(images[2*k:2*k + 2]) for k in range(6)]
        for (k, t) in enumerate(busy):
            r = t.result()
            for key in ('nb_bits', 'sse', 'nb_deads'):
                assert numpy.array_equal(r[key], expected[k % 6][key]), ('first', k, key)
        for (k, t) in enumerate(mine):
            r = t.result()
            for key in ('nb_bits', 'sse', 'nb_deads'):
                assert numpy.array_equal(r[key], expected[k][key]), ('second', k, key)
        # and interleaved afterwards
        for k in range(6):
            (a, b) = (first.submit(images[2*k:2*k + 2]), second.submit(images[2*k:2*k + 2]))
            assert numpy.array_equal(a.result()['nb_bits'], expected[k]['nb_bits']) and numpy.array_equal(b.result()['nb_bits'], expected[k]['nb_bits'])
    finally:
        first.close()
        second.close()


def test_graph_capture_with_one_transform_stream_at_kodak_size():
    """use_graphs with ONE transform stream on a batch whose step takes milliseconds: while slot 0's replay was still running, the
    capture of slot 1 (same stream) used to be invalidated by the result worker's event polls (hipErrorStreamCaptureInvalidated;
    the bench then sat in close() for its whole time limit). All slots are captured up front now: the steps below must all
    return, with the bits of the launch-by-launch path."""
    import bench
    from autoencoder_based_image_compression_amd import codec, device as dev, pipeline
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats
    variables = bench.synthetic_model(1.)
    images = torch.from_numpy(bench.synthetic_images(3, 24, 512, 768)).cuda()
    y0 = pipeline.DeviceEncoder(variables, False)(images)
    map_mean = dev.map_means(y0).cpu().numpy()
    probabilities = lossless_stats.compute_binary_probabilities(y0.cpu().numpy(), variables[var.BIN_WIDTHS_NAME], map_mean, 10)
    del y0
    with codec.BatchCodec(variables, False, variables[var.BIN_WIDTHS_NAME], map_mean, probabilities, 67, 24, 512, 768) as plain:
        expected = plain.submit(images).result()
    with codec.BatchCodec(variables, False, variables[var.BIN_WIDTHS_NAME], map_mean, probabilities, 67, 24, 512, 768, use_graphs=True,
                          nb_transform_streams=1) as graphed:
        tickets = [graphed.submit(images) for _ in range(8)]
        for t in tickets:
            r = t.result()
            assert numpy.array_equal(r['nb_bits'], expected['nb_bits']) and numpy.array_equal(r['sse'], expected['sse'])


@pytest.mark.parametrize('wait_mode', ['sequence', 'events'])
@pytest.mark.parametrize('graphs', [False, True])
def test_a_failed_submit_does_not_hang_the_codec(graphs, wait_mode, monkeypatch):
    """A launch that raises in the middle of `submit` must surface as that exception: the slot it had taken is given back, so
    leaving the `with` block (close -> drain) returns instead of waiting for a result nobody will post, and the codec goes on
    working (round 2: a failed graph capture showed up as a bench run that sat in drain() for its whole time limit) -- every slot
    of it, the one whose step was cut short included (`_resync`: its step counters read back, its ticket words and accumulators
    zeroed again), whichever way the result worker learns that a step is through (EAE_WORKER_WAIT: the step counter in pinned
    memory, or the events of rounds 3-4)."""
    from autoencoder_based_image_compression_amd import codec
    monkeypatch.setattr(codec, '_WAIT_MODE', wait_mode)
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    rng = numpy.random.RandomState(29)
    v = var.random_variables(1., False, seed=8, bias_std=0.01)
    images = torch.from_numpy(rng.randint(16, 236, size=(2, 64, 96)).astype(numpy.uint8)).cuda()
    bin_widths = numpy.ones(128, dtype=numpy.float32)
    map_mean = numpy.zeros(128, dtype=numpy.float32)
    with codec.BatchCodec(v, False, bin_widths, map_mean, probabilities, 67, 2, 64, 96, use_graphs=graphs) as c:
        good = c.submit(images).result()
        real = c._launch_coder

        def broken(slot, hook=None):
            raise RuntimeError('injected')

        c._launch_coder = broken
        if graphs:
            replay = torch.cuda.CUDAGraph.replay
            calls = [0]

            def failing_replay(self):
                calls[0] += 1
                if calls[0] == 2:                       # the coder graph of the step
                    raise RuntimeError('injected')
                return replay(self)

            monkeypatch.setattr(torch.cuda.CUDAGraph, 'replay', failing_replay)
        with pytest.raises(RuntimeError, match='injected'):
            c.submit(images)
        if graphs:
            monkeypatch.undo()
        c._launch_coder = real
        torch.cuda.synchronize()
        # twice round the slots, several in flight: same bits, same squared errors, same dead maps every time
        tickets = [c.submit(images) for _ in range(2*c.nb_slots)]
        for ticket in tickets:
            again = ticket.result()
            for key in ('nb_bits', 'sse', 'nb_deads'):
                assert numpy.array_equal(again[key], good[key]), key


@pytest.mark.parametrize('learned', [False, True])
@pytest.mark.parametrize('shape', [(3, 64, 96), (24, 128, 192)])
def test_fused_latent_stage_gives_the_same_results(learned, shape):
    """`fuse_latent`: the latent stage as the epilogue of the conv_3 launch (small batch: two launches inside the entry point;
    24 x 128x192: the fused conv_3 entry point on a mid-size batch) against the default codec."""
    from autoencoder_based_image_compression_amd import codec
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    rng = numpy.random.RandomState(29)
    v = var.random_variables(1., learned, seed=8, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    images = torch.from_numpy(rng.randint(16, 236, size=shape).astype(numpy.uint8)).cuda()
    bin_widths = rng.uniform(0.6, 1.4, size=128).astype(numpy.float32)
    map_mean = rng.normal(scale=0.05, size=128).astype(numpy.float32)
    results = []
    for fuse in (False, True):
        with codec.BatchCodec(v, learned, bin_widths, map_mean, probabilities, 67, shape[0], shape[1], shape[2], keep_reconstruction=True,
                              fuse_latent=fuse) as c:
            t = c.submit(images)
            results.append((t.result(), t.reconstruction_uint8.cpu().numpy()))
    for key in ('nb_bits', 'coder_bits', 'exception_bits', 'sse', 'nb_deads'):
        assert numpy.array_equal(results[0][0][key], results[1][0][key]), key
    assert numpy.array_equal(results[0][1], results[1][1])


def test_a_failed_hand_off_raises_from_the_ticket_and_the_next_batches_are_right(launch_options):
    """The cut-tile hand-off of the conv launches forced to fail (csrc/hip/conv_gemm_split.hip, eae_hip_debug_set_split_mute):
    `Ticket.result()` raises for exactly those batches; with the hook off again the same codec -- every slot, hence every
    workspace, used once more -- returns what a fresh codec returns."""
    from autoencoder_based_image_compression_amd import codec, device as dev
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    launch_options.clear()
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    v = var.random_variables(1., False, seed=8, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    images = torch.from_numpy(numpy.random.RandomState(9).randint(16, 236, size=(3, 64, 96)).astype(numpy.uint8)).cuda()
    ones = numpy.ones(128, dtype=numpy.float32)
    with codec.BatchCodec(v, False, ones, 0*ones, probabilities, 67, 3, 64, 96) as fresh:
        expected = fresh.submit(images).result()
    launch_options.setenv('EAE_HIP_GEMM', 's')                  # cut every conv launch, small as they are
    with codec.BatchCodec(v, False, ones, 0*ones, probabilities, 67, 3, 64, 96) as c:
        first = c.submit(images).result()
        for key in expected:
            assert numpy.array_equal(first[key], expected[key]), key
        launch_options.split_mute(True)
        failed = [c.submit(images) for _ in range(2)]
        for ticket in failed:
            with pytest.raises(dev.SplitHandOffTimeout):
                ticket.result()
        launch_options.split_mute(False)
        for _ in range(c.nb_slots + 1):
            r = c.submit(images).result()
            for key in expected:
                assert numpy.array_equal(r[key], expected[key]), key


def test_exception_map_symbols_beyond_the_histogram_radius_are_counted_again(tmp_path):
    """The reference's exception-map cost is a histogram from the smallest to the largest symbol, whatever they are
    (lossless/compression.py:68-75): a codec whose first histogram is too narrow must return the same bits (it counts the map
    again over all of int16), not raise."""
    from autoencoder_based_image_compression_amd import codec
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    path = str(tmp_path/'binary_probabilities.npy')
    numpy.save(path, probabilities)
    v = var.random_variables(1., False, seed=10, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    images = numpy.random.RandomState(11).randint(16, 236, size=(2, 64, 96)).astype(numpy.uint8)
    bin_widths = numpy.full(128, 0.05, dtype=numpy.float32)      # symbols of a few tens
    map_mean = numpy.zeros(128, dtype=numpy.float32)
    (nb_bits, nb_deads, rec, psnr) = reference_shaped_path(v, False, images, bin_widths, map_mean, path, 67)
    device_images = torch.from_numpy(images).cuda()
    with codec.BatchCodec(v, False, bin_widths, map_mean, probabilities, 67, 2, 64, 96) as wide:
        a = wide.submit(device_images).result()
    with codec.BatchCodec(v, False, bin_widths, map_mean, probabilities, 67, 2, 64, 96, hist_radius=1) as narrow:
        b = narrow.submit(device_images).result()
        b2 = narrow.submit(device_images).result()
    assert int(numpy.abs(a['exception_bits']).sum()) > 0
    for key in a:
        assert numpy.array_equal(a[key], b[key]) and numpy.array_equal(a[key], b2[key]), key
    assert numpy.array_equal(a['nb_bits'], nb_bits)


@pytest.mark.parametrize('mode', ['launches', 'two_streams', 'graphs'])
def test_host_in_host_out_equals_the_device_resident_path(mode):
    """The feed and fetch of the reference's `sess.run` (eae/batching.py:95-99, 49-53: numpy in, numpy out) as the codec does
    them: `submit()` of a PINNED host batch, reconstructions copied to pinned host memory by the result worker
    (`fetch_reconstruction=True`). Same bits, errors, dead maps and reconstructions as with the batch resident on the device, in
    every launch mode, with more batches than slots (the pinned buffers come round)."""
    from autoencoder_based_image_compression_amd import codec
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    rng = numpy.random.RandomState(29)
    v = var.random_variables(1., False, seed=6, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    batches = [rng.randint(16, 236, size=(3, 64, 96)).astype(numpy.uint8) for _ in range(9)]
    bin_widths = numpy.ones(128, dtype=numpy.float32)
    map_mean = rng.normal(scale=0.05, size=128).astype(numpy.float32)
    kw = {'launches': {}, 'two_streams': {'nb_transform_streams': 2}, 'graphs': {'nb_transform_streams': 2, 'use_graphs': True}}[mode]
    with codec.BatchCodec(v, False, bin_widths, map_mean, probabilities, 67, 3, 64, 96, nb_in_flight=2, keep_reconstruction=True, **kw) as c:
        want = []
        for b in batches:
            t = c.submit(torch.from_numpy(b).cuda())
            want.append((t.result(), t.reconstruction_uint8.cpu().numpy()))
    with codec.BatchCodec(v, False, bin_widths, map_mean, probabilities, 67, 3, 64, 96, nb_in_flight=2, fetch_reconstruction=True, **kw) as c:
        with pytest.raises(ValueError):
            c.submit(torch.from_numpy(batches[0]))                     # pageable host memory: the copy would not be asynchronous
        pinned = [torch.from_numpy(b).pin_memory() for b in batches]
        tickets = [c.submit(p) for p in pinned]                        # nine batches through four slots
        for (k, t) in enumerate(tickets):
            r = t.result()
            assert t.fed_event is not None and t.fed_event.query()
            for key in ('nb_bits', 'sse', 'nb_deads'):
                assert numpy.array_equal(r[key], want[k][0][key]), (k, key)
        # a slot's pinned buffer is rewritten when the slot comes round: the last nb_slots reconstructions are still there
        for k in range(len(batches) - c.nb_slots, len(batches)):
            assert numpy.array_equal(tickets[k].reconstruction_host, want[k][1]), k


def test_host_reconstructions_one_at_a_time_are_all_right():
    """Submit, wait, compare, for more batches than slots: every reconstruction that reaches the host is the device's."""
    from autoencoder_based_image_compression_amd import codec
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    rng = numpy.random.RandomState(31)
    v = var.random_variables(1., False, seed=7, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    bin_widths = numpy.ones(128, dtype=numpy.float32)
    map_mean = numpy.zeros(128, dtype=numpy.float32)
    with codec.BatchCodec(v, False, bin_widths, map_mean, probabilities, 67, 2, 32, 48, nb_in_flight=1, fetch_reconstruction=True,
                          nb_transform_streams=2, use_graphs=True) as c:
        for k in range(8):
            b = torch.from_numpy(rng.randint(16, 236, size=(2, 32, 48)).astype(numpy.uint8)).pin_memory()
            t = c.submit(b)
            t.result()
            assert numpy.array_equal(t.reconstruction_host, t.reconstruction_uint8.cpu().numpy()), k


def test_the_product_mode_at_kodak_size_equals_the_conservative_one():
    """`codec.product_mode(h, w)` (three transform streams, hipGraph replay, six coder batches in flight at this size: what
    bench.py's headline runs) against the constructor's defaults (every launch on the caller's stream), 24 images of 512 x 768,
    twelve steps through the eight slots: same bits, errors, dead maps."""
    from autoencoder_based_image_compression_amd import codec
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    mode = codec.product_mode(512, 768)
    assert mode == {'nb_in_flight': 6, 'nb_transform_streams': 3, 'use_graphs': True}
    rng = numpy.random.RandomState(37)
    v = var.random_variables(1., False, seed=8, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    batches = [torch.from_numpy(rng.randint(16, 236, size=(24, 512, 768), dtype=numpy.uint8)).cuda() for _ in range(3)]
    bin_widths = numpy.ones(128, dtype=numpy.float32)
    map_mean = numpy.zeros(128, dtype=numpy.float32)
    with codec.BatchCodec(v, False, bin_widths, map_mean, probabilities, 67, 24, 512, 768) as plain:
        want = [plain.submit(b).result() for b in batches]
    with codec.BatchCodec(v, False, bin_widths, map_mean, probabilities, 67, 24, 512, 768, **mode) as c:
        tickets = [c.submit(batches[k % 3]) for k in range(12)]
        for (k, t) in enumerate(tickets):
            r = t.result()
            for key in ('nb_bits', 'sse', 'nb_deads'):
                assert numpy.array_equal(r[key], want[k % 3][key]), (k, key)


def test_the_product_library_has_no_chunked_coder():
    """`coder_chunks > 1` on lib/libeae_hip.so: the constructor says that the chunked round trip is experimental and where it lives,
    and nothing is left behind (no worker thread yet, no stream); the default (one chunk) is what every other test here runs."""
    from autoencoder_based_image_compression_amd import _native, codec
    from autoencoder_based_image_compression_amd import device as dev
    import bench
    if _native.has_experimental_coder():
        pytest.skip('EAE_HIP_LIB names a build with the experimental coder')
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    variables = bench.synthetic_model(0.25)
    with pytest.raises(dev.ExperimentalCoderMissing, match='libeae_hip_test.so'):
        codec.BatchCodec(variables, False, variables['piecewise_linear_function/bin_widths'], numpy.zeros(128, dtype=numpy.float32), probabilities,
                         67, 1, 128, 192, coder_chunks=4)


@pytest.mark.parametrize('graphs', [False, True])
def test_a_trailing_coder_gives_the_same_results(graphs, test_library):
    """One image per batch: the coder's chains cut into launches that overlap (`coder_chunks`) against the two calls one after
    the other (`coder_chunks=1`, the default), launched kernel by kernel and replayed as hipGraphs. The chunked form is
    experimental: it exists in the test build of the library only (fixture `test_library`)."""
    from autoencoder_based_image_compression_amd import codec
    import bench
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    variables = bench.synthetic_model(0.25)
    images = torch.from_numpy(bench.synthetic_images(7, 3, 128, 192)).cuda()
    mean = numpy.zeros(128, dtype=numpy.float32)
    results = {}
    for chunks in (None, 4, 7):
        kwargs = {'nb_in_flight': 2, 'nb_transform_streams': 2, 'use_graphs': True} if graphs else {}
        with codec.BatchCodec(variables, False, variables['piecewise_linear_function/bin_widths'], mean, probabilities, 67, 1, 128, 192,
                              coder_chunks=chunks, **kwargs) as c:
            assert c.coder_chunks == (1 if chunks is None else chunks)
            tickets = [c.submit(images[j:j + 1]) for j in (0, 1, 2, 0, 1, 2)]
            results[chunks] = [t.result() for t in tickets]
    for chunks in (4, 7):
        for (a, b) in zip(results[None], results[chunks]):
            assert all(numpy.array_equal(a[k], b[k]) for k in a), chunks


@pytest.mark.parametrize('graphs', [False, True])
@pytest.mark.parametrize('batch', [1, 3])
def test_steps_on_one_stream_give_the_same_results(graphs, batch):
    """`one_stream_steps`: a step's coder behind its synthesis transform on the step's own stream (one graph launch per step, no
    event between streams; the mode for one or two images per step) against the default schedule: every result of every step
    equal, the reconstructions too, through many more steps than there are slots."""
    from autoencoder_based_image_compression_amd import codec
    import bench
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    variables = bench.synthetic_model(0.25)
    images = torch.from_numpy(bench.synthetic_images(11, 3*batch, 128, 192)).cuda()
    mean = numpy.zeros(128, dtype=numpy.float32)
    (results, reconstructions) = ({}, {})
    for one in (False, True):
        kwargs = {'nb_in_flight': 4, 'nb_transform_streams': 4, 'use_graphs': True} if graphs else {'nb_in_flight': 2, 'nb_transform_streams': 2}
        with codec.BatchCodec(variables, False, variables['piecewise_linear_function/bin_widths'], mean, probabilities, 67, batch, 128, 192,
                              keep_reconstruction=True, one_stream_steps=one, **kwargs) as c:
            (results[one], reconstructions[one]) = ([], [])
            for j in [0, 1, 2]*7:
                t = c.submit(images[j*batch:(j + 1)*batch])
                results[one].append(t.result())
                reconstructions[one].append(t.reconstruction_uint8.cpu().numpy().copy())
            tickets = [c.submit(images[j*batch:(j + 1)*batch]) for j in [2, 1, 0]*5]      # ... and with the steps in flight together
            results[one].extend(t.result() for t in tickets)
    assert len(results[True]) == len(results[False]) == 36
    for (a, b) in zip(results[False], results[True]):
        assert all(numpy.array_equal(a[k], b[k]) for k in a)
    for (a, b) in zip(reconstructions[False], reconstructions[True]):
        assert numpy.array_equal(a, b)
    assert sum(int(r['nb_bits'].sum()) for r in results[True]) > 0


@pytest.mark.parametrize('graphs', [False, True])
@pytest.mark.parametrize('by_caller, early', [(False, False), (True, False), (True, True), (False, True)])
def test_who_forms_the_results_and_when_changes_nothing(graphs, by_caller, early, monkeypatch):
    """Small steps (round 6): the results formed by the worker thread or by the thread that called `result()` (`codec._Job`), with the
    analysis side's blocks published at the end of the step or ahead of the coder (`BatchCodec._early_publish`: exception-map bits,
    dead-map counts and squared errors then formed while the coder still runs): every combination gives the same values, one step
    at a time and with steps in flight, with an exception map whose histogram overflows its radius on the way (the recount
    path) -- and the range-check error still arrives as the reference's AssertionError."""
    from autoencoder_based_image_compression_amd import codec
    import bench
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    variables = bench.synthetic_model(0.25)
    images = torch.from_numpy(bench.synthetic_images(13, 4, 128, 192)).cuda()
    mean = numpy.zeros(128, dtype=numpy.float32)
    kwargs = {'nb_in_flight': 3, 'nb_transform_streams': 3, 'use_graphs': True} if graphs else {}

    def run(publishes_early):
        with codec.BatchCodec(variables, False, variables['piecewise_linear_function/bin_widths'], mean, probabilities, 67, 1, 128, 192,
                              hist_radius=2, **kwargs) as c:                      # (radius 2: the exception map's histogram overflows, `recount`)
            assert c._early_publish == publishes_early
            one_at_a_time = [c.submit(images[j:j + 1]).result() for j in [0, 1, 2, 3]*4]
            tickets = [c.submit(images[j:j + 1]) for j in [3, 2, 1, 0]*3]
            return one_at_a_time + [t.result() for t in tickets]

    monkeypatch.setattr(codec, '_RESULT_BY_CALLER', False)
    monkeypatch.setattr(codec, '_EARLY_PUBLISH', False)
    want = run(False)
    monkeypatch.setattr(codec, '_RESULT_BY_CALLER', by_caller)
    monkeypatch.setattr(codec, '_EARLY_PUBLISH', early)
    got = run(early)
    assert len(got) == len(want) == 28
    for (a, b) in zip(want, got):
        assert sorted(a) == sorted(b) and all(numpy.array_equal(a[k], b[k]) for k in a)
    assert sum(int(r['exception_bits'].sum()) for r in got) > 0
    # the range check of the quantiser (tools.py:130-132) through the same path: a bin width that takes symbols out of int16
    tiny = dict(variables)
    tiny['piecewise_linear_function/bin_widths'] = numpy.full(128, 1e-7, dtype=numpy.float32)
    with codec.BatchCodec(tiny, False, tiny['piecewise_linear_function/bin_widths'], mean, probabilities, 67, 1, 128, 192, **kwargs) as c:
        with pytest.raises(AssertionError, match='16-bit signed integers'):
            c.submit(images[0:1]).result()


@pytest.mark.parametrize('words', [2, 50, 4096, 300001])
def test_publish_step_copies_clears_and_counts(words):
    """include/eae_hip.h: eae_hip_publish_step on its own -- the block reaches pinned memory whole, the source is zeroed from `clear_from`
    on and untouched below it, the step counter is bumped once per call and left in pinned memory, the ticket word is back at zero
    (one block and many blocks); with a conv workspace whose error word is set, the count lands in the error word inside the block
    before it is copied and the workspace is all zero again."""
    from autoencoder_based_image_compression_amd import device as dev
    rng = numpy.random.RandomState(words)
    src_host = rng.randint(1, 1 << 30, size=words).astype(numpy.int32)
    src = torch.from_numpy(src_host.copy()).cuda()
    dst = torch.zeros(words, dtype=torch.int32).pin_memory()
    seq = torch.zeros(2, dtype=torch.int32, device='cuda')
    word = torch.zeros(1, dtype=torch.int32).pin_memory()
    clear_from = words//3
    for call in (1, 2, 3):
        dev.publish_step(src, dst, clear_from, seq[1:2], seq[0:1], word)
        torch.cuda.synchronize()
        expected = src_host.copy()
        if call > 1:
            expected[clear_from:] = 0
        assert numpy.array_equal(dst.numpy(), expected)
        assert numpy.array_equal(src.cpu().numpy()[:clear_from], src_host[:clear_from]) and not src[clear_from:].any().item()
        assert int(word.item()) == call and seq.cpu().tolist() == [call, 0]
    if words <= 65536:
        ws = dev.conv_workspace('cuda')
        ws[255] = 3                                           # the error word of a workspace: three tails gave up
        ws[300] = 1                                           # a flag nobody reset
        src.copy_(torch.from_numpy(src_host))
        src[words - 1] = 0
        dev.publish_step(src, dst, words, seq[1:2], seq[0:1], word, conv_ws=ws, error_word=src[words - 1:words])
        torch.cuda.synchronize()
        assert int(dst[words - 1].item()) == 3 and numpy.array_equal(dst.numpy()[:-1], src_host[:-1])
        assert not ws.any().item() and int(word.item()) == 4
        with pytest.raises(dev.HipError):
            dev.publish_step(src, dst, 0, seq[1:2], seq[0:1], word, conv_ws=ws)      # a workspace without its error word


@pytest.mark.parametrize('graphs', [False, True])
def test_steps_on_one_stream_with_host_batches_and_the_host_coder(graphs):
    """`one_stream_steps` next to the codec's other options: a pinned host batch in and the reconstruction back in pinned memory
    (`fetch_reconstruction`), and the host coder (`coder='host'`): the same results and reconstructions as the default schedule."""
    from autoencoder_based_image_compression_amd import codec
    import bench
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    variables = bench.synthetic_model(0.25)
    host_images = torch.from_numpy(bench.synthetic_images(13, 4, 128, 192)).pin_memory()
    mean = numpy.zeros(128, dtype=numpy.float32)
    outcomes = {}
    cases = ((False, 'device'), (True, 'device')) + (() if graphs else ((True, 'host'),))      # (graph replay needs the coder on the device)
    for (one, coder) in cases:
        kwargs = {'nb_in_flight': 3, 'nb_transform_streams': 3, 'use_graphs': True} if graphs else {'nb_in_flight': 2, 'nb_transform_streams': 2}
        with codec.BatchCodec(variables, False, variables['piecewise_linear_function/bin_widths'], mean, probabilities, 67, 2, 128, 192,
                              fetch_reconstruction=True, one_stream_steps=one, coder=coder, host_coder_threads=2, **kwargs) as c:
            rows = []
            for j in [0, 1]*6:
                t = c.submit(host_images[2*j:2*j + 2])
                r = t.result()
                rows.append((r, t.reconstruction_host.copy()))
            outcomes[(one, coder)] = rows
    reference = outcomes[(False, 'device')]
    for key in cases[1:]:
        for ((a, rec_a), (b, rec_b)) in zip(reference, outcomes[key]):
            assert all(numpy.array_equal(a[k], b[k]) for k in ('nb_bits', 'sse', 'nb_deads')), key
            assert numpy.array_equal(rec_a, rec_b), key
