"""Host-side logic of the reference-shaped surface that needs no GPU: argument validation, naming helpers, the float64
tails (entropy, PSNR) fed with exact integer counts, the TF shim, sharding arithmetic. Expected values come from the
reference's own Python (tests/golden/tools_golden.npz, oracle/gen_golden.py)."""
import os

import numpy
import pytest

from autoencoder_based_image_compression_amd import sharding
from autoencoder_based_image_compression_amd.kodak import tf_shim
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
from autoencoder_based_image_compression_amd.kodak.eae.graph.EntropyAutoencoder import EntropyAutoencoder
from autoencoder_based_image_compression_amd.kodak.eae.graph.IsolatedDecoder import IsolatedDecoder
from autoencoder_based_image_compression_amd.kodak.lossless import compression
from autoencoder_based_image_compression_amd.kodak.lossless import stats
from autoencoder_based_image_compression_amd.kodak.tools import tools as tls

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tools_golden.npz')


@pytest.fixture(scope='module')
def gold():
    with numpy.load(GOLD) as data:
        return {k: data[k] for k in data.files}


def test_float_to_str(gold):
    got = [tls.float_to_str(v) for v in (1., 0.5, 1.25, -2., 10000., -0.75)]
    assert got == list(gold['float_to_str']) == ['1', '0dot5', '1dot25', 'minus2', '10000', 'minus0dot75']


def test_subdivide_set():
    assert tls.subdivide_set(24, 4) == 6
    with pytest.raises(ValueError):
        tls.subdivide_set(25, 4)


def test_entropy_tail_matches_reference(gold):
    """tools.py:523-537 from integer histograms: the reference example -5*0.1*log2(0.1) - 0.2*log2(0.2) - 0.3*log2(0.3)."""
    hist = numpy.array([3, 2, 1, 1, 1, 1, 1], dtype=numpy.int64)
    expected = -5*0.1*numpy.log2(0.1) - 0.2*numpy.log2(0.2) - 0.3*numpy.log2(0.3)
    assert tls._entropy_from_hist(hist) == gold['ent_out']
    assert abs(tls._entropy_from_hist(hist) - expected) < 1e-15
    assert tls._entropy_from_hist(gold['lat_count_symbols_3']) == gold['lat_entropy'][3]


def test_psnr_tail_matches_reference(gold):
    """tools.py:873-881: psnr_2d(12s, 15s) = 38.5883785143 (test_tools.py:494-509)."""
    assert tls.psnr_from_sse(9*24, 24) == gold['psnr_known']
    assert round(float(tls.psnr_from_sse(9*24, 24)), 10) == 38.5883785143
    sse = int(((gold['psnr_a'].astype(numpy.int64) - gold['psnr_b'].astype(numpy.int64))**2).sum())
    assert tls.psnr_from_sse(sse, gold['psnr_a'].size) == gold['psnr_ab']
    with pytest.raises(ValueError):
        tls.psnr_from_sse(0, 24)


def test_rate_tail_matches_reference(gold):
    entropies = gold['lat_entropy']
    rate = tls.rate_from_entropies(entropies, 8, 12, 128, 192)
    assert rate == gold['lat_rate'][0]


def test_argument_validation_without_gpu():
    data = numpy.zeros((1, 2, 2, 4), dtype=numpy.float32)
    with pytest.raises(ValueError):
        tls.quantize_per_map(data, numpy.ones((2, 2), dtype=numpy.float32))
    with pytest.raises(ValueError):
        tls.quantize_per_map(data, numpy.ones(3, dtype=numpy.float32))
    with pytest.raises(ValueError):
        tls.quantize_per_map(data, numpy.array([1., 0., 1., 1.], dtype=numpy.float32))
    with pytest.raises(ValueError):
        tls.quantize_per_map(data[0], numpy.ones(4, dtype=numpy.float32))   # ndim != 4 -> unpacking error
    with pytest.raises(TypeError):
        tls.cast_bt601(numpy.zeros(3, dtype=numpy.int32))
    with pytest.raises(TypeError):
        tls.cast_float_to_int16(numpy.zeros(3, dtype=numpy.uint8))
    with pytest.raises(ValueError):
        tls.count_nb_deads(numpy.zeros((2, 2, 2)))
    with pytest.raises(ValueError):
        tls.count_symbols(numpy.zeros(3, dtype=numpy.float32), 0.)
    with pytest.raises(ValueError):
        tls.rate_3d(numpy.zeros((2, 2, 4), dtype=numpy.float32), numpy.ones((1, 4), dtype=numpy.float32), 32, 32)
    with pytest.raises(ValueError):
        tls.rate_3d(numpy.zeros((2, 2, 4), dtype=numpy.float32), numpy.ones(5, dtype=numpy.float32), 32, 32)
    a = numpy.zeros((4, 4), dtype=numpy.uint8)
    with pytest.raises(TypeError):
        tls.psnr_2d(a.astype(numpy.float32), a)
    with pytest.raises(TypeError):
        tls.psnr_2d(a, a.astype(numpy.int16))
    with pytest.raises(ValueError):
        tls.psnr_2d(a[None], a[None])
    with pytest.raises(ValueError):
        tls.psnr_2d(a, a[:2])
    with pytest.raises(ValueError):
        stats.count_binary_decisions(numpy.array([-1.], dtype=numpy.float32), 1., 4)
    with pytest.raises(ValueError):
        compression.rescale_compress_lossless_maps(numpy.zeros((2, 2, 3), dtype=numpy.float32), numpy.ones((1, 3), dtype=numpy.float32), 'x.npy')
    with pytest.raises(ValueError):
        compression.rescale_compress_lossless_maps(numpy.zeros((2, 2, 3), dtype=numpy.float32), numpy.ones(4, dtype=numpy.float32), 'x.npy')
    with pytest.raises(TypeError):
        compression.compress_lossless_maps(numpy.zeros((2, 2, 3), dtype=numpy.int32), 'x.npy')


def test_graph_constructors_check_shapes_like_the_reference():
    """EntropyAutoencoder.py:77-80."""
    with pytest.raises(ValueError) as info:
        EntropyAutoencoder(4, 500, 768, 1., 10000., '', False)
    assert 'height' in str(info.value)
    with pytest.raises(ValueError) as info:
        EntropyAutoencoder(4, 512, 770, 1., 10000., '', False)
    assert 'width' in str(info.value)
    ae = EntropyAutoencoder(4, 512, 768, 1., 10000., '', False)
    assert ae.node_visible_units.shape == (4, 512, 768, 1)
    dec = IsolatedDecoder(4, 512, 768, True)
    assert dec.node_quantized_y.shape == (4, 32, 48, 128)
    with pytest.raises(RuntimeError):
        ae.get_bin_widths()


def test_session_shim():
    ph = tf_shim.Placeholder((2, 3), 'x')
    node = tf_shim.Node(lambda v: v*2, ph, 'twice')
    with tf_shim.Session() as sess:
        out = sess.run(node, feed_dict={ph: numpy.ones((2, 3))})
        assert numpy.array_equal(out, 2*numpy.ones((2, 3)))
        with pytest.raises(ValueError):
            sess.run(node, feed_dict={ph: numpy.ones((3, 3))})
        with pytest.raises(ValueError):
            sess.run(node)
    tf_shim.reset_default_graph()


def test_random_variables_follow_the_reference_initialiser(tmp_path):
    """EntropyAutoencoder.py:130-224, tfutils.py:445-478: shapes, symmetric gamma in [2e-5, 0.01], beta = 1, bw."""
    v = var.random_variables(0.5, False, seed=3)
    assert set(v) == set(var.ENCODER_NAMES + var.ENCODER_NAMES_FIXED_BW + var.DECODER_NAMES + var.DECODER_NAMES_FIXED_BW + (var.BIN_WIDTHS_NAME,))
    for (name, array) in v.items():
        assert array.dtype == numpy.float32 and tuple(array.shape) == var.SHAPES[name]
    g = v['encoder/gamma_2']
    assert numpy.array_equal(g, g.T) and g.min() >= 2e-5 and g.max() <= 0.01
    assert numpy.all(v['decoder/beta_5'] == 1.) and numpy.all(v['encoder/biases_1'] == 0.)
    assert numpy.all(v[var.BIN_WIDTHS_NAME] == numpy.float32(0.5))
    assert abs(float(v['encoder/weights_3'].std()) - 0.05) < 0.002 and abs(float(v['decoder/weights_6'].std()) - 0.01) < 0.001
    learned = var.random_variables(0.5, True, seed=3)
    assert 'encoder/gamma_3' not in learned and 'decoder/gamma_4' not in learned
    path = str(tmp_path/'model_10.npz')
    var.save_variables(path, v)
    back = var.load_variables(path)
    assert all(numpy.array_equal(back[k], v[k]) for k in v)
    with pytest.raises(ValueError):
        var.initialize_weights_gdn(128, 0.02, numpy.random.RandomState(0))


def test_shard_bounds():
    assert [sharding.shard_bounds(512, r, 8) for r in range(8)] == [(64*r, 64*r + 64) for r in range(8)]
    assert [sharding.shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [sharding.shard_bounds(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert sharding.shard_bounds(0, 0, 2) == (0, 0)
    with pytest.raises(ValueError):
        sharding.shard_bounds(4, 4, 4)
    assert numpy.array_equal(sharding.reduce_statistics([1., 2.]), [1., 2.])


def test_decisions_from_hist_is_stats_py_181_195(gold):
    hist_abs = numpy.zeros(16, dtype=numpy.int64)
    for s in numpy.round(gold['cbd1_in']/numpy.float32(0.05)).astype(int):
        hist_abs[s] += 1
    (z, o) = stats._decisions_from_hist(hist_abs, 7)
    assert numpy.array_equal(z, gold['cbd1_zeros']) and numpy.array_equal(o, gold['cbd1_ones'])
    assert list(z) == [0, 1, 1, 1, 2, 0, 0] and list(o) == [6, 5, 4, 3, 1, 1, 1]


def test_exception_map_bits(gold):
    """compression.py:73-74 from a histogram."""
    sym = gold['lossless_symbols'][:, :, 67].astype(numpy.int64)
    hist = numpy.bincount(sym.reshape(-1) + 255, minlength=511)
    assert int(compression.exception_map_nb_bits(hist, sym.size)) == int(gold['lossless_bits_each_map'][67])


def test_dropin_directory_shadows_the_reference_imports():
    """The imports at the top of reconstructing_eae_kodak.py (:21-29) resolve to this build when `dropin/` is first on
    the path; out-of-scope helpers fail with a clear message instead of silently missing."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys; sys.path.insert(0, {0!r}); sys.path.insert(0, {1!r});'
            'import tensorflow as tf; import eae.batching; import lossless.compression; import lossless.interface_cython;'
            'import lossless.stats; import tools.tools as tls; import tfutils.tfutils;'
            'from eae.graph.EntropyAutoencoder import EntropyAutoencoder; from eae.graph.IsolatedDecoder import IsolatedDecoder;'
            'import eae.graph.constants as csts;'
            'assert csts.STRIDE_PROD == 16 and tls.float_to_str(0.5) == "0dot5";'
            'assert "autoencoder_based_image_compression_amd" in tf.Session.__module__;'
            'tf.reset_default_graph();'
            'assert callable(tls.compute_bjontegaard) and callable(tls.visualize_rotated_luminance);'
            '\ntry:\n    tls.plot_graphs\n    raise SystemExit(3)\nexcept NotImplementedError:\n    pass\n'
            'print("dropin ok")').format(root, os.path.join(root, 'autoencoder_based_image_compression_amd', 'dropin'))
    out = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert out.returncode == 0 and 'dropin ok' in out.stdout, out.stdout


def test_jensen_shannon_divergence_and_probability_intervals_against_the_reference():
    """tools.py:615-666 and stats.py:70-134 (host numpy in the reference too): values from the reference's own functions
    (tests/golden/tools_golden.npz), and their ValueErrors."""
    import os
    from autoencoder_based_image_compression_amd.kodak.lossless import stats
    from autoencoder_based_image_compression_amd.kodak.tools import tools as tls
    with numpy.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tools_golden.npz')) as g:
        assert tls.jensen_shannon_divergence(g['js_p0'], g['js_p1']) == g['js_out']
        y = g['stats_y']
        (edges, probs) = stats.compute_probabilities_intervals(y[:, :, :, 12], 1.)
        assert numpy.array_equal(edges, g['stats_edges12']) and numpy.array_equal(probs, g['stats_probs12'])
        (edges, probs) = stats.compute_probabilities_intervals(y[:, :, :, 3], 0.5)
        assert numpy.array_equal(edges, g['stats_edges_half']) and numpy.array_equal(probs, g['stats_probs_half'])
    assert tls.jensen_shannon_divergence(numpy.array([0.5, 0.5]), numpy.array([0.5, 0.5])) == 0.
    with pytest.raises(ValueError):
        tls.jensen_shannon_divergence(numpy.array([0., 1.]), numpy.array([0.5, 0.5]))
    with pytest.raises(ValueError):
        tls.jensen_shannon_divergence(numpy.array([0.5, 0.5]), numpy.array([0.3, 0.3]))
    with pytest.raises(ValueError):
        stats.compute_probabilities_intervals(numpy.zeros(4, dtype=numpy.float32), 1.)        # range 0 < interval
    with pytest.raises(ValueError):
        stats.compute_probabilities_intervals(numpy.array([0.2, 2.7], dtype=numpy.float32), 0.7)   # 3 / 0.7 not an integer


def test_compute_bjontegaard_against_the_reference():
    """tools.py:157-263 (host numpy in the reference too) on the committed curves, and its argument checks."""
    import os
    from autoencoder_based_image_compression_amd.kodak.tools import tools as tls
    with numpy.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tools_golden.npz')) as g:
        (r0, p0, r1, p1, expected) = (g['bd_r0'], g['bd_p0'], g['bd_r1'], g['bd_p1'], g['bd_out'])
    assert tls.compute_bjontegaard(r0, p0, r1, p1) == expected and expected < 0.      # curve 1 saves bitrate
    assert abs(tls.compute_bjontegaard(r0, p0, r0, p0)) < 1e-9
    with pytest.raises(ValueError):
        tls.compute_bjontegaard(r0.reshape(1, -1), p0, r1, p1)
    with pytest.raises(ValueError):
        tls.compute_bjontegaard(r0, p0[:-1], r1, p1)
    with pytest.raises(AssertionError):
        tls.compute_bjontegaard(-r0, p0, r1, p1)


def test_png_dump_helpers(tmp_path):
    """crop_repeat_2d / visualize_crops / visualize_rotated_luminance / save_image (tools.py:434-484, 1172-1218, 1292-1330,
    1082-1106): host-side harness helpers, same outputs and exceptions."""
    import PIL.Image
    from autoencoder_based_image_compression_amd.kodak.tools import tools as tls
    image = numpy.random.RandomState(0).randint(0, 256, size=(100, 130)).astype(numpy.uint8)
    crop = tls.crop_repeat_2d(image, 3, 7)
    assert crop.shape == (160, 160) and crop.dtype == numpy.uint8
    assert numpy.array_equal(crop, numpy.repeat(numpy.repeat(image[3:83, 7:87], 2, axis=0), 2, axis=1))
    with pytest.raises(TypeError):
        tls.crop_repeat_2d(image.astype(numpy.float32), 0, 0)
    with pytest.raises(ValueError):
        tls.crop_repeat_2d(image, 20, 0)          # 20 + 80 >= 100
    with pytest.raises(ValueError):
        tls.crop_repeat_2d(image, 0, 50)          # 50 + 80 >= 130
    positions = numpy.array([[0, 10], [5, 15]], dtype=numpy.int32)
    paths = [str(tmp_path/name) for name in ('full.png', 'crop_0.png', 'crop_1.png')]
    tls.visualize_rotated_luminance(image, True, positions, paths)
    rotated = numpy.rot90(image, k=3)
    assert numpy.array_equal(numpy.asarray(PIL.Image.open(paths[0])), rotated)
    assert numpy.array_equal(numpy.asarray(PIL.Image.open(paths[2])), tls.crop_repeat_2d(rotated.copy(), 10, 15))
    with pytest.raises(ValueError):
        tls.visualize_crops(image, positions[:1], paths[1:])
    with pytest.raises(ValueError):
        tls.visualize_crops(image, positions, paths[:1])
    with pytest.raises(TypeError):
        tls.save_image(paths[0], image.astype(numpy.int32))


def test_e3_closed_form_equals_the_reference_loop_for_every_interval():
    """coder_simd.hip replaces the E3 loop of BinaryArithmeticCoder.cpp:238-245 / 300-318 by a closed form (run of positions
    where `low` has a 1 and `high` a 0, capped where `high` reaches 0xBFFE because the reference's three-quarters constant is
    3 * 0x3FFF). Exhaustive over every (low, high) with top bits 0 / 1 -- the state after E1/E2 -- in which E3 can start, and
    over a code register value per pair: same count, same low / high / code as the loop."""
    (quarter, three_quarters, top) = (0x3FFF, 49149, 0x8000)
    highs = numpy.arange(0x8000, 0x10000, dtype=numpy.int64)
    rng = numpy.random.RandomState(5)
    checked = 0
    for low0 in range(0x4000, 0x8000, 37):                      # every 37th `low` (443 values) x all 32768 `high`
        low = numpy.full(highs.shape, low0, dtype=numpy.int64)
        high = highs.copy()
        code = rng.randint(0, 0x10000, size=highs.shape).astype(numpy.int64)
        stream = rng.randint(0, 2, size=(15,) + highs.shape).astype(numpy.int64)       # the next stream bits, in order
        # the reference loop
        (l, h, c) = (low.copy(), high.copy(), code.copy())
        count = numpy.zeros(highs.shape, dtype=numpy.int64)
        for step in range(15):
            go = (l > quarter) & (h <= three_quarters)
            taken = numpy.take_along_axis(stream, numpy.minimum(count, 14)[None], axis=0)[0]
            l = numpy.where(go, ((l - (quarter + 1)) << 1) & 0xFFFF, l)
            h = numpy.where(go, (((h - (quarter + 1)) << 1) | 1) & 0xFFFF, h)
            c = numpy.where(go, ((((c - (quarter + 1)) << 1) & 0xFFFF) | taken), c)
            count += go
        assert not ((l > quarter) & (h <= three_quarters)).any()
        # the closed form
        run_bits = (low & ~high) & 0x7FFF
        run = numpy.zeros(highs.shape, dtype=numpy.int64)
        alive = numpy.ones(highs.shape, dtype=bool)
        for position in range(14, -1, -1):
            alive &= ((run_bits >> position) & 1) == 1
            run += alive
        trailing_ones = numpy.zeros(highs.shape, dtype=numpy.int64)
        alive = numpy.ones(highs.shape, dtype=bool)
        for position in range(16):
            alive &= ((high >> position) & 1) == 1
            trailing_ones += alive
        k = numpy.where((high > three_quarters) | (low <= quarter), 0, numpy.minimum(run, 14 - trailing_ones))
        new_bits = numpy.zeros(highs.shape, dtype=numpy.int64)
        for step in range(15):
            new_bits = numpy.where(step < k, (new_bits << 1) | stream[step], new_bits)
        assert numpy.array_equal(k, count)
        assert numpy.array_equal((((low - top) << k) + top) & 0xFFFF, l)
        assert numpy.array_equal((((high - top) << k) + top + ((1 << k) - 1)) & 0xFFFF, h)
        assert numpy.array_equal((((code - top) << k) + top + new_bits) & 0xFFFF, c)
        checked += highs.size
    assert checked > 14_000_000


def test_numpy_mean_over_leading_axes_is_a_row_by_row_float32_sum():
    """What eae_hip_map_means reproduces (lossless/stats.py:306): numpy reduces the leading axes of a C-contiguous float32
    array into one float32 accumulator per map, row after row, and divides by the float32 count. If a numpy release ever
    changed that order this test, not the GPU test, says so."""
    rng = numpy.random.RandomState(0)
    for shape in ((5, 32, 48, 128), (3, 7, 5, 128), (1, 1, 1, 128), (2, 64, 64, 128)):
        y = (rng.standard_normal(shape)*3 + 0.7).astype(numpy.float32)
        rows = y.reshape(-1, 128)
        acc = numpy.zeros(128, dtype=numpy.float32)
        for row in rows:
            acc = acc + row
        assert numpy.array_equal(numpy.mean(y, axis=(0, 1, 2)), acc/numpy.float32(rows.shape[0]))


def test_the_package_asks_for_its_hardware_queues_when_it_still_can():
    """GPU_MAX_HW_QUEUES: the caller's value wins; the package sets 16 while no HIP call has been made; too late otherwise."""
    import autoencoder_based_image_compression_amd as package
    env = {}
    assert package._configure_hw_queues(env, runtime_is_up=False) == (16, 'package') and env['GPU_MAX_HW_QUEUES'] == '16'
    env = {'GPU_MAX_HW_QUEUES': '8'}
    assert package._configure_hw_queues(env, runtime_is_up=False) == (8, 'caller') and env['GPU_MAX_HW_QUEUES'] == '8'
    assert package._configure_hw_queues({'GPU_MAX_HW_QUEUES': '24'}, runtime_is_up=True) == (24, 'caller')
    env = {}
    assert package._configure_hw_queues(env, runtime_is_up=True) == (4, 'runtime default') and 'GPU_MAX_HW_QUEUES' not in env
    assert package._configure_hw_queues({'GPU_MAX_HW_QUEUES': 'many'}, runtime_is_up=False) == (4, 'caller')
    assert package.HW_QUEUES[0] >= 1 and package.HW_QUEUES[1] in ('caller', 'package', 'runtime default')


def test_a_codec_keeps_no_more_busy_streams_than_the_process_has_hardware_queues():
    from autoencoder_based_image_compression_amd import codec
    # the product mode on the queues the package asks for: untouched, nothing to say
    assert codec.stream_budget(3, 6, hw_queues=16) == (3, 6, None)
    assert codec.stream_budget(3, 8, hw_queues=16, copies=1) == (3, 8, None)
    # four queues (the runtime's default): coder streams go first, down to two, then the transform streams
    (nt, nf, message) = codec.stream_budget(3, 6, hw_queues=4)
    assert (nt, nf) == (1, 2) and 'GPU_MAX_HW_QUEUES' in message and '3 transform + 6 coder' in message
    assert codec.stream_budget(3, 6, hw_queues=8)[:2] == (3, 4)
    assert codec.stream_budget(3, 6, hw_queues=6)[:2] == (3, 2)
    assert codec.stream_budget(3, 6, hw_queues=5)[:2] == (2, 2)
    assert codec.stream_budget(1, 3, hw_queues=4)[:2] == (1, 2)
    assert codec.stream_budget(1, 1, hw_queues=1) == (1, 1, None)
    for queues in range(1, 20):
        for nt0 in range(1, 6):
            for nf0 in range(1, 9):
                (nt, nf, message) = codec.stream_budget(nt0, nf0, hw_queues=queues)
                assert 1 <= nt <= nt0 and 1 <= nf <= nf0
                assert nt + nf <= max(2, queues - 1) or (nt, nf) == (1, 1)
                assert (message is None) == ((nt, nf) == (nt0, nf0))


class _FakeQueue(object):
    def __init__(self, size):
        self.size = size

    def qsize(self):
        return self.size


def _drive_wait(worker, queued, device_done_at, start):
    """Runs `_wait_sequence` for one job on a clock of the test's own: the pinned step counter turns 1 once the clock has reached
    `device_done_at`. Returns (the sleeps the worker took, the clock afterwards)."""
    import numpy
    clock = [start]
    words = numpy.zeros(1, dtype=numpy.int32)
    sleeps = []

    def sleep(seconds):
        sleeps.append(seconds)
        clock[0] += seconds
        if clock[0] >= device_done_at:
            words[0] = 1

    worker.jobs = _FakeQueue(queued)
    worker._sleep = sleep
    worker._now = lambda: clock[0]
    if start >= device_done_at:
        words[0] = 1
    worker._wait_sequence(words, (1,))
    return (sleeps, clock[0])


def test_the_result_worker_sleeps_long_only_in_a_steady_regime():
    """codec._Worker._wait_sequence: with jobs queued behind the one it waits for (a pipelined run) or one job at a time for a while
    (a caller that waits for every result), the first sleep is three quarters of what the wait has lately been and short polls
    follow; a job that finds nothing queued after a pipelined run (its last batches) is polled from the start."""
    from autoencoder_based_image_compression_amd import codec
    worker = codec._Worker(1536, 128, None, -1, 0)
    poll = codec._SEQUENCE_POLL_SECONDS
    # a pipelined run: two jobs queued behind each; every step takes 3 ms
    t = 0.
    (sleeps, t) = _drive_wait(worker, 2, t + 3e-3, t)
    assert all(abs(s - poll) < 1e-12 for s in sleeps) and len(sleeps) >= 25          # nothing known yet: polls only
    assert 0. < worker._typical_wait < 3.2e-3
    for _ in range(16):                                                               # (the mean moves a quarter of the way per job)
        (sleeps, t) = _drive_wait(worker, 2, t + 3e-3, t)
    typical = worker._typical_wait
    assert 2.8e-3 < typical < 3.3e-3
    (sleeps, t) = _drive_wait(worker, 2, t + 3e-3, t)
    assert abs(sleeps[0] - 0.75*typical) < 1e-9 and 2e-3 < sleeps[0] < 3e-3           # one long sleep ...
    assert all(abs(s - poll) < 1e-12 for s in sleeps[1:]) and 1 <= len(sleeps) - 1 <= 12    # ... then a few polls
    # the run's last batches: nothing queued behind them, and they come back sooner than the mean -- no long sleep through that
    (sleeps, t) = _drive_wait(worker, 0, t + 1e-3, t)
    assert all(abs(s - poll) < 1e-12 for s in sleeps) and 8 <= len(sleeps) <= 12
    # one step at a time (the caller waits for each result): after a few such jobs the long sleep is back, with polls twice as dense
    for _ in range(5):
        (sleeps, t) = _drive_wait(worker, 0, t + 1e-3, t)
    assert worker._alone >= 4
    assert 0.5e-3 < sleeps[0] < 1e-3 and all(abs(s - 0.5*poll) < 1e-12 for s in sleeps[1:])
    # a step that is through before the worker looks: no sleep at all
    (sleeps, t) = _drive_wait(worker, 2, t, t)
    assert sleeps == []


def _drive_caller_wait(worker, device_done_at, start):
    """`_wait_sequence_by_caller` on a clock of the test's own that also moves a microsecond every time it is read (the spin reads it).
    Returns (sleeps, reads of the clock while spinning, the clock afterwards)."""
    import numpy
    clock = [start]
    words = numpy.zeros(1, dtype=numpy.int32)
    sleeps = []
    reads = [0]

    def sleep(seconds):
        sleeps.append(seconds)
        clock[0] += seconds
        if clock[0] >= device_done_at:
            words[0] = 1

    def now():
        reads[0] += 1
        clock[0] += 1e-6
        if clock[0] >= device_done_at:
            words[0] = 1
        return clock[0]

    worker._sleep = sleep
    worker._now = now
    worker._wait_sequence_by_caller(words, (1,))
    return (sleeps, reads[0], clock[0])


def test_a_caller_that_waits_for_its_own_step_sleeps_once_and_then_watches_the_counter():
    """codec._Worker._wait_sequence_by_caller (`Ticket.result()` on a step nobody has started on): nothing known -> the counter is
    watched for the spin budget, then polled; after three waits that agree, ONE sleep up to a third of the budget before the shortest of
    them and the counter is watched from there (no poll when the step is on time); waits that stop agreeing switch the sleep off."""
    from autoencoder_based_image_compression_amd import codec
    worker = codec._Worker(1536, 128, None, -1, 0)
    (spin, poll) = (codec._RESULT_SPIN_SECONDS, codec._SEQUENCE_POLL_SECONDS)
    t = 0.
    (sleeps, reads, t) = _drive_caller_wait(worker, t + 1e-3, t)
    assert sleeps and all(abs(x - 0.5*poll) < 1e-12 for x in sleeps) and len(sleeps) >= 10     # spun through the budget, then polls
    for _ in range(2):
        (sleeps, reads, t) = _drive_caller_wait(worker, t + 1e-3, t)
    assert len(worker._caller_waits) == 3 and max(worker._caller_waits) < 1.1e-3
    (sleeps, reads, t) = _drive_caller_wait(worker, t + 1e-3, t)
    assert len(sleeps) == 1 and abs(sleeps[0] - (min(worker._caller_waits[:2] + (1e-3,)) - spin/3.)) < 1e-4 and 0.8e-3 < sleeps[0] < 1e-3
    assert reads*1e-6 <= spin                                                                    # the rest of the wait: reads of the counter, under the budget
    # a step 8 % slower than the last ones: still inside the budget (a sleep, then the counter)
    (sleeps, reads, t) = _drive_caller_wait(worker, t + 1.08e-3, t)
    assert len(sleeps) == 1
    # another regime (steps of 3 ms): the first one is polled to its end, and while the last three waits disagree nothing sleeps long
    (sleeps, reads, t) = _drive_caller_wait(worker, t + 3e-3, t)
    assert len(sleeps) > 10 and sleeps[0] < 1e-3
    (sleeps, reads, t) = _drive_caller_wait(worker, t + 3e-3, t)
    assert all(abs(x - 0.5*poll) < 1e-12 for x in sleeps)
    # a step that is through when the caller looks: neither sleep nor spin
    (sleeps, reads, t) = _drive_caller_wait(worker, t, t)
    assert sleeps == [] and reads <= 4


def test_the_part_of_the_results_that_does_not_wait_for_the_coder_is_formed_while_the_coder_runs():
    """`_wait_sequence_by_caller(..., early=)`: once the regime is steady, the synthesis side's counter (the last word) is polled
    during the long sleep and `early` runs as soon as it has been reached -- well before the coder's counter moves --; without a
    steady regime the wait does not call it (`process` does, behind the waits)."""
    import numpy
    from autoencoder_based_image_compression_amd import codec
    worker = codec._Worker(1536, 128, None, -1, 0)

    def drive(synthesis_at, coder_at, start):
        clock = [start]
        words = numpy.zeros(2, dtype=numpy.int32)
        called = []

        def tick():
            if clock[0] >= synthesis_at:
                words[1] = 1
            if clock[0] >= coder_at:
                words[0] = 1

        def sleep(seconds):
            clock[0] += seconds
            tick()

        def now():
            clock[0] += 1e-6
            tick()
            return clock[0]

        worker._sleep = sleep
        worker._now = now
        worker._wait_sequence_by_caller(words, (1, 1), early=lambda: called.append(clock[0]))
        return (called, clock[0])

    t = 0.
    for _ in range(3):                                        # nothing known yet: the wait never calls `early`
        (called, t) = drive(t + 0.4e-3, t + 1e-3, t)
        assert called == []
    (called, t2) = drive(t + 0.4e-3, t + 1e-3, t)             # steady: called once, between the two reports
    assert len(called) == 1 and t + 0.4e-3 <= called[0] < t + 0.7e-3 and t2 >= t + 1e-3
    t = t2
    (called, t2) = drive(t + 0.99e-3, t + 1e-3, t)            # the synthesis side reports inside the spin: left to `process`
    assert called == []


def test_whoever_claims_a_step_first_forms_its_results_and_nobody_does_it_twice():
    """`Ticket.result()` on a step the worker has not taken off its queue claims it and forms the results on the calling thread; the
    worker then finds the job claimed and leaves it alone."""
    import threading
    import numpy
    from autoencoder_based_image_compression_amd import codec

    class View(object):
        def __init__(self, array):
            self.array = array

        def numpy(self):
            return self.array

    worker = codec._Worker(1536, 128, None, -1, 0)            # never started: whatever gets done is done by the caller
    words = numpy.zeros(2, dtype=numpy.int32)
    results = numpy.zeros((4, 2*128), dtype=numpy.int32)
    results[0] = 100
    results[1] = 3
    views = [View(results), View(numpy.zeros((0,), dtype=numpy.int32)), View(numpy.zeros(2, dtype=numpy.int32)),
             View(numpy.ones((2, 128), dtype=numpy.int32)), View(numpy.zeros(1, dtype=numpy.int32)), View(numpy.array([7, 9, 0], dtype=numpy.int64))]
    ticket = codec.Ticket(2)
    slot_free = threading.Event()
    job = codec._Job(ticket, (), views, None, slot_free, None, None, (words, (1, 1)))
    ticket._job = (job, worker)
    worker.jobs.put(job)
    threading.Timer(0.02, lambda: words.__setitem__(slice(None), 1)).start()      # "the device" reports 20 ms from now
    values = ticket.result()
    assert slot_free.is_set() and ticket._job is None
    assert values['nb_bits'].tolist() == [128*103, 128*103] and values['sse'].tolist() == [7, 9] and values['nb_deads'].tolist() == [0, 0]
    # the worker comes to the job afterwards: claimed, so not processed again
    calls = []
    worker.process = lambda job, by_caller=False: calls.append(job)
    worker.jobs.put(None)
    worker.run()
    assert calls == [] and ticket.result() is values
    # and the other way round: the worker first, the caller only waits
    ticket2 = codec.Ticket(2)
    job2 = codec._Job(ticket2, (), views, None, threading.Event(), None, None, (words, (1, 1)))
    ticket2._job = (job2, worker)
    del worker.process
    worker.jobs.put(job2)
    worker.jobs.put(None)
    worker.run()
    assert ticket2._done.is_set() and not job2.claim()
    assert ticket2.result()['sse'].tolist() == [7, 9]


def test_a_step_the_device_never_reports_fails_the_codec(monkeypatch):
    """The sequence wait gives up after EAE_WORKER_SEQUENCE_TIMEOUT_SECONDS with StepTimeout (the codec then refuses further batches:
    nothing says the device is through with the slot's buffers)."""
    from autoencoder_based_image_compression_amd import codec
    worker = codec._Worker(1536, 128, None, -1, 0)
    monkeypatch.setattr(codec, '_SEQUENCE_TIMEOUT_SECONDS', 0.01)
    with pytest.raises(codec.StepTimeout):
        _drive_wait(worker, 0, 1e9, 0.)
    assert issubclass(codec.StepTimeout, RuntimeError)


def test_an_unknown_wait_mode_is_refused_at_import():
    import subprocess
    import sys
    code = 'import autoencoder_based_image_compression_amd.codec'
    env = dict(os.environ, EAE_WORKER_WAIT='evnets')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    done = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, cwd=root)
    assert done.returncode != 0 and 'EAE_WORKER_WAIT' in done.stderr
