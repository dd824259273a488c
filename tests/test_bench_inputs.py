"""bench.py takes real inputs (--kodak-npy, --checkpoint, --stats-dir) with the reference's own file contracts
(datasets/kodak/kodak.py:66-83; eae/graph/EntropyAutoencoder.py:452-458; lossless/stats.py:243-320). Parsing and loading run here without a GPU."""
import os
import pickle
import sys

import numpy
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope='module')
def bench():
    import bench as module
    return module


def _files(tmp_path, bench):
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    kodak = numpy.random.RandomState(0).randint(16, 236, size=(24, 512, 768)).astype(numpy.uint8)
    numpy.save(str(tmp_path/'kodak.npy'), kodak)
    variables = bench.synthetic_model(1.)
    variables[var.BIN_WIDTHS_NAME] = numpy.linspace(0.8, 1.6, 128).astype(numpy.float32)
    var.save_variables(str(tmp_path/'model_10.npz'), variables)
    stats = tmp_path/'training_index_10'
    stats.mkdir()
    numpy.save(str(stats/'map_mean.npy'), numpy.arange(128, dtype=numpy.float32)/64.)
    with open(str(stats/'idx_map_exception.pkl'), 'wb') as f:
        pickle.dump(67, f, protocol=2)
    numpy.save(str(stats/'binary_probabilities_1dot25.npy'), numpy.full((128, 10), 0.5))
    return (kodak, variables, str(stats))


def test_defaults_are_synthetic(bench):
    args = bench.parse_args([])
    assert (args.kodak_npy, args.checkpoint, args.stats_dir) == (None, None, None)
    inputs = bench.load_inputs(args)
    assert inputs['data'] == 'synthetic' and inputs['images'] is None and inputs['variables'] is None and inputs['statistics'] is None


def test_real_inputs_are_loaded_with_the_reference_contracts(tmp_path, bench):
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    (kodak, variables, stats) = _files(tmp_path, bench)
    args = bench.parse_args(['--kodak-npy', str(tmp_path/'kodak.npy'), '--checkpoint', str(tmp_path/'model_10.ckpt'), '--stats-dir', stats,
                             '--bin-width', '1.25'])
    inputs = bench.load_inputs(args)
    assert numpy.array_equal(inputs['images'], kodak) and inputs['images'].flags.c_contiguous
    # the multiplier scales the checkpoint's own bin widths in float32 (reconstructing_eae_kodak.py:182-184)
    assert numpy.array_equal(inputs['variables'][var.BIN_WIDTHS_NAME], (1.25*variables[var.BIN_WIDTHS_NAME]).astype(numpy.float32))
    assert numpy.array_equal(inputs['variables']['encoder/weights_2'], variables['encoder/weights_2'])
    (map_mean, probabilities, idx) = inputs['statistics']
    assert idx == 67 and probabilities.shape == (128, 10) and probabilities.dtype == numpy.float64 and map_mean.dtype == numpy.float32
    assert 'kodak.npy' in inputs['data'] and 'model_10.ckpt' in inputs['data'] and 'synthetic' not in inputs['data']
    ctx = bench.Context(args, None, 1, 0, 1)
    ctx.inputs = inputs
    assert numpy.array_equal(ctx.images(5, 24, 512, 768), kodak) and numpy.array_equal(ctx.images(5, 1, 512, 768), kodak[:1])
    assert numpy.array_equal(ctx.images(5, 2, 64, 96), bench.synthetic_images(5, 2, 64, 96))        # another shape: synthetic
    assert ctx.statistics(None, None)[2] == 67


def test_bad_inputs_are_refused(tmp_path, bench):
    numpy.save(str(tmp_path/'float.npy'), numpy.zeros((2, 32, 48), dtype=numpy.float32))
    numpy.save(str(tmp_path/'odd.npy'), numpy.zeros((2, 30, 48), dtype=numpy.uint8))
    for name in ('float.npy', 'odd.npy'):
        with pytest.raises(SystemExit):
            bench.load_inputs(bench.parse_args(['--kodak-npy', str(tmp_path/name)]))
    with pytest.raises(IOError):
        bench.load_inputs(bench.parse_args(['--checkpoint', str(tmp_path/'absent.ckpt')]))
    (_, _, stats) = _files(tmp_path, bench)
    with pytest.raises(IOError):      # the directory holds the table of multiplier 1.25 only
        bench.load_inputs(bench.parse_args(['--stats-dir', stats, '--bin-width', '2.0']))


def test_the_exception_index_file_is_read_as_an_integer_and_nothing_else(tmp_path, bench):
    """`idx_map_exception.pkl` comes from a directory the caller names: a pickle that asks for any class or function is refused
    before anything is looked up, and so is one that holds something other than an integer."""
    import pickle
    good = tmp_path/'good.pkl'
    good.write_bytes(pickle.dumps(67, protocol=2))
    assert bench.load_pickled_int(str(good)) == 67
    numpy_int = tmp_path/'numpy_int.pkl'
    numpy_int.write_bytes(pickle.dumps(numpy.int64(5), protocol=2))           # needs numpy's reconstructor: refused
    with pytest.raises(pickle.UnpicklingError):
        bench.load_pickled_int(str(numpy_int))
    evil = tmp_path/'evil.pkl'
    evil.write_bytes(b"cos\nsystem\n(S'echo pwned > /dev/null'\ntR.")
    with pytest.raises(pickle.UnpicklingError):
        bench.load_pickled_int(str(evil))
    text = tmp_path/'text.pkl'
    text.write_bytes(pickle.dumps('67', protocol=2))
    with pytest.raises(SystemExit):
        bench.load_pickled_int(str(text))
