"""Inputs of the rate-distortion harness tests, shared by the fixture generator (oracle/gen_harness_golden.py, which runs the
REFERENCE's own `fix_gamma` / `vary_gamma_fix_bin_widths` on them in the build container) and by tests/test_gpu_surface.py
(which runs this build's mirror of those functions on the GPU and compares with the committed outputs). Everything is
derived from seeds and from tests/golden/coder_golden.npz; nothing here needs the reference."""
import os
import pickle

import numpy

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def float_to_str(value):
    """tls.float_to_str (tools.py: '1.0' -> '1', '0.5' -> '0dot5')."""
    text = str(value)
    if text.endswith('.0'):
        text = text[:-2]
    return text.replace('.', 'dot').replace('-', 'minus')


def random_variables(bin_width, learned, seed):
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    v = var.random_variables(bin_width, learned, seed=seed, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)     # leave the clip floor
    return v


def fix_gamma_case(learned):
    """One trained model, three multipliers, four 32x48 images."""
    with numpy.load(os.path.join(GOLDEN, 'coder_golden.npz')) as g:
        map_mean = g['real_map_mean'].astype(numpy.float32)*numpy.float32(0.1)
        probabilities = g['real_probabilities_2'].copy()
    rng = numpy.random.RandomState(32)
    x = rng.randint(16, 236, size=(4, 32, 48)).astype(numpy.float64)
    x = numpy.round((x + numpy.roll(x, 1, 1) + numpy.roll(x, 1, 2))/3.).astype(numpy.uint8)
    bin_width_init = 0.5 if learned else 1.
    return {'variables': random_variables(bin_width_init, learned, 31), 'images': x,
            'multipliers': numpy.array([1., 1.25, 4.], dtype=numpy.float32), 'map_mean': map_mean, 'probabilities': probabilities,
            'idx_map_exception': 67, 'bin_width_init': bin_width_init, 'gamma_scaling': 10000., 'idx_training': 10, 'batch_size': 2,
            'suffix': 'learning_bw_0dot5_10000' if learned else '1_10000', 'learned': learned}


def vary_gamma_case():
    """Two fixed-bin-width models trained with different scaling coefficients, two 16x32 images."""
    with numpy.load(os.path.join(GOLDEN, 'coder_golden.npz')) as g:
        probabilities = g['real_probabilities_1'].copy()
    return {'variables': [random_variables(1., False, 41), random_variables(1., False, 42)],
            'images': numpy.random.RandomState(43).randint(16, 236, size=(2, 16, 32)).astype(numpy.uint8),
            'gammas_scaling': numpy.array([10000., 12000.]), 'idxs_training': numpy.array([10, 10], dtype=numpy.int32),
            'map_mean': numpy.zeros(128, dtype=numpy.float32), 'probabilities': probabilities, 'idx_map_exception': 67,
            'bin_width_init': 1., 'batch_size': 2, 'suffixes': ['1_10000', '1_12000']}


def write_model_files(root, suffix, idx_training, map_mean, idx_map_exception, probabilities, multipliers):
    """The files around a model that `fix_gamma` reads relative to `root` (reconstructing_eae_kodak.py:88-93, 170-176, 201-202),
    except the model itself. Returns (directory of the model, directory of the statistics)."""
    model_dir = os.path.join(root, 'eae/results', suffix)
    stats_dir = os.path.join(root, 'lossless/results', suffix, 'training_index_{}'.format(idx_training))
    os.makedirs(model_dir, exist_ok=True)
    os.makedirs(stats_dir, exist_ok=True)
    with open(os.path.join(model_dir, 'nb_itvs_per_side_{}.pkl'.format(idx_training)), 'wb') as f:
        pickle.dump(91, f, protocol=2)
    numpy.save(os.path.join(stats_dir, 'map_mean.npy'), map_mean)
    with open(os.path.join(stats_dir, 'idx_map_exception.pkl'), 'wb') as f:
        pickle.dump(idx_map_exception, f, protocol=2)
    for m in multipliers:
        numpy.save(os.path.join(stats_dir, 'binary_probabilities_{}.npy'.format(float_to_str(float(m)))), probabilities)
    return (model_dir, stats_dir)
