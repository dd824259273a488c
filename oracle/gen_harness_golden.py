"""Fixture for SURVEY.md 8a28 from the reference's OWN loops -- TEST INFRASTRUCTURE ONLY, runs in the build container.

    python oracle/gen_harness_golden.py      (needs /root/reference and `make -C oracle oracle ref`)

Imports /root/reference/kodak_tensorflow/reconstructing_eae_kodak.py itself and calls its `fix_gamma` (:31-243, lossless and
approximate rate, fixed and learned bin widths) and `vary_gamma_fix_bin_widths` (:401-556) on the seeded cases of
tests/harness_cases.py. Everything the script does between the transforms is the reference's own code, imported from the
mount: `eae.batching`, `tools.tools`, `lossless.compression`, the path strings, the loops, the exception-map rule. Only what
the image lacks is shadowed:
  * `tensorflow` and `eae.graph.EntropyAutoencoder` / `IsolatedDecoder`: a Session whose `run(node, feed_dict)` evaluates the
    analysis / synthesis transform with oracle/transforms.py on the variables of a `.npz` next to the `.ckpt` path (TensorFlow
    is absent; this is the same restatement the transform tests pin);
  * `lossless.interface_cython`: the reference's own C++ coder, compiled where it lies (oracle/_ref), instead of the Cython
    build;
  * `hevc.hevc`, `jpeg2000.jpeg2000`: empty modules (external codecs, out of scope);
  * `plot_nb_dead_feature_maps`: captures the dead-map counts instead of plotting them.
Writes tests/golden/harness_golden.npz: rate / PSNR / dead maps per case. Data only; the reference never travels.
"""
import os
import sys
import tempfile
import types

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference/kodak_tensorflow'
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import harness_cases                                    # noqa: E402
from oracle import coder as oracle_coder                # noqa: E402
from oracle import transforms as T                      # noqa: E402

BIN_WIDTHS = 'piecewise_linear_function/bin_widths'


def install_shadows():
    numpy.float = numpy.floating                        # numpy >= 1.24 removed it; tools.py:91,124 use it
    os.environ.setdefault('MPLBACKEND', 'Agg')
    sys.path.insert(0, REF)

    class Node(object):
        pass

    class Session(object):
        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

        def run(self, node, feed_dict=None):
            (model, kind) = node.owner
            (key, value) = next(iter(feed_dict.items()))
            if kind == 'y':
                assert key is model.node_visible_units and value.dtype == numpy.float32
                return T.encoder(value, model.variables, model.are_bin_widths_learned)
            assert key is model.node_quantized_y
            return T.decoder(value, model.variables, model.are_bin_widths_learned)

    fake_tf = types.ModuleType('tensorflow')
    fake_tf.Session = Session
    fake_tf.reset_default_graph = lambda: None
    sys.modules['tensorflow'] = fake_tf

    def load_variables(path_to_restore):
        with numpy.load(path_to_restore.replace('.ckpt', '.npz')) as archive:
            return {name: archive[name] for name in archive.files}

    class EntropyAutoencoder(object):
        def __init__(self, batch_size, h_in, w_in, bin_width_init, gamma_scaling, path_to_nb_itvs_per_side_load, are_bin_widths_learned):
            assert os.path.isfile(path_to_nb_itvs_per_side_load)
            self.are_bin_widths_learned = are_bin_widths_learned
            (self.node_visible_units, self.node_y) = (Node(), Node())
            self.node_y.owner = (self, 'y')

        def initialization(self, sess, path_to_restore):
            self.variables = load_variables(path_to_restore)

        def get_bin_widths(self):
            return self.variables[BIN_WIDTHS]

    class IsolatedDecoder(object):
        def __init__(self, batch_size, h_in, w_in, are_bin_widths_learned):
            self.are_bin_widths_learned = are_bin_widths_learned
            (self.node_quantized_y, self.node_reconstruction) = (Node(), Node())
            self.node_reconstruction.owner = (self, 'reconstruction')

        def initialization(self, sess, path_to_restore):
            self.variables = load_variables(path_to_restore)

    for (name, cls) in (('eae.graph.EntropyAutoencoder', EntropyAutoencoder), ('eae.graph.IsolatedDecoder', IsolatedDecoder)):
        module = types.ModuleType(name)
        setattr(module, cls.__name__, cls)
        sys.modules[name] = module
    for name in ('hevc', 'hevc.hevc', 'jpeg2000', 'jpeg2000.jpeg2000'):
        sys.modules[name] = types.ModuleType(name)
    ref = oracle_coder.CoderLib('ref')
    cython = types.ModuleType('lossless.interface_cython')
    cython.compress_lossless_flattened_map = lambda ref_map_int16, probabilities: ref.compress_lossless(ref_map_int16, probabilities)
    import lossless
    sys.modules['lossless.interface_cython'] = cython
    lossless.interface_cython = cython
    import reconstructing_eae_kodak as reference_script
    assert reference_script.__file__.startswith(REF)
    return reference_script


def main():
    script = install_shadows()
    captured = {}
    script.plot_nb_dead_feature_maps = lambda rate, array_nb_deads, paths: captured.__setitem__('nb_deads', array_nb_deads.copy())
    out = {}
    positions_top_left = numpy.zeros((2, 0), dtype=numpy.int32)      # no crops: they need images larger than 80 pixels
    here = os.getcwd()
    for learned in (False, True):
        case = harness_cases.fix_gamma_case(learned)
        with tempfile.TemporaryDirectory() as root:
            (model_dir, _) = harness_cases.write_model_files(root, case['suffix'], case['idx_training'], case['map_mean'],
                                                             case['idx_map_exception'], case['probabilities'], case['multipliers'])
            numpy.savez(os.path.join(model_dir, 'model_{}.npz'.format(case['idx_training'])), **case['variables'])
            os.chdir(root)                              # the reference's paths are relative to the working directory
            try:
                for is_lossless in (True, False):
                    (rate, psnr) = script.fix_gamma(case['images'], case['bin_width_init'], case['multipliers'], case['idx_training'],
                                                    case['gamma_scaling'], case['batch_size'], learned, is_lossless,
                                                    os.path.join(root, 'checking_r'), [1], positions_top_left)
                    tag = 'fix_gamma_{0}_{1}'.format('learned' if learned else 'fixed', 'lossless' if is_lossless else 'approx')
                    out[tag + '_rate'] = rate
                    out[tag + '_psnr'] = psnr
                    out[tag + '_nb_deads'] = captured.pop('nb_deads')
            finally:
                os.chdir(here)
    case = harness_cases.vary_gamma_case()
    with tempfile.TemporaryDirectory() as root:
        for (suffix, variables) in zip(case['suffixes'], case['variables']):
            (model_dir, _) = harness_cases.write_model_files(root, suffix, 10, case['map_mean'], case['idx_map_exception'],
                                                             case['probabilities'], [1.])
            numpy.savez(os.path.join(model_dir, 'model_10.npz'), **variables)
        os.chdir(root)
        try:
            (rate, psnr) = script.vary_gamma_fix_bin_widths(case['images'], case['bin_width_init'], case['idxs_training'],
                                                            case['gammas_scaling'], case['batch_size'], os.path.join(root, 'checking_r'),
                                                            [], positions_top_left)
        finally:
            os.chdir(here)
        out['vary_gamma_rate'] = rate
        out['vary_gamma_psnr'] = psnr
    path = os.path.join(ROOT, 'tests', 'golden', 'harness_golden.npz')
    numpy.savez_compressed(path, **out)
    for (key, value) in sorted(out.items()):
        print(key, value.shape, value.dtype, float(numpy.mean(value)))


if __name__ == '__main__':
    main()
