"""Variables of the entropy autoencoder under their TensorFlow names, as numpy arrays in the TF layouts.

Reference: kodak_tensorflow/eae/graph/EntropyAutoencoder.py:108-224 (encoder/decoder/bin-width variables and their
initialisers) and kodak_tensorflow/eae/graph/IsolatedDecoder.py:54-97. Models are restored from the reference's own
TensorFlow checkpoints (`model_<i>.ckpt`, V1 or V2, read by `tf_checkpoint` without TensorFlow) or from a ``.npz``
keyed by the same names (the trained ``model_*.ckpt`` files are absent from the reference mount,
.MISSING_LARGE_BLOBS:1-9).
"""
import os

import numpy

from . import constants as csts

ENCODER_NAMES = ('encoder/weights_1', 'encoder/biases_1', 'encoder/gamma_1', 'encoder/beta_1',
                 'encoder/weights_2', 'encoder/biases_2', 'encoder/gamma_2', 'encoder/beta_2',
                 'encoder/weights_3', 'encoder/biases_3')
ENCODER_NAMES_FIXED_BW = ('encoder/gamma_3', 'encoder/beta_3')
DECODER_NAMES = ('decoder/weights_4', 'decoder/biases_4', 'decoder/gamma_5', 'decoder/beta_5',
                 'decoder/weights_5', 'decoder/biases_5', 'decoder/gamma_6', 'decoder/beta_6',
                 'decoder/weights_6')
DECODER_NAMES_FIXED_BW = ('decoder/gamma_4', 'decoder/beta_4')
BIN_WIDTHS_NAME = 'piecewise_linear_function/bin_widths'

SHAPES = {
    'encoder/weights_1': (9, 9, 1, 128), 'encoder/weights_2': (5, 5, 128, 128), 'encoder/weights_3': (5, 5, 128, 128),
    'decoder/weights_4': (5, 5, 128, 128), 'decoder/weights_5': (5, 5, 128, 128), 'decoder/weights_6': (9, 9, 1, 128),
    BIN_WIDTHS_NAME: (128,),
}
for _i in (1, 2, 3):
    SHAPES['encoder/biases_{}'.format(_i)] = (128,)
    SHAPES['encoder/gamma_{}'.format(_i)] = (128, 128)
    SHAPES['encoder/beta_{}'.format(_i)] = (128,)
for _i in (4, 5, 6):
    SHAPES['decoder/gamma_{}'.format(_i)] = (128, 128)
    SHAPES['decoder/beta_{}'.format(_i)] = (128,)
for _i in (4, 5):
    SHAPES['decoder/biases_{}'.format(_i)] = (128,)


def initialize_weights_gdn(nb_maps, min_gamma, rng):
    """Symmetric GDN/IGDN weights, 0.5*(U + U^T) with U ~ U[min_gamma, 0.01] (tfutils.py:445-478)."""
    if min_gamma > 0.01 or min_gamma <= 0.:
        raise ValueError('`min_gamma` does not belong to ]0., 0.01].')
    gamma_non_symmetric = rng.uniform(low=min_gamma, high=0.01, size=(nb_maps, nb_maps)).astype(numpy.float32)
    return (numpy.float32(0.5)*(gamma_non_symmetric + gamma_non_symmetric.T)).astype(numpy.float32)


def random_variables(bin_width_init, are_bin_widths_learned, seed=None, bias_std=0.):
    """Random initialisation shaped like the reference's (`initialization(sess, '')`).

    EntropyAutoencoder.py:130-224: weights ~ N(0, 0.01 / 0.02 / 0.05) for layers 1 / 2 / 3 (mirrored 0.05 / 0.02 /
    0.01 for layers 4 / 5 / 6), biases 0, gamma = initialize_weights_gdn, beta = 1, bin widths = bin_width_init.
    `bias_std` > 0 draws non-zero biases instead (tests use it so that the bias path is exercised).
    """
    rng = numpy.random.RandomState(seed)
    v = {}

    def normal(shape, std):
        return (rng.standard_normal(size=shape)*std).astype(numpy.float32)

    def bias():
        return normal((128,), bias_std) if bias_std > 0. else numpy.zeros(128, dtype=numpy.float32)
    v['encoder/weights_1'] = normal((9, 9, 1, 128), 0.01)
    v['encoder/weights_2'] = normal((5, 5, 128, 128), 0.02)
    v['encoder/weights_3'] = normal((5, 5, 128, 128), 0.05)
    v['decoder/weights_4'] = normal((5, 5, 128, 128), 0.05)
    v['decoder/weights_5'] = normal((5, 5, 128, 128), 0.02)
    v['decoder/weights_6'] = normal((9, 9, 1, 128), 0.01)
    for i in (1, 2, 3):
        v['encoder/biases_{}'.format(i)] = bias()
    for i in (4, 5):
        v['decoder/biases_{}'.format(i)] = bias()
    idx_enc = (1, 2) if are_bin_widths_learned else (1, 2, 3)
    idx_dec = (5, 6) if are_bin_widths_learned else (4, 5, 6)
    for i in idx_enc:
        v['encoder/gamma_{}'.format(i)] = initialize_weights_gdn(128, csts.MIN_GAMMA_BETA, rng)
        v['encoder/beta_{}'.format(i)] = numpy.ones(128, dtype=numpy.float32)
    for i in idx_dec:
        v['decoder/gamma_{}'.format(i)] = initialize_weights_gdn(128, csts.MIN_GAMMA_BETA, rng)
        v['decoder/beta_{}'.format(i)] = numpy.ones(128, dtype=numpy.float32)
    v[BIN_WIDTHS_NAME] = (numpy.float32(bin_width_init)*numpy.ones(128, dtype=numpy.float32)).astype(numpy.float32)
    return v


def check_variables(variables, names):
    for name in names:
        if name not in variables:
            raise KeyError('The variable "{}" is missing.'.format(name))
        array = variables[name]
        if array.dtype != numpy.float32 or tuple(array.shape) != SHAPES[name]:
            raise ValueError('The variable "{0}" must be float32 with shape {1}.'.format(name, SHAPES[name]))


def save_variables(path, variables):
    """Writes the variables to a `.npz` keyed by the TF names ('/' is kept)."""
    numpy.savez(path, **variables)


def load_variables(path):
    with numpy.load(path) as data:
        return {name: numpy.ascontiguousarray(data[name]) for name in data.files}


def model_names(are_bin_widths_learned, side='both'):
    """TF names of the variables the inference path reads: `side` = 'encoder', 'decoder' or 'both'."""
    names = []
    if side in ('encoder', 'both'):
        names += list(ENCODER_NAMES) + ([] if are_bin_widths_learned else list(ENCODER_NAMES_FIXED_BW))
    if side in ('decoder', 'both'):
        names += list(DECODER_NAMES) + ([] if are_bin_widths_learned else list(DECODER_NAMES_FIXED_BW))
    return names + [BIN_WIDTHS_NAME]


def restore_variables(path_to_restore, are_bin_widths_learned, side='both'):
    """What `tf.train.Saver().restore(sess, path_to_restore)` gives the inference path (EntropyAutoencoder.py:454-458,
    IsolatedDecoder.py:123-124): the variables of `side`, float32, in the TF layouts.

    `path_to_restore` ends with ".ckpt" like the reference's. Looked up in this order: a TensorFlow checkpoint at that
    prefix (V2 `.index` + `.data-*`, then V1 single file), then the sibling `.npz`; a path that does not end with
    ".ckpt" is an `.npz` file.

    Raises
    ------
    IOError
        If no model is found at `path_to_restore`.
    KeyError, ValueError
        If a variable is missing (e.g. a fixed-bin-width graph restored from a learned-bin-width model), or has the
        wrong dtype or shape.
    """
    from . import tf_checkpoint
    names = model_names(are_bin_widths_learned, side)
    if tf_checkpoint.exists(path_to_restore):
        variables = tf_checkpoint.load_checkpoint(path_to_restore, names=names)
    else:
        path = path_to_restore[:-5] + '.npz' if path_to_restore.endswith('.ckpt') else path_to_restore
        if not os.path.isfile(path):
            raise IOError('The model "{}" does not exist.'.format(path_to_restore))
        variables = load_variables(path)
    check_variables(variables, names)
    return {name: numpy.ascontiguousarray(variables[name]) for name in names}
