"""numpy-facing view of oracle/transforms_oracle.c plus the encoder / decoder compositions -- TEST INFRASTRUCTURE ONLY.

`encoder` / `decoder` follow kodak_tensorflow/eae/graph/components.py:86-142 / 11-84 op by op; variable names and
layouts are the TF ones.
"""
import ctypes
import os

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(os.path.join(_HERE, '_build', 'liboracle_transforms.so'))
        fp = ctypes.POINTER(ctypes.c_float)
        _lib.orc_conv2d_same.restype = None
        _lib.orc_conv2d_same.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, fp, fp]
        _lib.orc_conv2d_transpose_same.restype = None
        _lib.orc_conv2d_transpose_same.argtypes = _lib.orc_conv2d_same.argtypes
        _lib.orc_conv2d_transpose_same_col2im.restype = None
        _lib.orc_conv2d_transpose_same_col2im.argtypes = _lib.orc_conv2d_same.argtypes
        _lib.orc_gdn.restype = None
        _lib.orc_gdn.argtypes = [fp, ctypes.c_int64, ctypes.c_int, fp, fp, ctypes.c_int, fp]
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if a is not None else None


def _f32(a):
    return numpy.ascontiguousarray(a, dtype=numpy.float32)


def conv2d_same(x, w, stride, bias=None):
    """tf.nn.conv2d(x, w, [1,s,s,1], 'SAME') (+ tf.nn.bias_add). x NHWC f32, w HWIO."""
    x = _f32(x)
    w = _f32(w)
    bias = _f32(bias) if bias is not None else None
    (n, h, wd, cin) = x.shape
    (k, k2, cin2, cout) = w.shape
    assert k == k2 and cin == cin2
    out = numpy.empty((n, -(-h//stride), -(-wd//stride), cout), dtype=numpy.float32)
    lib().orc_conv2d_same(_fp(x), n, h, wd, cin, _fp(w), k, stride, cout, _fp(bias), _fp(out))
    return out


def conv2d_transpose_same(x, w, stride, bias=None, col2im=False):
    """tf.nn.conv2d_transpose(x, w[k,k,cout,cin], [n, s*h, s*w, cout], [1,s,s,1], 'SAME') (+ bias).

    `col2im=False`: every output element one fmaf chain (channel block, u, v, channel). `col2im=True`: the order of a GEMM +
    col2im implementation -- one 128-channel chain per (site, tap), the overlapping taps of an output pixel then added site
    by site in raster order (transforms_oracle.c: orc_conv2d_transpose_same_col2im): the order of transpose_conv_3."""
    x = _f32(x)
    w = _f32(w)
    bias = _f32(bias) if bias is not None else None
    (n, h, wd, cin) = x.shape
    (k, k2, cout, cin2) = w.shape
    assert k == k2 and cin == cin2
    out = numpy.empty((n, h*stride, wd*stride, cout), dtype=numpy.float32)
    fn = lib().orc_conv2d_transpose_same_col2im if col2im else lib().orc_conv2d_transpose_same
    fn(_fp(x), n, h, wd, cin, _fp(w), k, stride, cout, _fp(bias), _fp(out))
    return out


def gdn(x, gamma, beta, inverse=False):
    """tfuls.gdn / tfuls.inverse_gdn (tfutils.py:363-397, 480-509) on the last axis."""
    x = _f32(x)
    gamma = _f32(gamma)
    beta = _f32(beta)
    c = x.shape[-1]
    out = numpy.empty_like(x)
    lib().orc_gdn(_fp(x), x.size//c, c, _fp(gamma), _fp(beta), 1 if inverse else 0, _fp(out))
    return out


# The forward path as data: (kind, filter or gamma, stride or beta, bias) in graph order. `encoder` / `decoder` below interpret
# these tables; tests/test_oracle_graph.py holds them against the op graph of the reference's own `.ckpt.meta` files
# (tests/golden/ckpt_graph.json, extracted by oracle/gen_ckpt_graph.py): same kinds, order, variables, strides, 'SAME' / NHWC.
# A trailing True marks the normalisation that only the fixed-bin-width model has (components.py:137-142, 53-58).
ENCODER_LAYERS = (
    ('conv2d', 'encoder/weights_1', 4, 'encoder/biases_1'),
    ('gdn', 'encoder/gamma_1', 'encoder/beta_1'),
    ('conv2d', 'encoder/weights_2', 2, 'encoder/biases_2'),
    ('gdn', 'encoder/gamma_2', 'encoder/beta_2'),
    ('conv2d', 'encoder/weights_3', 2, 'encoder/biases_3'),
    ('gdn', 'encoder/gamma_3', 'encoder/beta_3', True),
)
DECODER_LAYERS = (
    ('inverse_gdn', 'decoder/gamma_4', 'decoder/beta_4', True),
    ('conv2d_transpose', 'decoder/weights_4', 2, 'decoder/biases_4'),
    ('inverse_gdn', 'decoder/gamma_5', 'decoder/beta_5'),
    ('conv2d_transpose', 'decoder/weights_5', 2, 'decoder/biases_5'),
    ('inverse_gdn', 'decoder/gamma_6', 'decoder/beta_6'),
    ('conv2d_transpose', 'decoder/weights_6', 4, None),
)


# Layers whose summation order is that of a GEMM + col2im implementation: none. Round 6 built transpose_conv_3 that way (the single
# chain per output pixel costs the gfx950 kernel 7 structural zeros in every 16 products), bit for bit against
# orc_conv2d_transpose_same_col2im, measured it and kept the round-5 kernel and order (the kernel and its logs:
# scratch/r06/, profiles/r06_tconv3_*); the restatement of that order stays here, checked against the float64 definition
# (tests/test_oracle_transforms.py), for whoever takes it up again.
COL2IM_ORDER_LAYERS = frozenset()


def layers_of(table, are_bin_widths_learned):
    """The rows of a table that the model has."""
    return tuple(row for row in table if not (are_bin_widths_learned and len(row) == 4 and row[3] is True))


def _run(x, table, variables, are_bin_widths_learned):
    outputs = []
    for row in layers_of(table, are_bin_widths_learned):
        if row[0] == 'conv2d':
            x = conv2d_same(x, variables[row[1]], row[2], variables[row[3]] if row[3] else None)
        elif row[0] == 'conv2d_transpose':
            x = conv2d_transpose_same(x, variables[row[1]], row[2], variables[row[3]] if row[3] else None,
                                      col2im=row[1] in COL2IM_ORDER_LAYERS)
        else:
            x = gdn(x, variables[row[1]], variables[row[2]], inverse=row[0] == 'inverse_gdn')
        outputs.append(x)
    return (x, outputs)


def encoder(visible_units_float32, variables, are_bin_widths_learned, return_intermediates=False):
    """components.encoder (components.py:86-142). `variables`: dict keyed by the TF variable names."""
    (y, outputs) = _run(visible_units_float32, ENCODER_LAYERS, variables, are_bin_widths_learned)
    if return_intermediates:
        return y, {'gdn_1': outputs[1], 'gdn_2': outputs[3], 'conv_3': outputs[4]}
    return y


def decoder(y_tilde, variables, are_bin_widths_learned, return_intermediates=False):
    """components.decoder (components.py:11-84)."""
    (tc3, outputs) = _run(y_tilde, DECODER_LAYERS, variables, are_bin_widths_learned)
    if return_intermediates:
        first = 0 if are_bin_widths_learned else 1          # igdn_1 is the input itself when the model has no inverse_gdn #4
        return tc3, {'igdn_1': y_tilde if are_bin_widths_learned else outputs[0], 'igdn_2': outputs[first + 1], 'igdn_3': outputs[first + 3]}
    return tc3
