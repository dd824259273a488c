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


def conv2d_transpose_same(x, w, stride, bias=None):
    """tf.nn.conv2d_transpose(x, w[k,k,cout,cin], [n, s*h, s*w, cout], [1,s,s,1], 'SAME') (+ bias)."""
    x = _f32(x)
    w = _f32(w)
    bias = _f32(bias) if bias is not None else None
    (n, h, wd, cin) = x.shape
    (k, k2, cout, cin2) = w.shape
    assert k == k2 and cin == cin2
    out = numpy.empty((n, h*stride, wd*stride, cout), dtype=numpy.float32)
    lib().orc_conv2d_transpose_same(_fp(x), n, h, wd, cin, _fp(w), k, stride, cout, _fp(bias), _fp(out))
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


def encoder(visible_units_float32, variables, are_bin_widths_learned, return_intermediates=False):
    """components.encoder (components.py:86-142). `variables`: dict keyed by the TF variable names."""
    v = variables
    conv_1 = conv2d_same(visible_units_float32, v['encoder/weights_1'], 4, v['encoder/biases_1'])
    gdn_1 = gdn(conv_1, v['encoder/gamma_1'], v['encoder/beta_1'])
    conv_2 = conv2d_same(gdn_1, v['encoder/weights_2'], 2, v['encoder/biases_2'])
    gdn_2 = gdn(conv_2, v['encoder/gamma_2'], v['encoder/beta_2'])
    conv_3 = conv2d_same(gdn_2, v['encoder/weights_3'], 2, v['encoder/biases_3'])
    y = conv_3 if are_bin_widths_learned else gdn(conv_3, v['encoder/gamma_3'], v['encoder/beta_3'])
    if return_intermediates:
        return y, {'gdn_1': gdn_1, 'gdn_2': gdn_2, 'conv_3': conv_3}
    return y


def decoder(y_tilde, variables, are_bin_widths_learned, return_intermediates=False):
    """components.decoder (components.py:11-84)."""
    v = variables
    t = y_tilde if are_bin_widths_learned else gdn(y_tilde, v['decoder/gamma_4'], v['decoder/beta_4'], inverse=True)
    tc1 = conv2d_transpose_same(t, v['decoder/weights_4'], 2, v['decoder/biases_4'])
    igdn_2 = gdn(tc1, v['decoder/gamma_5'], v['decoder/beta_5'], inverse=True)
    tc2 = conv2d_transpose_same(igdn_2, v['decoder/weights_5'], 2, v['decoder/biases_5'])
    igdn_3 = gdn(tc2, v['decoder/gamma_6'], v['decoder/beta_6'], inverse=True)
    tc3 = conv2d_transpose_same(igdn_3, v['decoder/weights_6'], 4, None)
    if return_intermediates:
        return tc3, {'igdn_1': t, 'igdn_2': igdn_2, 'igdn_3': igdn_3}
    return tc3
