"""numpy-facing view of oracle/svhn_oracle.c and the SVHN encoder / decoder compositions -- TEST INFRASTRUCTURE ONLY.
Follows svhn/eae/EntropyAutoencoder.py:218-278 of the reference."""
import ctypes
import os

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def dense(x, w, b, leaky):
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(os.path.join(_HERE, '_build', 'liboracle_svhn.so'))
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.orc_dense_f64.restype = None
        _lib.orc_dense_f64.argtypes = [dp, dp, dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    x = numpy.ascontiguousarray(x, dtype=numpy.float64)
    w = numpy.ascontiguousarray(w, dtype=numpy.float64)
    b = numpy.ascontiguousarray(b, dtype=numpy.float64).reshape(-1)
    out = numpy.empty((x.shape[0], w.shape[1]), dtype=numpy.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    _lib.orc_dense_f64(x.ctypes.data_as(dp), w.ctypes.data_as(dp), b.ctypes.data_as(dp), out.ctypes.data_as(dp),
                       x.shape[0], x.shape[1], w.shape[1], 1 if leaky else 0)
    return out


def encoder(x, p):
    hidden = dense(x, p['weights_encoder']['l1'], p['biases_encoder']['l1'], True)
    return hidden, dense(hidden, p['weights_encoder']['latent'], p['biases_encoder']['latent'], False)


def decoder(y, p):
    hidden = dense(y, p['weights_decoder']['l1'], p['biases_decoder']['l1'], True)
    return hidden, dense(hidden, p['weights_decoder']['mean'], p['biases_decoder']['mean'], False)
