"""The two ops of the reference's `tfutils/tfutils.py` that sit on the inference path, as eager numpy -> numpy
functions on the MI355X: `gdn` :363-397, `inverse_gdn` :480-509, plus the random initialiser `initialize_weights_gdn`
:445-478. (In the reference they build TF graph nodes; here a call IS the evaluation.)"""
import numpy

from ... import device as dev
from .. import _backend as bk
from ..eae.graph.variables import initialize_weights_gdn as _initialize_weights_gdn


def _apply(input, gamma, beta, inverse):
    if input.ndim != 4:
        raise ValueError('`input.ndim` is not equal to 4.')
    nb_maps = input.shape[3]
    if nb_maps != 128 or tuple(gamma.shape) != (128, 128) or tuple(beta.shape) != (128,):
        raise ValueError('GDN/IGDN on this path has 128 feature maps: `gamma` (128, 128), `beta` (128,).')
    out = dev.gdn(bk.to_device(input, numpy.float32), dev.pack_gamma(bk.to_device(gamma, numpy.float32)),
                  bk.to_device(beta, numpy.float32), inverse=inverse)
    return bk.to_host(out)


def gdn(input, gamma, beta):
    """Generalized Divisive Normalization: input/sqrt(matmul(input**2, gamma) + beta) (tfutils.py:393-397)."""
    return _apply(input, gamma, beta, False)


def initialize_weights_gdn(nb_maps, min_gamma, seed=None):
    """0.5*(U + U^T), U ~ U[min_gamma, 0.01] (tfutils.py:445-478). ValueError if min_gamma is outside ]0, 0.01]."""
    return _initialize_weights_gdn(nb_maps, min_gamma, numpy.random.RandomState(seed))


def inverse_gdn(input, gamma, beta):
    """Inverse GDN: input*sqrt(matmul(input**2, gamma) + beta) (tfutils.py:505-509)."""
    return _apply(input, gamma, beta, True)
