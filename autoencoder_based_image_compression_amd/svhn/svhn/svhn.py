"""`preprocess_svhn` of the reference (svhn/svhn/svhn.py:170-210): (uint8 - mean)/std in float64, on the device."""
import numpy

from ... import device as dev
from ...kodak import _backend as bk


def preprocess_svhn(images_uint8, mean_training, std_training):
    """images_uint8 (nb_images, 3072) uint8; mean_training (1, 3072) float64; std_training float -> float64."""
    if images_uint8.dtype != numpy.uint8:
        raise TypeError('`images_uint8.dtype` is not equal to `numpy.uint8`.')
    if images_uint8.ndim != 2:
        raise ValueError('`images_uint8.ndim` is not equal to 2.')
    mean = numpy.ascontiguousarray(mean_training, dtype=numpy.float64).reshape(-1)
    return bk.to_host(dev.svhn_preprocess(bk.to_device(images_uint8), bk.to_device(mean), float(std_training)))
