"""Mini-batch loops of the reference (kodak_tensorflow/eae/batching.py:11-100), same signatures.

`encode_mini_batches` :56-100 and `decode_mini_batches` :11-54. The training loops (:102-166) are out of scope.
Each `sess.run` moves one mini-batch host -> HBM -> host like a TF session with a GPU device would; the
device-resident path without those copies is `DeviceEncoder` / `DeviceDecoder` (pipeline.py), used by bench.py.
"""
import numpy

from .graph import constants as csts
from ..tools import tools as tls

# The functions are sorted in alphabetic order.


def decode_mini_batches(quantized_y_float32, sess, isolated_decoder, batch_size):
    """Reconstruction of the luminance images from the quantized latent variables, one mini-batch at a time.

    Parameters and return value as in the reference (:11-39): float32 (N, h, w, 128) -> uint8 (N, 16h, 16w, 1).
    """
    (nb_images, h_in, w_in, _) = quantized_y_float32.shape
    nb_batches = tls.subdivide_set(nb_images, batch_size)
    expanded_reconstruction_uint8 = numpy.zeros((nb_images, h_in*csts.STRIDE_PROD, w_in*csts.STRIDE_PROD, 1), dtype=numpy.uint8)
    for i in range(nb_batches):
        reconstruction_float32 = sess.run(
            isolated_decoder.node_reconstruction,
            feed_dict={isolated_decoder.node_quantized_y: quantized_y_float32[i*batch_size:(i + 1)*batch_size, :, :, :]}
        )
        expanded_reconstruction_uint8[i*batch_size:(i + 1)*batch_size, :, :, :] = tls.cast_bt601(reconstruction_float32)
    return expanded_reconstruction_uint8


def encode_mini_batches(luminances_uint8, sess, entropy_ae, batch_size):
    """Latent variables of the luminance images, one mini-batch at a time.

    Parameters and return value as in the reference (:56-84): uint8 (N, H, W, 1) -> float32 (N, H/16, W/16, 128).

    Raises
    ------
    TypeError
        If `luminances_uint8.dtype` is not equal to `numpy.uint8`.
    """
    if luminances_uint8.dtype != numpy.uint8:
        raise TypeError('`luminances_uint8.dtype` is not equal to `numpy.uint8`.')
    (nb_images, h_in, w_in, _) = luminances_uint8.shape
    nb_batches = tls.subdivide_set(nb_images, batch_size)
    y_float32 = numpy.zeros((nb_images, h_in//csts.STRIDE_PROD, w_in//csts.STRIDE_PROD, csts.NB_MAPS_3), dtype=numpy.float32)
    for i in range(nb_batches):
        batch_float32 = luminances_uint8[i*batch_size:(i + 1)*batch_size, :, :, :].astype(numpy.float32)
        y_float32[i*batch_size:(i + 1)*batch_size, :, :, :] = sess.run(
            entropy_ae.node_y,
            feed_dict={entropy_ae.node_visible_units: batch_float32}
        )
    return y_float32
