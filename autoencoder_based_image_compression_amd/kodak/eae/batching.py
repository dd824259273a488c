"""Mini-batch loops with the reference's signatures (kodak_tensorflow/eae/batching.py: `decode_mini_batches` :11-54,
`encode_mini_batches` :56-100; the training loops :102-166 are out of scope).

Each `sess.run` moves one mini-batch host -> HBM -> host, like a TF session with a GPU device would; the device-resident
path without those copies is `pipeline.DeviceEncoder` / `DeviceDecoder`, and `codec.BatchCodec` is what bench.py times.
"""
import numpy

from .graph import constants as csts
from ..tools import tools as tls


def _mini_batches(nb_examples, batch_size):
    """Slices of the consecutive mini-batches; ValueError (from `tls.subdivide_set`) unless they tile the set exactly."""
    for index in range(tls.subdivide_set(nb_examples, batch_size)):
        yield slice(index*batch_size, (index + 1)*batch_size)


def decode_mini_batches(quantized_y_float32, sess, isolated_decoder, batch_size):
    """float32 quantized latents (N, h, w, 128) -> uint8 reconstructions (N, 16 h, 16 w, 1), BT.601 range (:11-54)."""
    (nb_images, h_map, w_map) = quantized_y_float32.shape[:3]
    out = numpy.zeros((nb_images, csts.STRIDE_PROD*h_map, csts.STRIDE_PROD*w_map, 1), dtype=numpy.uint8)
    for chunk in _mini_batches(nb_images, batch_size):
        fetched = sess.run(isolated_decoder.node_reconstruction,
                           feed_dict={isolated_decoder.node_quantized_y: quantized_y_float32[chunk]})
        out[chunk] = tls.cast_bt601(fetched)
    return out


def encode_mini_batches(luminances_uint8, sess, entropy_ae, batch_size):
    """uint8 luminances (N, H, W, 1) -> float32 latents (N, H/16, W/16, 128) (:56-100).

    Raises
    ------
    TypeError
        If `luminances_uint8.dtype` is not equal to `numpy.uint8`.
    """
    if luminances_uint8.dtype != numpy.uint8:
        raise TypeError('`luminances_uint8.dtype` is not equal to `numpy.uint8`.')
    (nb_images, h_in, w_in) = luminances_uint8.shape[:3]
    latents = numpy.zeros((nb_images, h_in//csts.STRIDE_PROD, w_in//csts.STRIDE_PROD, csts.NB_MAPS_3), dtype=numpy.float32)
    for chunk in _mini_batches(nb_images, batch_size):
        latents[chunk] = sess.run(entropy_ae.node_y,
                                  feed_dict={entropy_ae.node_visible_units: luminances_uint8[chunk].astype(numpy.float32)})
    return latents
