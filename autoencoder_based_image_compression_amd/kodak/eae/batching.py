"""Mini-batch loops with the reference's signatures (kodak_tensorflow/eae/batching.py: `decode_mini_batches` :11-54,
`encode_mini_batches` :56-100; the training loops :102-166 are out of scope).

Same arguments, checks, exceptions and returned arrays. With this build's own `EntropyAutoencoder` / `IsolatedDecoder` the set
goes to HBM once, consecutive mini-batches are handed together to the whole-path entry point (`eae_hip_encode` /
`eae_hip_decode`, which casts to BT.601 itself: no float32 reconstruction travels to the host and back for `tls.cast_bt601`),
and the result comes back in one copy and stays resident behind the returned array (`_backend.publish`) for the calls the
harness makes next (`tls.quantize_per_map`, `tls.psnr_2d` ...). Any other object with the two nodes goes through
`sess.run` mini-batch by mini-batch exactly like the reference. `codec.BatchCodec` is the fused path bench.py's headline times.
"""
import numpy

from .graph import constants as csts
from .. import _backend as bk
from ..tools import tools as tls


# Mini-batches exist in the reference to bound what one `sess.run` holds in memory; every image goes through the transforms on
# its own, so the latents / reconstructions do not depend on how the set is cut (tests/test_gpu_surface.py holds launches of 1,
# `batch_size` and the whole set against each other bit for bit). With this build's own model objects consecutive mini-batches
# are therefore handed to the device together, up to this many pixels per launch (24 Kodak images: 1.3 GB of activations).
_PIXELS_PER_LAUNCH = 24*512*768


def _launches(nb_examples, batch_size, pixels_per_image):
    """Slices of whole mini-batches, as many per slice as `_PIXELS_PER_LAUNCH` allows (at least one); ValueError (from
    `tls.subdivide_set`) unless the mini-batches tile the set exactly."""
    nb_batches = tls.subdivide_set(nb_examples, batch_size)
    per_launch = max(1, _PIXELS_PER_LAUNCH//max(1, batch_size*pixels_per_image))
    for first in range(0, nb_batches, per_launch):
        yield slice(first*batch_size, min(nb_batches, first + per_launch)*batch_size)


def _mini_batches(nb_examples, batch_size):
    """Slices of the consecutive mini-batches; ValueError (from `tls.subdivide_set`) unless they tile the set exactly."""
    for index in range(tls.subdivide_set(nb_examples, batch_size)):
        yield slice(index*batch_size, (index + 1)*batch_size)


def decode_mini_batches(quantized_y_float32, sess, isolated_decoder, batch_size):
    """float32 quantized latents (N, h, w, 128) -> uint8 reconstructions (N, 16 h, 16 w, 1), BT.601 range (:11-54)."""
    (nb_images, h_map, w_map) = quantized_y_float32.shape[:3]
    decode_device = getattr(isolated_decoder, 'decode_device_into', None)
    if (decode_device is not None and nb_images > 0 and
            (batch_size,) + tuple(quantized_y_float32.shape[1:]) == tuple(isolated_decoder.node_quantized_y.shape)):
        import torch
        chunks = list(_launches(nb_images, batch_size, 256*h_map*w_map))            # ValueError first, like the reference
        latents = bk.to_device(quantized_y_float32, numpy.float32)     # resident if a call of this package returned it, else one upload
        reconstruction = torch.empty((nb_images, csts.STRIDE_PROD*h_map, csts.STRIDE_PROD*w_map, 1), dtype=torch.uint8, device=latents.device)
        for chunk in chunks:
            decode_device(latents[chunk], reconstruction[chunk])
        (out, _) = bk.publish(reconstruction)
        isolated_decoder.check()
        return out
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
    encode_device = getattr(entropy_ae, 'encode_device_into', None)
    if (encode_device is not None and nb_images > 0 and
            (batch_size,) + tuple(luminances_uint8.shape[1:]) == tuple(entropy_ae.node_visible_units.shape)):
        import torch
        chunks = list(_launches(nb_images, batch_size, h_in*w_in))            # ValueError first, like the reference
        images = bk.to_device(luminances_uint8, numpy.uint8)
        y = torch.empty((nb_images, h_in//csts.STRIDE_PROD, w_in//csts.STRIDE_PROD, csts.NB_MAPS_3), dtype=torch.float32, device=images.device)
        for chunk in chunks:
            encode_device(images[chunk], y[chunk])
        (latents, _) = bk.publish(y)
        entropy_ae.check()
        return latents
    latents = numpy.zeros((nb_images, h_in//csts.STRIDE_PROD, w_in//csts.STRIDE_PROD, csts.NB_MAPS_3), dtype=numpy.float32)
    for chunk in _mini_batches(nb_images, batch_size):
        latents[chunk] = sess.run(entropy_ae.node_y,
                                  feed_dict={entropy_ae.node_visible_units: luminances_uint8[chunk].astype(numpy.float32)})
    return latents
