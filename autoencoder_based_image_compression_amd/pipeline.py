"""Device-resident analysis / synthesis transforms: the arithmetic behind `sess.run(entropy_ae.node_y)` and
`sess.run(isolated_decoder.node_reconstruction)` of the reference (kodak_tensorflow/eae/batching.py:96-99, 49-52).

`DeviceEncoder` / `DeviceDecoder` keep the variables of one trained model in HBM (7 MB) and chain the kernels of
include/eae_hip.h on torch's current stream; activations never leave the device. Both are the hot path of
`bench.py` and the engine under the reference-shaped classes in `kodak/eae/graph/`.
"""
import numpy
import torch

from . import device as dev
from .kodak.eae.graph import constants as csts
from .kodak.eae.graph import variables as var


def _to_device(array, device):
    return torch.from_numpy(numpy.ascontiguousarray(array, dtype=numpy.float32)).to(device)


class DeviceEncoder(object):
    """components.encoder (kodak_tensorflow/eae/graph/components.py:86-142) on the GPU."""

    def __init__(self, variables, are_bin_widths_learned, device='cuda'):
        names = var.ENCODER_NAMES + (() if are_bin_widths_learned else var.ENCODER_NAMES_FIXED_BW)
        var.check_variables(variables, names)
        self.are_bin_widths_learned = are_bin_widths_learned
        self.device = torch.device(device)
        self.v = {name: _to_device(variables[name], self.device) for name in names}
        # kernel-side layouts, packed once on the device (include/eae_hip.h "packed" channel order)
        self.w1 = dev.pack_conv9x9s4_weights(self.v['encoder/weights_1'])
        self.w2 = dev.pack_conv_weights(self.v['encoder/weights_2'])
        self.w3 = dev.pack_conv_weights(self.v['encoder/weights_3'])
        self.g = {i: dev.pack_gamma(self.v['encoder/gamma_{}'.format(i)]) for i in ((1, 2) if are_bin_widths_learned else (1, 2, 3))}
        # the same variables behind the library's whole-path entry point (include/eae_hip.h: eae_hip_encode), built on first use:
        # the per-layer layouts above are what codec.BatchCodec chains itself (it fuses gdn_3 into the latent stage and times
        # every launch) and a codec that is never called image by image should not pay for a second copy of the weights
        self._model = None
        self._model_variables = {name: variables[name] for name in names}

    @property
    def model(self):
        if self._model is None:
            with torch.cuda.device(self.device if self.device.index is not None else torch.cuda.current_device()):
                self._model = dev.Model(self._model_variables, self.are_bin_widths_learned)
            self._model_variables = None
        return self._model

    def check(self):
        """Waits for the calls issued so far and raises if one of them left tiles unfinished (device.Model.check)."""
        if self._model is not None:
            self._model.check(wait=True)

    def __call__(self, luminances_uint8, out=None):
        """uint8 [N,H,W] or [N,H,W,1] (device) -> float32 latents [N,H/16,W/16,128] (device); `out`: tensor to write them into."""
        if luminances_uint8.dtype != torch.uint8:
            raise TypeError('`luminances_uint8.dtype` is not equal to `torch.uint8`.')
        (h_in, w_in) = (luminances_uint8.shape[1], luminances_uint8.shape[2])
        if h_in % csts.STRIDE_PROD != 0:
            raise ValueError('The height of the input images is not divisible by the product of the three strides.')
        if w_in % csts.STRIDE_PROD != 0:
            raise ValueError('The width of the input images is not divisible by the product of the three strides.')
        if luminances_uint8.dim() == 4:
            luminances_uint8 = luminances_uint8[:, :, :, 0]
        return self.model.encode(luminances_uint8.contiguous(), out=out)


class DeviceDecoder(object):
    """components.decoder (components.py:11-84) + tls.cast_bt601 (batching.py:53) on the GPU."""

    def __init__(self, variables, are_bin_widths_learned, device='cuda'):
        names = var.DECODER_NAMES + (() if are_bin_widths_learned else var.DECODER_NAMES_FIXED_BW)
        var.check_variables(variables, names)
        self.are_bin_widths_learned = are_bin_widths_learned
        self.device = torch.device(device)
        self.v = {name: _to_device(variables[name], self.device) for name in names}
        # kernel-side weight layouts, packed once on the device
        self.w4 = dev.pack_tconv_weights(self.v['decoder/weights_4'])
        self.w5 = dev.pack_tconv_weights(self.v['decoder/weights_5'])
        self.w6 = dev.pack_tconv9x9s4_weights(self.v['decoder/weights_6'])
        self.g = {i: dev.pack_gamma(self.v['decoder/gamma_{}'.format(i)]) for i in ((5, 6) if are_bin_widths_learned else (4, 5, 6))}
        self._model = None           # eae_hip_decode, built on first use (see DeviceEncoder)
        self._model_variables = {name: variables[name] for name in names}

    @property
    def model(self):
        if self._model is None:
            with torch.cuda.device(self.device if self.device.index is not None else torch.cuda.current_device()):
                self._model = dev.Model(self._model_variables, self.are_bin_widths_learned)
            self._model_variables = None
        return self._model

    def check(self):
        if self._model is not None:
            self._model.check(wait=True)

    def __call__(self, quantized_y, want_float=False, want_uint8=True, reference_uint8=None, sse=None, out_uint8=None):
        """float32 [N,h,w,128] (device) -> (float32 [N,16h,16w] or None, uint8 [N,16h,16w] or None, sse or None)."""
        return self.model.decode(quantized_y.contiguous(), want_f32=want_float, want_u8=want_uint8, ref_u8=reference_uint8, sse=sse,
                                 out_u8=out_uint8)


# Algorithmic work per INPUT pixel of each launch, fixed-bin-width model.
# MACs: conv1 648 + GDN1 1024; conv2 6400 + GDN2 256; conv3 1600 + GDN3 64; IGDN4 64; tconv1 1600 + IGDN5 256;
# tconv2 6400 + IGDN6 1024; tconv3 648.
FLOP_PER_PIXEL = {
    'conv1_gdn1': 2*(648 + 1024),
    'conv2_gdn2': 2*(6400 + 256),
    'conv3_gdn3': 2*(1600 + 64),
    'igdn4': 2*64,
    'tconv1_igdn5': 2*(1600 + 256),
    'tconv2_igdn6': 2*(6400 + 1024),
    'tconv3': 2*648,
}
FLOP_PER_PIXEL_TOTAL = sum(FLOP_PER_PIXEL.values())   # 39,968
