"""Inference surface of the reference's `EntropyAutoencoder` (kodak_tensorflow/eae/graph/EntropyAutoencoder.py).

Kept: the constructor signature and its shape checks (:36-80), `node_visible_units` / `node_y` (:248-251),
`initialization` (:440-463), `get_bin_widths` (:398-409), `save` (:465-482, to `.npz`). The training graph
(:252-396, :411-438) -- noise, piecewise-linear density, optimisers, expansion -- is out of scope.
"""
import pickle

import numpy

from . import constants as csts
from . import variables as var
from ... import _backend as bk
from ... import tf_shim
from .... import pipeline


class EntropyAutoencoder(object):
    """Entropy autoencoder, encoder side, on the MI355X."""

    def __init__(self, batch_size, h_in, w_in, bin_width_init, gamma_scaling, path_to_nb_itvs_per_side_load, are_bin_widths_learned):
        """Same parameters as the reference (:36-74). `gamma_scaling` only matters for training and is stored;
        `path_to_nb_itvs_per_side_load` is read like the reference does (:81-85) and otherwise unused at inference.

        Raises
        ------
        ValueError
            If the height (width) of the input images is not divisible by the product of the three strides.
        """
        if h_in % csts.STRIDE_PROD != 0:
            raise ValueError('The height of the input images is not divisible by the product of the three strides.')
        if w_in % csts.STRIDE_PROD != 0:
            raise ValueError('The width of the input images is not divisible by the product of the three strides.')
        if path_to_nb_itvs_per_side_load:
            with open(path_to_nb_itvs_per_side_load, 'rb') as file:
                self.nb_itvs_per_side = pickle.load(file)
        else:
            self.nb_itvs_per_side = 10   # csts.NB_ITVS_PER_SIDE_INIT (constants.py:35)
        self.batch_size = batch_size
        self.h_in = h_in
        self.w_in = w_in
        self.bin_width_init = bin_width_init
        self.gamma_scaling = gamma_scaling
        self.are_bin_widths_learned = are_bin_widths_learned
        self._variables = None
        self._encoder = None
        self.node_visible_units = tf_shim.Placeholder((batch_size, h_in, w_in, 1), 'visible_units')
        self.node_y = tf_shim.Node(self._run_encoder, self.node_visible_units, 'y')

    def _run_encoder(self, batch_float32):
        if self._encoder is None:
            raise RuntimeError('Attempting to use uninitialized value encoder/weights_1: call `initialization` first.')
        # The reference feeds float32 pixels that were uint8 one line earlier (batching.py:95); the kernel takes the
        # bytes directly, so values must be exactly representable as uint8.
        as_uint8 = batch_float32.astype(numpy.uint8)
        if not numpy.array_equal(as_uint8.astype(batch_float32.dtype), batch_float32):
            raise ValueError('`node_visible_units` must be fed 8-bit luminance values cast to float (eae/batching.py:95).')
        y = bk.to_host(self._encoder(bk.to_device(as_uint8[..., 0])))
        self._encoder.check()            # the copy above waited for the launches: a failed hand-off raises here, not later
        return y

    def encode_uint8_device(self, luminances_uint8_device):
        """Device-resident entry (no host copies): uint8 [N,H,W] tensor -> float32 latents tensor."""
        return self._encoder(luminances_uint8_device)

    def encode_device_into(self, luminances_uint8_device, latents_device):
        """One mini-batch on the device: uint8 [batch,H,W,1] tensor -> `latents_device` float32 [batch,H/16,W/16,128]
        (`eae.batching.encode_mini_batches` calls this per mini-batch; what `sess.run(node_y)` computes, eae/batching.py:96-99)."""
        if self._encoder is None:
            raise RuntimeError('Attempting to use uninitialized value encoder/weights_1: call `initialization` first.')
        self._encoder(luminances_uint8_device, out=latents_device)

    def check(self):
        """Waits for the launches issued so far and raises if one of them left tiles unfinished (device.Model.check)."""
        self._encoder.check()

    def get_bin_widths(self):
        """Quantization bin widths, 1D `numpy.float32` (:398-409)."""
        if self._variables is None:
            raise RuntimeError('Attempting to use uninitialized value piecewise_linear_function/bin_widths.')
        return self._variables[var.BIN_WIDTHS_NAME].copy()

    def initialization(self, sess, path_to_restore, seed=None):
        """Either initializes all variables or restores a previous model (:440-463).

        `path_to_restore`: '' -> random initialisation like the reference's; otherwise the ".ckpt" prefix of a
        TensorFlow checkpoint (V1 or V2, read without TensorFlow) or of a sibling `.npz` keyed by the TF variable
        names (`variables.restore_variables`).
        """
        if path_to_restore:
            self._variables = var.restore_variables(path_to_restore, self.are_bin_widths_learned, 'encoder')
        else:
            self._variables = var.random_variables(self.bin_width_init, self.are_bin_widths_learned, seed=seed)
        self._encoder = pipeline.DeviceEncoder(self._variables, self.are_bin_widths_learned, bk.device())
        self._encoder.model           # `Saver.restore` puts the variables in place here (:454-458), not at the first `sess.run`

    def set_variables(self, variables):
        """Installs variables given as a dict keyed by the TF names (used by tests and by drivers holding weights in memory)."""
        self._variables = dict(variables)
        self._encoder = pipeline.DeviceEncoder(self._variables, self.are_bin_widths_learned, bk.device())
        self._encoder.model

    def save(self, sess, path_to_model, path_to_nb_itvs_per_side_save):
        """Saves the variables (`.npz`) and the number of unit intervals (:465-482)."""
        path = path_to_model[:-5] + '.npz' if path_to_model.endswith('.ckpt') else path_to_model
        var.save_variables(path, self._variables)
        with open(path_to_nb_itvs_per_side_save, 'wb') as file:
            pickle.dump(self.nb_itvs_per_side, file, protocol=2)
