"""The reference's `IsolatedDecoder` (kodak_tensorflow/eae/graph/IsolatedDecoder.py:21-129): decoder-only model fed
with quantized latent variables, on the MI355X."""

from . import constants as csts
from . import variables as var
from ... import _backend as bk
from ... import tf_shim
from .... import pipeline


class IsolatedDecoder(object):
    """Isolated decoder."""

    def __init__(self, batch_size, h_in, w_in, are_bin_widths_learned):
        """Same parameters as the reference (:21-47); the placeholder has shape
        (batch_size, h_in//16, w_in//16, 128) (:101-102)."""
        self.batch_size = batch_size
        self.h_in = h_in
        self.w_in = w_in
        self.are_bin_widths_learned = are_bin_widths_learned
        self._variables = None
        self._decoder = None
        self.node_quantized_y = tf_shim.Placeholder((batch_size, h_in//csts.STRIDE_PROD, w_in//csts.STRIDE_PROD, csts.NB_MAPS_3),
                                                    'quantized_y')
        self.node_reconstruction = tf_shim.Node(self._run_decoder, self.node_quantized_y, 'reconstruction')

    def _run_decoder(self, quantized_y_float32):
        if self._decoder is None:
            raise RuntimeError('Attempting to use uninitialized value decoder/weights_4: call `initialization` first.')
        (rec, _, _) = self._decoder(bk.to_device(quantized_y_float32, 'float32'), want_float=True, want_uint8=False)
        rec = bk.to_host(rec)[..., None]   # (batch, h_in, w_in, 1) float32, as the TF node returns
        self._decoder.check()
        return rec

    def decode_device(self, quantized_y_device, reference_uint8_device=None, sse=None):
        """Device-resident entry: returns (uint8 reconstruction tensor, per-image squared error or None)."""
        (_, rec_uint8, sse) = self._decoder(quantized_y_device, want_float=False, want_uint8=True,
                                            reference_uint8=reference_uint8_device, sse=sse)
        return (rec_uint8, sse)

    def decode_device_into(self, quantized_y_device, reconstruction_uint8_device):
        """One mini-batch on the device: float32 [batch,h,w,128] tensor -> `reconstruction_uint8_device` uint8 [batch,16h,16w,1],
        clipped to BT.601 and rounded (`sess.run(node_reconstruction)` + `tls.cast_bt601`, eae/batching.py:49-53, in one call)."""
        if self._decoder is None:
            raise RuntimeError('Attempting to use uninitialized value decoder/weights_4: call `initialization` first.')
        self._decoder(quantized_y_device, want_float=False, want_uint8=True, out_uint8=reconstruction_uint8_device)

    def check(self):
        """Waits for the launches issued so far and raises if one of them left tiles unfinished (device.Model.check)."""
        self._decoder.check()

    def initialization(self, sess, path_to_restore, seed=None):
        """Either initializes all variables or restores a previous model (:109-129)."""
        if path_to_restore:
            self._variables = var.restore_variables(path_to_restore, self.are_bin_widths_learned, 'decoder')
        else:
            self._variables = var.random_variables(1., self.are_bin_widths_learned, seed=seed)
        self._decoder = pipeline.DeviceDecoder(self._variables, self.are_bin_widths_learned, bk.device())
        self._decoder.model           # the variables go to the device here, like `Saver.restore` (:123-124)

    def set_variables(self, variables):
        self._variables = dict(variables)
        self._decoder = pipeline.DeviceDecoder(self._variables, self.are_bin_widths_learned, bk.device())
        self._decoder.model
