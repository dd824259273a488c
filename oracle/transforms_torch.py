"""The analysis / synthesis transforms on torch-CPU (oneDNN / MKL, float32, all usable host threads) -- TEST INFRASTRUCTURE
AND `bench.py: cpu_baseline` ONLY; the product path never imports this.

Why it exists: the reference runs its transforms inside TensorFlow-CPU (`sess.run`, kodak_tensorflow/eae/batching.py:94-99,
49-53), whose Eigen / oneDNN kernels are blocked, vectorised GEMM-style convolutions. The plain-C oracle
(oracle/transforms_oracle.c) is a per-element FMA chain written for bit-exactness, not speed (~0.1 TFLOP/s on 16 cores), so
timing IT as "the CPU path" flatters the GPU by an order of magnitude. TensorFlow is not installable here; torch-CPU's oneDNN
convolutions are the closest stand-in for TF's kernels ("CPU baseline timing plan").

Same op graph as oracle/transforms.py (the tables ENCODER_LAYERS / DECODER_LAYERS, pinned to the reference's `.ckpt.meta`
graphs by tests/test_oracle_graph.py): tf.nn.conv2d 'SAME' = explicit asymmetric zero padding + cross-correlation;
tf.nn.conv2d_transpose 'SAME' = conv_transpose2d cropped at pad_before; GDN = x / sqrt(x^2 Gamma + beta)
as a 1x1 convolution (tfutils.py:393-397). The summation order is the library's, so values agree with the C oracle only to
float32 rounding (tests/test_oracle_transforms.py: 1e-4 relative) -- good enough for a timing stand-in, never used as a checker.
"""
import numpy
import torch
import torch.nn.functional as F

from . import transforms as table


def _same_padding(size, k, s):
    out = -(-size//s)
    total = max((out - 1)*s + k - size, 0)
    return (total//2, total - total//2)


class CpuTransforms(object):
    """Weights converted once (NCHW / OIHW, channels_last memory format); `encoder` / `decoder` take and return NHWC numpy."""

    def __init__(self, variables, are_bin_widths_learned, threads=None, dtype=numpy.float32):
        # dtype numpy.float64: the same graph evaluated in double precision (oracle/order_sensitivity.py: the order-free reference)
        self.dtype = numpy.dtype(dtype)
        if threads:
            torch.set_num_threads(int(threads))
        self.learned = are_bin_widths_learned
        self.threads = torch.get_num_threads()
        self.v = {}
        for rows in (table.ENCODER_LAYERS, table.DECODER_LAYERS):
            for row in table.layers_of(rows, are_bin_widths_learned):
                if row[1] not in variables:          # an encoder-only or decoder-only set of variables
                    continue
                if row[0] == 'conv2d':
                    self.v[row[1]] = torch.from_numpy(numpy.ascontiguousarray(variables[row[1]].transpose(3, 2, 0, 1), dtype=self.dtype))       # HWIO -> OIHW
                elif row[0] == 'conv2d_transpose':
                    # TF filter [k, k, out, in]; conv_transpose2d wants [in, out, k, k]
                    self.v[row[1]] = torch.from_numpy(numpy.ascontiguousarray(variables[row[1]].transpose(3, 2, 0, 1), dtype=self.dtype))
                else:
                    # d[c] = beta[c] + sum_k x[k]^2 Gamma[k, c]: a 1x1 convolution with weight [c, k, 1, 1] = Gamma^T
                    self.v[row[1]] = torch.from_numpy(numpy.ascontiguousarray(variables[row[1]].T.reshape(128, 128, 1, 1), dtype=self.dtype))
                    self.v[row[2]] = torch.from_numpy(numpy.ascontiguousarray(variables[row[2]], dtype=self.dtype))
                if row[0] != 'gdn' and row[0] != 'inverse_gdn' and row[3]:
                    self.v[row[3]] = torch.from_numpy(numpy.ascontiguousarray(variables[row[3]], dtype=self.dtype))

    def _run(self, x_nhwc, rows):
        x = torch.from_numpy(numpy.ascontiguousarray(x_nhwc, dtype=self.dtype)).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            for row in table.layers_of(rows, self.learned):
                if row[0] == 'conv2d':
                    w = self.v[row[1]]
                    (k, s) = (w.shape[2], row[2])
                    (pt, pb) = _same_padding(x.shape[2], k, s)
                    (pl, pr) = _same_padding(x.shape[3], k, s)
                    x = F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, self.v[row[3]] if row[3] else None, stride=s)
                elif row[0] == 'conv2d_transpose':
                    w = self.v[row[1]]
                    (k, s) = (w.shape[2], row[2])
                    (h, wd) = (x.shape[2], x.shape[3])
                    pb = _same_padding(h*s, k, s)[0]
                    pl = _same_padding(wd*s, k, s)[0]
                    full = F.conv_transpose2d(x, w, None, stride=s)
                    x = full[:, :, pb:pb + s*h, pl:pl + s*wd]
                    if row[3]:
                        x = x + self.v[row[3]].view(1, -1, 1, 1)
                else:
                    d = torch.sqrt(F.conv2d(x*x, self.v[row[1]], self.v[row[2]]))
                    x = x*d if row[0] == 'inverse_gdn' else x/d
        return x.permute(0, 2, 3, 1).contiguous().numpy()

    def encoder(self, visible_units_float32):
        return self._run(visible_units_float32, table.ENCODER_LAYERS)

    def decoder(self, y_tilde):
        return self._run(y_tilde, table.DECODER_LAYERS)
