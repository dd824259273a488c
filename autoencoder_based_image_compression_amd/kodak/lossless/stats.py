"""Statistics on the latent variable feature maps that parameterise the coder; mirrors the hot-path part of
kodak_tensorflow/lossless/stats.py: `compute_binary_probabilities` :13-68 and `count_binary_decisions` :136-195.

The symbol conversion and the per-map histograms run on the MI355X; the prefix sums over at most L+1 histogram
bins per map and the float64 probabilities are formed on the host with the reference's expressions.
"""
import numpy

from .. import _backend as bk
from ..tools import tools as tls


def _decisions_from_hist(hist_abs, truncated_unary_length):
    """stats.py:181-195 from the histogram of absolute symbols (index = |symbol|)."""
    cumulated_zeros = numpy.zeros(truncated_unary_length, dtype=numpy.int64)
    cumulated_ones = numpy.zeros(truncated_unary_length, dtype=numpy.int64)
    for ii in numpy.flatnonzero(hist_abs):
        if ii < truncated_unary_length:
            cumulated_ones[0:ii] += hist_abs[ii]
            cumulated_zeros[ii] += hist_abs[ii]
        else:
            cumulated_ones += hist_abs[ii]
    return (cumulated_zeros, cumulated_ones)


def _abs_histograms(quantized, bin_widths):
    """[N, hw, C] quantised floats -> histograms of |symbol| accumulated over N: int64 [C, R+1]."""
    symbols = tls._quantized_to_symbols(quantized, bin_widths)
    (hist, radius) = tls._symbol_histograms(symbols)
    (n, c) = (quantized.shape[0], quantized.shape[2])
    hist = hist.reshape(n, c, -1).sum(axis=0)
    hist_abs = hist[:, radius:].copy()
    hist_abs[:, 1:] += hist[:, :radius][:, ::-1]
    return hist_abs


def compute_map_mean(y_float32):
    """Per-map mean of the latent variables, `numpy.mean(y_float32, axis=(0, 1, 2))` of stats.py:306, on the device.

    float64 accumulation then rounding to float32: within a few float32 ulps of numpy's float32 pairwise mean.
    """
    from ... import device as dev
    return bk.to_host(dev.map_means(bk.to_device(y_float32, numpy.float32)))


# The functions are sorted in alphabetic order.

def compute_binary_probabilities(y_float32, bin_widths_test, map_mean, truncated_unary_length):
    """Probability that each binary decision of the truncated unary prefix is 0, per map (stats.py:13-68).

    y_float32 (N, h, w, C) float32; bin_widths_test (C,) float32; map_mean (C,) float32 -> float64 (C, L).
    """
    (nb_images, height_map, width_map, nb_maps) = y_float32.shape
    centered_y_float32 = y_float32 - numpy.tile(map_mean, (nb_images, height_map, width_map, 1))
    centered_quantized_y_float32 = tls.quantize_per_map(centered_y_float32, bin_widths_test)
    hist_abs = _abs_histograms(centered_quantized_y_float32.reshape(nb_images, height_map*width_map, nb_maps),
                               numpy.asarray(bin_widths_test, dtype=numpy.float32))
    cumulated_zeros = numpy.zeros((nb_maps, truncated_unary_length), dtype=numpy.int64)
    cumulated_ones = numpy.zeros((nb_maps, truncated_unary_length), dtype=numpy.int64)
    for i in range(nb_maps):
        (cumulated_zeros[i, :], cumulated_ones[i, :]) = _decisions_from_hist(hist_abs[i], truncated_unary_length)
    total = cumulated_zeros + cumulated_ones
    with numpy.errstate(invalid='ignore'):
        binary_probabilities = cumulated_zeros.astype(numpy.float64)/total.astype(numpy.float64)
    binary_probabilities[numpy.isnan(binary_probabilities)] = 0.5
    binary_probabilities[binary_probabilities == 0.] = 0.01
    binary_probabilities[binary_probabilities == 1.] = 0.99
    return binary_probabilities


def count_binary_decisions(abs_centered_quantized_data, bin_width_test, truncated_unary_length):
    """Counts the zeros and ones of each binary decision of the truncated unary prefix (stats.py:136-195).

    Raises
    ------
    ValueError
        If an element of `abs_centered_quantized_data` is not positive.
    """
    if numpy.any(abs_centered_quantized_data < 0.):
        raise ValueError('An element of `abs_centered_quantized_data` is not positive.')
    if bin_width_test <= 0.:
        raise ValueError('The quantization bin width is not strictly positive.')
    flat = numpy.ascontiguousarray(abs_centered_quantized_data, dtype=numpy.float32).reshape(1, -1, 1)
    hist_abs = _abs_histograms(flat, numpy.array([bin_width_test], dtype=numpy.float32))
    return _decisions_from_hist(hist_abs[0], truncated_unary_length)
