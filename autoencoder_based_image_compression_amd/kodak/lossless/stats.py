"""Statistics on the latent variable feature maps that parameterise the coder; mirrors
kodak_tensorflow/lossless/stats.py: `compute_binary_probabilities` :13-68, `compute_probabilities_intervals` :70-134,
`count_binary_decisions` :136-195, `find_index_map_exception` :197-241 and `save_statistics` :243-320.

The passes over the latents (symbol conversion, per-map histograms, minima / maxima, means) run on the MI355X; the
sums over a few dozen histogram bins per map, the float64 probabilities and divergences are formed on the host with the
reference's expressions, from exact integer counts.
"""
import os
import pickle

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

    Bit for bit numpy's result: one float32 accumulator per map, rows added in order, divided by float32(rows)
    (eae_hip_map_means; tests/test_gpu_full_size.py::test_map_means compares with `numpy.mean` using `array_equal`).
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


def _unit_interval_counts(y_float32):
    """For every map of y (N, h, w, C): (floor(min), ceil(max), int64 counts over the unit intervals between them), the
    histogram `compute_probabilities_intervals(map, 1.)` takes with `numpy.histogram` (last interval closed)."""
    from ... import device as dev
    y = bk.to_device(y_float32, numpy.float32)
    minmax = bk.to_host(dev.map_minmax(y)).astype(numpy.float64)
    edges_left = numpy.floor(minmax[0])
    edges_right = numpy.ceil(minmax[1])
    radius = int(max(numpy.abs(edges_left).max(), numpy.abs(edges_right).max())) + 1
    (hist, overflow) = dev.floor_histograms(y, radius)
    if int(overflow.sum().item()) != 0:
        raise ValueError('The latent variables contain non-finite values.')
    hist = bk.to_host(hist).astype(numpy.int64)
    out = []
    for i in range(hist.shape[0]):
        (lo, hi) = (int(edges_left[i]), int(edges_right[i]))
        counts = hist[i, lo + radius:hi + radius].copy()
        if counts.size:
            counts[-1] += hist[i, hi + radius]      # numpy.histogram closes the last interval: values equal to ceil(max)
        out.append((lo, hi, counts))
    return out


def compute_probabilities_intervals(data, size_interval):
    """Probability that a data value belongs to each axis interval of size `size_interval` between floor(min) and
    ceil(max) (stats.py:70-134). Host numpy, like the reference (the device path for unit intervals over whole maps is
    `find_index_map_exception`).

    Raises
    ------
    ValueError
        If the interval size exceeds the range of the data values, or the range cannot be split into an integer number
        of intervals.
    """
    edge_left = numpy.floor(numpy.amin(data)).item()
    edge_right = numpy.ceil(numpy.amax(data)).item()
    difference_edges = edge_right - edge_left
    if difference_edges < size_interval:
        raise ValueError('The interval size exceeds the range of the data values.')
    nb_edges_minus_1_float = difference_edges/size_interval
    if not float(nb_edges_minus_1_float).is_integer():
        raise ValueError('The range of the data values cannot be split into '
                         + 'an integer number of intervals of size {}.'.format(size_interval))
    bin_edges = numpy.linspace(edge_left, edge_right, num=int(nb_edges_minus_1_float) + 1)
    hist = numpy.histogram(data, bins=bin_edges, density=True)[0]
    return (bin_edges, hist*size_interval)


def map_divergences(y_float32):
    """Jensen-Shannon divergence between each map's unit-interval distribution and the uniform one (the loop body of
    stats.py:226-240), float64 (C,)."""
    divergences = numpy.zeros(y_float32.shape[3])
    for (i, (lo, hi, counts)) in enumerate(_unit_interval_counts(y_float32)):
        if hi - lo < 1.:
            raise ValueError('The interval size exceeds the range of the data values.')
        # numpy.histogram(density=True): n / diff(bin_edges) / n.sum(), then * size_interval (= 1.)
        probs = counts/numpy.ones(counts.size)/counts.sum()*1.
        probs_non_zero = numpy.extract(probs != 0., probs)
        nb_remaining_probs = probs_non_zero.size
        if nb_remaining_probs > 1:
            uniform_probs = (1./nb_remaining_probs)*numpy.ones(nb_remaining_probs)
            divergences[i] = tls.jensen_shannon_divergence(probs_non_zero, uniform_probs)
        else:
            divergences[i] = 1.
    return divergences


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


def find_index_map_exception(y_float32):
    """Index of the latent variable feature map that is not compressed as the other maps: the one whose distribution is
    closest (Jensen-Shannon) to the uniform distribution (stats.py:197-241)."""
    return numpy.argmin(map_divergences(y_float32)).item()


def save_statistics(luminances_uint8, sess, entropy_ae, batch_size, multipliers, truncated_unary_length,
                    path_to_map_mean, path_to_idx_map_exception, paths_to_binary_probabilities):
    """Saves the statistics on the latent variable feature maps that the coder needs (stats.py:243-320): the map means
    (.npy), the index of the exception map (.pkl, pickle protocol 2) and one table of binary probabilities per
    multiplier (.npy). Same files, same skip rule when they all exist.

    Raises
    ------
    ValueError
        If `len(paths_to_binary_probabilities)` is not equal to `multipliers.size`.
    """
    from ..eae import batching
    nb_multipliers = multipliers.size
    if len(paths_to_binary_probabilities) != nb_multipliers:
        raise ValueError('`len(paths_to_binary_probabilities)` is not equal to `multipliers.size`.')
    booleans = [os.path.isfile(path_to_binary_probability) for path_to_binary_probability in paths_to_binary_probabilities]
    if os.path.isfile(path_to_map_mean) and os.path.isfile(path_to_idx_map_exception) and all(booleans):
        print('The statistics on the latent variable feature maps already exist.')
        print('Delete them manually to recompute them.')
        return
    y_float32 = batching.encode_mini_batches(luminances_uint8, sess, entropy_ae, batch_size)
    map_mean = compute_map_mean(y_float32)
    numpy.save(path_to_map_mean, map_mean)
    idx_map_exception = find_index_map_exception(y_float32)
    with open(path_to_idx_map_exception, 'wb') as file:
        pickle.dump(idx_map_exception, file, protocol=2)
    for i in range(nb_multipliers):
        bin_widths_test = multipliers[i]*entropy_ae.get_bin_widths()
        binary_probabilities = compute_binary_probabilities(y_float32, bin_widths_test, map_mean, truncated_unary_length)
        numpy.save(paths_to_binary_probabilities[i], binary_probabilities)
