"""Hot-path subset of the reference's `tools/tools.py`, same names / arguments / exceptions, computed on the MI355X.

Reference: kodak_tensorflow/tools/tools.py -- `average_entropies` :25-59, `cast_bt601` :61-93,
`cast_float_to_int16` :95-133, `count_nb_deads` :294-320, `count_symbols` :322-388, `discrete_entropy` :486-537,
`float_to_str` :570-593, `psnr_2d` :831-881, `quantize_per_map` :883-929, `rate_3d` :931-989, `subdivide_set`
:1108-1132. Arrays come in and go out as numpy (the reference's contract); the element-wise and counting work runs in
the kernels of include/eae_hip.h; the few float64 scalars at the end (`-sum f log2 f`, `10 log10(255^2/mse)`) are
formed on the host from EXACT integer counts with the reference's own numpy expressions, so they are bit-identical.
Plotting, image I/O, Bjontegaard and dataset helpers of the reference file are out of scope.
"""
import numpy

from ... import device as dev
from .. import _backend as bk

# Histogram radius tried first; symbols outside trigger a re-run with the int16-covering radius (exact either way).
_FIRST_RADIUS = 255
_FULL_RADIUS = 32768


def _is_floating(array):
    return numpy.issubdtype(array.dtype, numpy.floating)


# The functions are sorted in alphabetic order, like the reference.

def average_entropies(data, bin_widths):
    """Quantizes the data and computes the mean entropy of the quantized data (tools.py:25-59)."""
    quantized_data = quantize_per_map(data, bin_widths)
    nb_maps = data.shape[3]
    entropies = _map_entropies(numpy.moveaxis(quantized_data, 3, 0).reshape(1, nb_maps, -1), bin_widths, planar=True)
    cumulated_entropy = 0.
    for i in range(nb_maps):
        cumulated_entropy += entropies[0, i]
    return cumulated_entropy/nb_maps


def cast_bt601(array_float):
    """Clips to [16., 235.], rounds half to even and casts to `numpy.uint8` (tools.py:61-93)."""
    if not _is_floating(array_float):
        raise TypeError('`array_float.dtype` is not smaller than `numpy.float` in type hierarchy.')
    if array_float.size == 0:
        return numpy.zeros(array_float.shape, dtype=numpy.uint8)
    return bk.to_host(dev.cast_bt601(bk.to_device(array_float, numpy.float32))).reshape(array_float.shape)


def cast_float_to_int16(array_float):
    """Rounds half to even and casts to `numpy.int16`; AssertionError outside [-32767, 32767] (tools.py:95-133)."""
    if not _is_floating(array_float):
        raise TypeError('`array_float.dtype` is not smaller than `numpy.float` in type hierarchy.')
    if array_float.size == 0:
        return numpy.zeros(array_float.shape, dtype=numpy.int16)
    (out, range_error) = dev.cast_int16(bk.to_device(array_float, numpy.float32))
    if int(range_error.item()) != 0:
        raise AssertionError('The rounded array elements cannot be represented as 16-bit signed integers.')
    return bk.to_host(out).reshape(array_float.shape)


def compute_bjontegaard(rates_0, psnrs_0, rates_1, psnrs_1):
    """Bjontegaard's metric: average per cent saving in bitrate of curve 1 over curve 0 (tools.py:157-263). Host float64:
    cubic fit of log-rate against PSNR for each curve, integrated over the common PSNR range.

    Raises
    ------
    ValueError
        If a rate array is not 1D or a PSNR array does not have the shape of its rate array.
    AssertionError
        If a rate or a PSNR is not strictly positive.
    """
    curves = ((rates_0, psnrs_0, '0'), (rates_1, psnrs_1, '1'))
    for (rates, _, tag) in curves:
        if rates.ndim != 1:
            raise ValueError('`rates_{}.ndim` is not equal to 1.'.format(tag))
    for (rates, psnrs, tag) in curves:
        if psnrs.shape != rates.shape:
            raise ValueError('`psnrs_{0}.shape` is not equal to `rates_{0}.shape`.'.format(tag))
    for (name, values) in (('rates_0', rates_0), ('rates_1', rates_1), ('psnrs_0', psnrs_0), ('psnrs_1', psnrs_1)):
        numpy.testing.assert_array_less(0., values, err_msg='An element of `{}` is not strictly positive.'.format(name))
    minimum = max(numpy.amin(psnrs_0).item(), numpy.amin(psnrs_1).item())
    maximum = min(numpy.amax(psnrs_0).item(), numpy.amax(psnrs_1).item())

    def integral_of_log_rate(rates, psnrs):
        antiderivative = numpy.polyint(numpy.polyfit(psnrs, numpy.log(rates), 3))
        return numpy.polyval(antiderivative, maximum) - numpy.polyval(antiderivative, minimum)

    difference = integral_of_log_rate(rates_1, psnrs_1) - integral_of_log_rate(rates_0, psnrs_0)
    return 100.*(numpy.exp(difference/(maximum - minimum)).item() - 1.)


def count_nb_deads(array_4d):
    """Number of dead feature maps (sum of absolute values exactly 0) per first-axis component (tools.py:294-320)."""
    if array_4d.ndim != 4:
        raise ValueError('`array_4d.ndim` is not equal to 4.')
    found = bk.resident(array_4d)
    if found is not None and found[1] == 0 and 'nonzero_flags' in found[0].extras and array_4d.shape == tuple(found[0].tensor.shape):
        flags = bk.to_host(found[0].extras['nonzero_flags'])        # the array `quantize_per_map` returned: counted in that pass
    else:
        flags = bk.to_host(dev.nonzero_flags(bk.to_device(array_4d, numpy.float32)))
    return numpy.sum(flags == 0, axis=1)


def _symbol_histograms(symbols_planar_device):
    """Exact per-map histograms of int16 symbols: (hist int64 [n_maps, 2R+1], R)."""
    radius = _FIRST_RADIUS
    (hist, overflow) = dev.symbol_histograms(symbols_planar_device, radius)
    if int(overflow.sum().item()) != 0:
        radius = _FULL_RADIUS
        (hist, overflow) = dev.symbol_histograms(symbols_planar_device, radius)
    return (bk.to_host(hist).astype(numpy.int64), radius)


def _quantized_to_symbols(quantized_planar_or_nhwc, bin_widths_float32, check_quantized=True):
    """float quantised samples [N, hw, C] -> int16 symbols [N, C, hw] on the device (+ the tools.py:372-375 check)."""
    res = dev.quantize_maps(bk.to_device(quantized_planar_or_nhwc, numpy.float32), bk.to_device(bin_widths_float32, numpy.float32),
                            None, want_symbols=True)
    checks = res['checks'].cpu().tolist()
    if check_quantized and checks[1] != 0:
        raise AssertionError('\nArrays are not almost equal to 10 decimals\nThe quantization was omitted.')
    if checks[0] != 0:
        raise ValueError('A symbol does not fit in 16 bits: outside the domain of this build (lossless/compression.py:142 '
                         'casts the symbols to int16).')
    return res['symbols']


def count_symbols(quantized_samples, bin_width):
    """Number of occurrences of each symbol from the smallest to the largest quantized sample (tools.py:322-388)."""
    if bin_width <= 0.:
        raise ValueError('The quantization bin width is not strictly positive.')
    if numpy.size(quantized_samples) == 0:   # numpy.amin of the reference (tools.py:376)
        raise ValueError('zero-size array to reduction operation minimum which has no identity')
    flat = numpy.ascontiguousarray(quantized_samples, dtype=numpy.float32).reshape(1, -1, 1)
    symbols = _quantized_to_symbols(flat, numpy.array([bin_width], dtype=numpy.float32))
    (hist, radius) = _symbol_histograms(symbols)
    occupied = numpy.flatnonzero(hist[0])
    return hist[0, occupied[0]:occupied[-1] + 1]


def _entropy_from_hist(hist):
    """tools.py:523-537 verbatim, from the integer histogram."""
    hist_non_zero = numpy.extract(hist != 0, hist)
    frequency = hist_non_zero.astype(numpy.float64)/numpy.sum(hist_non_zero)
    disc_entropy = -numpy.sum(frequency*numpy.log2(frequency))
    if disc_entropy < 0.:
        raise ValueError('The entropy is not positive.')
    if disc_entropy > numpy.log2(hist_non_zero.size):
        raise ValueError('The entropy is not smaller than its upper bound.')
    return disc_entropy


def crop_repeat_2d(image_uint8, row_top_left, column_top_left):
    """80x80 crop with every pixel repeated 2x2 -> uint8 160x160 (tools.py:434-484). Host-side harness helper.

    Raises
    ------
    TypeError
        If `image_uint8.dtype` is not equal to `numpy.uint8`.
    ValueError
        If the image is not strictly larger than the crop's bottom / right edge (or is not 2D).
    """
    if image_uint8.dtype != numpy.uint8:
        raise TypeError('`image_uint8.dtype` is not equal to `numpy.uint8`.')
    (height_image, width_image) = image_uint8.shape
    for (start, extent, axis) in ((row_top_left, height_image, 0), (column_top_left, width_image, 1)):
        if start + 80 >= extent:
            raise ValueError('`image_uint8.shape[{0}]` is not strictly larger than `{1}_top_left + 80`.'.format(
                axis, 'row' if axis == 0 else 'column'))
    return numpy.kron(image_uint8[row_top_left:row_top_left + 80, column_top_left:column_top_left + 80],
                      numpy.ones((2, 2), dtype=numpy.uint8))


def discrete_entropy(quantized_samples, bin_width):
    """Entropy of the quantized samples (tools.py:486-537)."""
    return _entropy_from_hist(count_symbols(quantized_samples, bin_width))


def _map_entropies(quantized, bin_widths, planar=False):
    """Entropy of every map: `quantized` is [N, hw, C] (or [N, C, hw] if planar) -> float64 [N, C]."""
    if planar:
        quantized = numpy.ascontiguousarray(numpy.swapaxes(quantized, 1, 2))
    if numpy.any(bin_widths <= 0.):
        raise ValueError('The quantization bin width is not strictly positive.')
    symbols = _quantized_to_symbols(quantized, bin_widths)
    (hist, radius) = _symbol_histograms(symbols)
    (n, c) = (quantized.shape[0], quantized.shape[2])
    entropies = numpy.zeros((n, c))
    for i in range(n*c):
        row = hist[i]
        occupied = numpy.flatnonzero(row)
        entropies[i//c, i % c] = _entropy_from_hist(row[occupied[0]:occupied[-1] + 1])
    return entropies


def _image_of_published_batch(array_3d):
    """If `array_3d` (h, w, C) is image j of a 4-D array a function of this package returned (`cq[j, :, :, :]` in the reference's
    harness, reconstructing_eae_kodak.py:214-223): (its `_backend.Resident`, j), else None."""
    found = bk.resident(array_3d)
    if found is None:
        return None
    (record, first) = found
    tensor = record.tensor
    if tensor.dim() != 4 or tuple(tensor.shape[1:]) != array_3d.shape or first % array_3d.size != 0 or tensor.device != bk.device():
        return None
    return (record, first//array_3d.size)


_ROW_SUMS_IN_NUMPY_ORDER = []      # [True / False] once checked


def _row_sums(values, bounds):
    """`numpy.sum(values[bounds[i]:bounds[i + 1]])` for every i (float64, contiguous): one call of the host library, which adds in
    numpy's pairwise order (include/eae_coder.h: eae_coder_pairwise_row_sums), instead of one `numpy.sum` per feature map. The
    first call holds the library against `numpy.sum` itself on runs of every length class (short, one block, halved once and
    twice, with remainders); should the installed numpy add in another order, `numpy.sum` per row it stays."""
    import ctypes
    from autoencoder_based_image_compression_amd import _native

    def library(v, b):
        out = numpy.empty(b.size - 1)
        rc = _native.coder().eae_coder_pairwise_row_sums(v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), b.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                                         b.size - 1, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        if rc != 0:
            raise RuntimeError('eae_coder_pairwise_row_sums failed ({0}).'.format(rc))
        return out

    def one_by_one(v, b):
        edges = b.tolist()
        return numpy.array([numpy.sum(v[edges[i]:edges[i + 1]]) for i in range(len(edges) - 1)], dtype=numpy.float64)

    if not _ROW_SUMS_IN_NUMPY_ORDER:
        lengths = [0, 1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 100, 127, 128, 129, 136, 255, 256, 257, 300, 511, 513, 1000, 2047]
        probe_bounds = numpy.concatenate(([0], numpy.cumsum(lengths))).astype(numpy.int64)
        rng = numpy.random.RandomState(12345)
        probe = -rng.rand(int(probe_bounds[-1]))*numpy.exp2(rng.randint(-30, 4, size=int(probe_bounds[-1])))
        _ROW_SUMS_IN_NUMPY_ORDER.append(library(probe, probe_bounds).tobytes() == one_by_one(probe, probe_bounds).tobytes())
    if _ROW_SUMS_IN_NUMPY_ORDER[0]:
        return library(values, bounds)
    return one_by_one(values, bounds)


def _entropies_from_hist_rows(hist_rows):
    """`_entropy_from_hist` (tools.py:523-537) of every row of `hist_rows` (int64 [C, bins]) -> float64 [C], bit for bit: the
    element-wise steps (counts / total, f*log2(f)) run once over the non-empty bins of all rows (element-wise results do not
    depend on where an element sits in its array), the sum of each row's terms is `numpy.sum` over that row's own contiguous run
    (numpy's pairwise order depends on the length only; `_row_sums`: the same order, all rows in one call), and a row whose
    entropy comes within 1e-9 of a bound goes through the verbatim function so that its two comparisons see the reference's
    own scalars."""
    nonzero = hist_rows != 0
    lengths = nonzero.sum(axis=1)
    counts = hist_rows[nonzero]                                     # row-major: each row's non-empty bins, ascending
    frequency = counts.astype(numpy.float64)/numpy.repeat(hist_rows.sum(axis=1), lengths)
    terms = frequency*numpy.log2(frequency)
    bounds = numpy.concatenate(([0], numpy.cumsum(lengths))).astype(numpy.int64)
    entropies = -_row_sums(numpy.ascontiguousarray(terms), bounds)
    suspicious = numpy.flatnonzero((entropies < 1.e-9) | (entropies > numpy.log2(lengths) - 1.e-9))
    for i in suspicious.tolist():
        entropies[i] = _entropy_from_hist(hist_rows[i, nonzero[i]])
    return entropies


def _resident_symbol_histograms(record, bin_widths):
    """Symbols and per-map histograms of EVERY image of a published batch of quantised latents, computed on the device copy at
    the first call that asks for one of its images and kept with it: host int64 [N, C, 2R+1] (R = `_FIRST_RADIUS`), or None
    when a check of `_quantized_to_symbols` / `_symbol_histograms` would not pass for the batch as a whole -- the caller then
    takes the image-by-image path, which raises (or widens the histogram) for exactly the image concerned."""
    key = ('symbol_histograms', numpy.asarray(bin_widths, dtype=numpy.float32).tobytes())
    if key not in record.extras:
        tensor = record.tensor
        (n, c) = (tensor.shape[0], tensor.shape[3])
        res = dev.quantize_maps(tensor.view(n, -1, c), bk.to_device(numpy.asarray(bin_widths, dtype=numpy.float32)), None, want_symbols=True)
        (hist, overflow) = dev.symbol_histograms(res['symbols'], _FIRST_RADIUS)
        # only the columns some map of the batch uses travel (a map's entropy is a function of its non-empty bins in ascending
        # order: columns that are empty in every map change nothing, and at 0.2 bpp they are 4,070 of the 4,095)
        used = (hist != 0).any(dim=0).nonzero().reshape(-1)
        span = used[[0, -1]] if used.numel() else used.new_zeros(2)
        state = torch_cat_to_host(res['checks'], overflow, span.to(hist.dtype))
        (checks, first, last) = (state[:-2], int(state[-2]), int(state[-1]))
        if checks[:3].any() or checks[3:].any():
            record.extras[key] = None
        else:
            record.extras[key] = bk.to_host(hist[:, first:last + 1].contiguous()).astype(numpy.int64).reshape(n, c, -1)
    return record.extras[key]


def torch_cat_to_host(*tensors):
    """Several small int32 device tensors -> one host array, one device -> host copy."""
    import torch
    return torch.cat([t.reshape(-1) for t in tensors]).cpu().numpy()


def float_to_str(float_in):
    """Converts the float into a string, "." -> "dot", "-" -> "minus" (tools.py:570-593)."""
    if float_in.is_integer():
        str_in = str(int(float_in))
    else:
        str_in = str(float_in).replace('.', 'dot')
    return str_in.replace('-', 'minus')


def jensen_shannon_divergence(probs_0, probs_1):
    """Jensen-Shannon divergence between two probability distributions (tools.py:615-666), float64 on the host.

    Raises
    ------
    ValueError
        If a probability does not belong to ]0.0, 1.0[, a distribution does not sum to 1.0, or the divergence leaves [0, 1].
    """
    if numpy.any(probs_0 <= 0.) or numpy.any(probs_0 >= 1.):
        raise ValueError('A probability in `probs_0` does not belong to ]0.0, 1.0[.')
    if numpy.any(probs_1 <= 0.) or numpy.any(probs_1 >= 1.):
        raise ValueError('A probability in `probs_1` does not belong to ]0.0, 1.0[.')
    if abs(numpy.sum(probs_0).item() - 1.) >= 1.e-9:
        raise ValueError('The probabilities in `probs_0` do not sum to 1.0.')
    if abs(numpy.sum(probs_1).item() - 1.) >= 1.e-9:
        raise ValueError('The probabilities in `probs_1` do not sum to 1.0.')
    denominator = 0.5*(probs_0 + probs_1)
    divergence = 0.5*numpy.sum(probs_0*numpy.log2(probs_0/denominator) + probs_1*numpy.log2(probs_1/denominator))
    if divergence < 0.:
        raise ValueError('The Jensen-Shannon divergence is not positive.')
    if divergence > 1.:
        raise ValueError('The Jensen-Shannon divergence is not smaller than 1.0.')
    return divergence


def psnr_2d(reference_uint8, reconstruction_uint8):
    """PSNR between the luminance image and its reconstruction (tools.py:831-881)."""
    if reference_uint8.dtype != numpy.uint8:
        raise TypeError('`reference_uint8.dtype` is not equal to `numpy.uint8`.')
    if reconstruction_uint8.dtype != numpy.uint8:
        raise TypeError('`reconstruction_uint8.dtype` is not equal to `numpy.uint8`.')
    if reference_uint8.ndim != 2:
        raise ValueError('`reference_uint8.ndim` is not equal to 2.')
    if reference_uint8.shape != reconstruction_uint8.shape:
        raise ValueError('`reference_uint8.shape` is not equal to `reconstruction_uint8.shape`.')
    sse = dev.sse_u8(bk.to_device(reference_uint8[None]), bk.to_device(reconstruction_uint8[None]))
    return psnr_from_sse(int(sse.item()), reference_uint8.size)


def psnr_from_sse(sse, nb_pixels):
    """tools.py:875-881 from the exact integer sum of squared errors (a sum of integers < 2^53 is exact in float64,
    so `numpy.mean((a - b)**2)` == sse/nb_pixels bit for bit)."""
    mse = numpy.float64(sse)/nb_pixels
    if mse == 0.:
        raise ValueError('The mean squared error between the luminance image and its reconstruction is 0.')
    return 10.*numpy.log10((255.**2)/mse)


def quantize_per_map(data, bin_widths):
    """Uniform scalar quantization of each map with its own bin width (tools.py:883-929)."""
    if bin_widths.ndim != 1:
        raise ValueError('`bin_widths.ndim` is not equal to 1.')
    (nb_examples, height_map, width_map, nb_maps) = data.shape
    if bin_widths.size != nb_maps:
        raise ValueError('`bin_widths.size` is not equal to `data.shape[3]`.')
    if numpy.any(bin_widths <= 0.):
        raise ValueError('A quantization bin width is not strictly positive.')
    if data.size == 0:
        return numpy.zeros(data.shape, dtype=numpy.float32)
    # one pass: the quantised values and, for `count_nb_deads` on the returned array, which maps have a non-zero value
    res = dev.quantize_maps(bk.to_device(data, numpy.float32), bk.to_device(bin_widths, numpy.float32), None, want_cq=True, want_flags=True)
    (quantized, record) = bk.publish(res['cq'])
    record.extras['nonzero_flags'] = res['nonzero_flags']
    return quantized


def rate_3d(quantized_latent_float32, bin_widths, h_in, w_in):
    """Rate (bits per pixel) of the quantized latent variables of one luminance image (tools.py:931-989)."""
    if bin_widths.ndim != 1:
        raise ValueError('`bin_widths.ndim` is not equal to 1.')
    (height_map, width_map, nb_maps) = quantized_latent_float32.shape
    if bin_widths.size != nb_maps:
        raise ValueError('`bin_widths.size` is not equal to `quantized_latent_float32.shape[2]`.')
    batch = _image_of_published_batch(quantized_latent_float32) if quantized_latent_float32.dtype == numpy.float32 else None
    if batch is not None and not numpy.any(numpy.asarray(bin_widths) <= 0.):
        # image j of a batch `quantize_per_map` returned: the histograms of the whole batch are taken once, on its device copy
        histograms = _resident_symbol_histograms(batch[0], bin_widths)
        if histograms is not None:
            return rate_from_entropies(_entropies_from_hist_rows(histograms[batch[1]]), height_map, width_map, h_in, w_in)
    entropies = _map_entropies(numpy.ascontiguousarray(quantized_latent_float32, dtype=numpy.float32).reshape(1, -1, nb_maps),
                               numpy.asarray(bin_widths, dtype=numpy.float32))
    return rate_from_entropies(entropies[0], height_map, width_map, h_in, w_in)


def rate_from_entropies(entropies, height_map, width_map, h_in, w_in):
    """tools.py:977-989 given the per-map entropies (same accumulation order)."""
    cumulated_rate = 0.
    for i in range(entropies.size):
        cumulated_rate += entropies[i]*height_map*width_map
    return cumulated_rate/(h_in*w_in)


def read_image_mode(path, mode):
    """Reads the image if its mode matches the given mode, e.g. 'RGB' or 'L' (tools.py:991-1017).

    Raises
    ------
    ValueError
        If the image mode is not equal to `mode`.
    """
    import PIL.Image
    image = PIL.Image.open(path)
    if image.mode != mode:
        raise ValueError('The image mode is {0} whereas the given mode is {1}.'.format(image.mode, mode))
    return numpy.asarray(image)


def rgb_to_ycbcr(rgb_uint8):
    """Converts the RGB image to YCbCr, ITU-R BT.601 like Matlab's `rgb2ycbcr` (tools.py:1019-1083), on the device.

    Raises
    ------
    TypeError
        If `rgb_uint8.dtype` is not equal to `numpy.uint8`.
    ValueError
        If `rgb_uint8.ndim` is not equal to 3 or `rgb_uint8.shape[2]` is not equal to 3.
    """
    if rgb_uint8.dtype != numpy.uint8:
        raise TypeError('`rgb_uint8.dtype` is not equal to `numpy.uint8`.')
    if rgb_uint8.ndim != 3:
        raise ValueError('`rgb_uint8.ndim` is not equal to 3.')
    if rgb_uint8.shape[2] != 3:
        raise ValueError('`rgb_uint8.shape[2]` is not equal to 3.')
    return bk.to_host(dev.rgb_to_ycbcr(bk.to_device(rgb_uint8))[0])


def save_image(path, array_uint8):
    """Saves the array as an image (tools.py:1082-1106).

    Raises
    ------
    TypeError
        If `array_uint8.dtype` is not equal to `numpy.uint8`.
    """
    if array_uint8.dtype != numpy.uint8:
        raise TypeError('`array_uint8.dtype` is not equal to `numpy.uint8`.')
    import PIL.Image
    PIL.Image.fromarray(array_uint8).save(path)


def subdivide_set(nb_examples, batch_size):
    """Number of mini-batches in the set of examples (tools.py:1108-1132)."""
    if nb_examples % batch_size != 0:
        raise ValueError('`nb_examples` is not divisible by `batch_size`.')
    return nb_examples//batch_size


def visualize_crops(image_uint8, positions_top_left, paths):
    """Saves one 160x160 `crop_repeat_2d` per column of `positions_top_left` (int32 (2, nb_crops)) (tools.py:1172-1218).

    Raises
    ------
    ValueError
        If `positions_top_left.shape[0]` is not equal to 2 or `len(paths)` is not the number of crops.
    """
    (nb_rows, nb_crops) = positions_top_left.shape
    if nb_rows != 2:
        raise ValueError('`positions_top_left.shape[0]` is not equal to 2.')
    if len(paths) != nb_crops:
        raise ValueError('`len(paths)` is not equal to `positions_top_left.shape[1]`.')
    for (path, (row, column)) in zip(paths, positions_top_left.T.tolist()):
        save_image(path, crop_repeat_2d(image_uint8, row, column))


def visualize_rotated_luminance(luminance_before_rotation_uint8, is_rotated, positions_top_left, paths):
    """Saves the luminance image, rotated by three quarter turns if required, to `paths[0]` and its crops to `paths[1:]`
    (tools.py:1292-1330)."""
    image_uint8 = numpy.rot90(luminance_before_rotation_uint8, k=3).copy() if is_rotated else luminance_before_rotation_uint8.copy()
    visualize_crops(image_uint8, positions_top_left, paths[1:])
    save_image(paths[0], image_uint8)
