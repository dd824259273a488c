"""Hot-path subset of the reference's `svhn/tools/tools.py` on the MI355X: `cast_float_to_uint8` :138-166,
`count_symbols` :168-231, `discrete_entropy` :289-, `leaky_relu` :676-694, `mean_psnr` :812-865, `quantization`
:1062-1095. Same names, arguments and exceptions; numpy in, numpy out."""
import numpy

from ... import device as dev
from ...kodak import _backend as bk


def _is_floating(array):
    return numpy.issubdtype(array.dtype, numpy.floating)


def cast_float_to_uint8(array_float):
    """Clips to [0., 255.], rounds half to even, casts to `numpy.uint8` (tools.py:138-166)."""
    if not _is_floating(array_float):
        raise TypeError('`array_float.dtype` is not smaller than `numpy.float` in type hierarchy.')
    flat = numpy.ascontiguousarray(array_float, dtype=numpy.float64).reshape(1, -1)
    zeros = numpy.zeros(flat.shape[1])
    (out, _) = dev.svhn_postprocess(bk.to_device(flat), 1., bk.to_device(zeros))
    return bk.to_host(out).reshape(array_float.shape)


def count_symbols(quantized_samples, bin_width):
    """Number of occurrences of each symbol from the smallest to the largest quantized sample (tools.py:168-231)."""
    if numpy.size(quantized_samples) == 0:
        raise ValueError('zero-size array to reduction operation minimum which has no identity')
    (_, symbols, checks) = dev.svhn_quantize(bk.to_device(quantized_samples, numpy.float64), bin_width, want_q=False, want_symbols=True)
    checks = checks.cpu().tolist()
    if checks[1] != 0:
        raise AssertionError('\nArrays are not almost equal to 10 decimals\nThe quantization was omitted.')
    if checks[0] != 0:
        raise ValueError('A symbol does not fit in 32 bits.')
    (hist, _) = dev.svhn_symbol_histogram(symbols)
    return hist


def discrete_entropy(quantized_samples, bin_width):
    """Entropy of the quantized samples (tools.py:289-): -sum f log2 f over the non-empty symbols, float64."""
    hist = count_symbols(quantized_samples, bin_width)
    hist_non_zero = numpy.extract(hist != 0, hist)
    frequency = hist_non_zero.astype(numpy.float64)/numpy.sum(hist_non_zero)
    disc_entropy = -numpy.sum(frequency*numpy.log2(frequency))
    if disc_entropy < 0.:
        raise ValueError('The entropy is not positive.')
    if disc_entropy > numpy.log2(hist_non_zero.size):
        raise ValueError('The entropy is not smaller than its upper bound.')
    return disc_entropy


def leaky_relu(input):
    """Leaky ReLU with slope 0.1 (tools.py:676-694); host helper, the device applies it inside the dense layers."""
    coefficients = numpy.ones(input.shape)
    coefficients[input < 0.] = 0.1
    return coefficients*input


def mean_psnr(reference_uint8, reconstruction_uint8):
    """Mean over the images of the PSNR between each image and its reconstruction (tools.py:812-865)."""
    if reference_uint8.dtype != numpy.uint8:
        raise TypeError('`reference_uint8.dtype` is not equal to `numpy.uint8`.')
    if reconstruction_uint8.dtype != numpy.uint8:
        raise TypeError('`reconstruction_uint8.dtype` is not equal to `numpy.uint8`.')
    if reference_uint8.ndim != 2:
        raise ValueError('`reference_uint8.ndim` is not equal to 2.')
    if reference_uint8.shape != reconstruction_uint8.shape:
        raise ValueError('`reference_uint8.shape` is not equal to `reconstruction_uint8.shape`.')
    sse = bk.to_host(dev.sse_u8(bk.to_device(reference_uint8), bk.to_device(reconstruction_uint8)))
    return mean_psnr_from_sse(sse, reference_uint8.shape[1])


def mean_psnr_from_sse(sse, nb_pixels):
    """tools.py:859-865 from the exact per-image integer sums of squared errors."""
    mse = sse.astype(numpy.float64)/nb_pixels
    if numpy.any(mse == 0.):
        raise ValueError('The mean square error between a reference image and its reconstruction is equal to 0.')
    return numpy.mean(10.*numpy.log10((255.**2)/mse))


def quantization(samples, bin_width):
    """Uniform scalar quantization with one bin width, float64 (tools.py:1062-1095)."""
    if not _is_floating(samples):
        raise TypeError('`samples.dtype` is not smaller than `numpy.float` in type hierarchy.')
    if bin_width <= 0.:
        raise ValueError('The quantization bin width is not strictly positive.')
    (q, _, _) = dev.svhn_quantize(bk.to_device(samples, numpy.float64), bin_width)
    return bk.to_host(q).reshape(samples.shape)
