"""Per-image lossless coding of the quantized latent variables; mirrors kodak_tensorflow/lossless/compression.py.

`compress_lossless_maps` :11-82 and `rescale_compress_lossless_maps` :84-154, same signatures, return values and
exceptions. Differences in HOW: the float -> int16 symbol conversion, the exception map's histogram AND the coder run on
the MI355X: the 127 coder calls of the reference's Python loop (:67-81) become the launches of
`eae_hip_coder_encode_batch` / `eae_hip_coder_decode_batch` (include/eae_hip.h) over symbols that never leave HBM; only
the per-map bit counts and statuses come back. `code_planar_symbols` is the same thing on the host cores through
`eae_coder_compress_maps` (include/eae_coder.h) for callers whose symbols live in host memory.
"""
import ctypes
import os

import numpy

from ... import _native
from ... import device as dev
from .. import _backend as bk
from ..tools import tools as tls
from . import interface_cython

_probabilities_cache = {}


def load_binary_probabilities(path_to_binary_probabilities):
    """`numpy.load` of the binary probabilities (the reference reloads the file on every call, compression.py:60;
    here the array is cached per (path, mtime, size))."""
    stat = os.stat(path_to_binary_probabilities)
    key = (os.path.abspath(path_to_binary_probabilities), stat.st_mtime_ns, stat.st_size)
    if key not in _probabilities_cache:
        _probabilities_cache.clear()
        _probabilities_cache[key] = numpy.load(path_to_binary_probabilities)
    return _probabilities_cache[key]


def code_planar_symbols(symbols_planar, binary_probabilities, idx_map_exception=-1, nb_threads=0, roundtrip=True,
                        verify_only=False):
    """Codes every map of a batch after the single device -> host copy.

    symbols_planar : int16 (nb_images, nb_maps, map_size), C-contiguous (the layout of `eae_hip_quantize_maps`).
    binary_probabilities : float64 (nb_maps, L).
    roundtrip : encode + decode like `compress_lossless`; with `verify_only` the decoded symbols are compared with the
    input inside the coder threads and not returned (AssertionError on a mismatch, compression.py:146-153).
    Returns (reconstruction int16 like `symbols_planar` or None, nb_bits uint32 (nb_images, nb_maps)); the entry of
    the exception map is 0 here -- its cost comes from its histogram (compression.py:68-75).
    Raises like `compress_lossless_flattened_map` for the first failing map.
    """
    (nb_images, nb_maps, map_size) = symbols_planar.shape
    probabilities = numpy.ascontiguousarray(binary_probabilities, dtype=numpy.float64)
    truncated_unary_length = probabilities.shape[1]
    if truncated_unary_length > 255:
        raise OverflowError('value too large to convert to numpy.uint8_t')   # interface_cython.pyx:49
    n = nb_images*nb_maps
    prob_row = numpy.tile(numpy.arange(nb_maps, dtype=numpy.int32), nb_images)
    if 0 <= idx_map_exception < nb_maps:      # any other index is no exception at all (compression.py:68: `i == idx`)
        prob_row[idx_map_exception::nb_maps] = -1
    nb_bits = numpy.zeros(n, dtype=numpy.uint32)
    status = numpy.zeros(n, dtype=numpy.int32)
    stage = numpy.zeros(n, dtype=numpy.int32)
    keep = roundtrip and not verify_only
    reconstruction = numpy.empty_like(symbols_planar) if keep else None
    mode = (2 if verify_only else 0) if roundtrip else 1
    lib = _native.coder()
    lib.eae_coder_compress_maps(n, map_size, _native.ptr(symbols_planar, _native.c_i16p),
                                _native.ptr(reconstruction, _native.c_i16p) if keep else None,
                                truncated_unary_length, _native.ptr(probabilities, _native.c_f64p),
                                _native.ptr(prob_row, _native.c_i32p), _native.ptr(nb_bits, _native.c_u32p),
                                _native.ptr(status, _native.c_i32p), _native.ptr(stage, _native.c_i32p),
                                mode, nb_threads)
    bad = numpy.flatnonzero(status)
    if bad.size and int(status[bad[0]]) == 6:
        raise AssertionError('\nArrays are not equal\nThe lossless compression has altered the centered quantized data.')
    if bad.size:
        interface_cython.raise_for_status(int(status[bad[0]]), int(stage[bad[0]]))
    return (reconstruction, nb_bits.reshape(nb_images, nb_maps))


def _launch_planar_symbols_device(symbols_planar, binary_probabilities, idx_map_exception, want_reconstruction):
    """The launches of `code_planar_symbols_device`, nothing waited for: (reconstruction device tensor or None, int32 device
    tensor [4, nb_images*nb_maps]: arithmetic-coded bits, bypass bits, status, stage of every map)."""
    import torch
    (nb_images, nb_maps, map_size) = symbols_planar.shape
    probabilities = numpy.ascontiguousarray(binary_probabilities, dtype=numpy.float64)
    truncated_unary_length = probabilities.shape[1]
    if truncated_unary_length > 255:
        raise OverflowError('value too large to convert to numpy.uint8_t')   # interface_cython.pyx:49
    device = symbols_planar.device
    prob_row = numpy.tile(numpy.arange(nb_maps, dtype=numpy.int32), nb_images)
    if 0 <= idx_map_exception < nb_maps:      # any other index is no exception at all (compression.py:68: `i == idx`)
        prob_row[idx_map_exception::nb_maps] = -1
    symbols = symbols_planar.reshape(nb_images*nb_maps, map_size)
    probabilities_device = torch.from_numpy(probabilities).to(device)
    prob_row_device = torch.from_numpy(prob_row).to(device)
    streams = dev.coder_encode_batch(symbols, probabilities_device, prob_row_device, truncated_unary_length)
    encode_results = streams.results.clone()
    reconstruction = None
    if want_reconstruction:
        reconstruction = dev.coder_decode_batch(streams, probabilities_device, prob_row_device)
        skipped = prob_row_device < 0
        reconstruction[skipped] = symbols[skipped]              # the exception map is passed through (compression.py:68-75)
        reconstruction = reconstruction.reshape(nb_images, nb_maps, map_size)
        decode_status = streams.status.clone()
        mismatch = ((reconstruction.reshape(-1, map_size) != symbols).any(dim=1) & (decode_status == 0)).to(torch.int32)*6
        final = torch.where(encode_results[2] != 0, encode_results[2], torch.where(decode_status != 0, decode_status, mismatch))
        stage = torch.where(encode_results[2] != 0, encode_results[3], streams.stage)
    else:
        dev.coder_decode_batch(streams, probabilities_device, prob_row_device, expected=symbols)
        (final, stage) = (streams.status, streams.stage)
    return (reconstruction, torch.stack([encode_results[0], encode_results[1], final, stage]))


def _raise_for_maps(status, stage):
    """Raises what the reference raises for the first map (in map order) whose coder status is not 0."""
    bad = numpy.flatnonzero(status)
    if bad.size and int(status[bad[0]]) == 6:
        raise AssertionError('\nArrays are not equal\nThe lossless compression has altered the centered quantized data.')
    if bad.size:
        interface_cython.raise_for_status(int(status[bad[0]]), int(stage[bad[0]]))


def code_planar_symbols_device(symbols_planar, binary_probabilities, idx_map_exception=-1, want_reconstruction=False):
    """`code_planar_symbols` for symbols that are already on the device (torch int16 (nb_images, nb_maps, map_size)):
    encode, decode and compare without leaving HBM. Returns (reconstruction device tensor or None, nb_bits uint32 numpy
    (nb_images, nb_maps)); raises exactly like `code_planar_symbols`."""
    (nb_images, nb_maps, _) = symbols_planar.shape
    (reconstruction, results) = _launch_planar_symbols_device(symbols_planar, binary_probabilities, idx_map_exception, want_reconstruction)
    host = results.cpu().numpy()      # the one device -> host copy
    _raise_for_maps(host[2], host[3])
    nb_bits = (host[0].astype(numpy.int64) + host[1].astype(numpy.int64)).astype(numpy.uint32)
    return (reconstruction, nb_bits.reshape(nb_images, nb_maps))


def _resident_lossless_costs(record, bin_widths_test, binary_probabilities, idx_map_exception):
    """`rescale_compress_lossless_maps` for EVERY image of a published batch of centred-quantised latents, on its device copy,
    at the first call that is handed one of its images (the reference's harness asks for all of them in turn,
    reconstructing_eae_kodak.py:212-218): one symbol pass, one coder round trip over all the maps of the batch, one histogram
    pass over its exception maps, one device -> host copy. Kept with the batch per (bin widths, probabilities, exception index).
    Returns {'status', 'stage', 'nb_bits' [N, C], 'exception_hist' [N, bins] or None} -- or None when one of the checks that the
    image-by-image path makes before / after coding would not pass for the batch as a whole: the caller then takes that path,
    which raises for exactly the image concerned."""
    import torch
    key = ('lossless_costs', bin_widths_test.astype(numpy.float32).tobytes(), binary_probabilities.tobytes(), binary_probabilities.shape,
           int(idx_map_exception))
    if key not in record.extras:
        tensor = record.tensor
        (n, c) = (tensor.shape[0], tensor.shape[3])
        res = dev.quantize_maps(tensor.view(n, -1, c), bk.to_device(bin_widths_test, numpy.float32), None, want_symbols=True)
        (_, results) = _launch_planar_symbols_device(res['symbols'], binary_probabilities, idx_map_exception, False)
        pieces = [res['checks'], results.reshape(-1)]
        has_exception = 0 <= idx_map_exception < c
        if has_exception:
            (hist, overflow) = dev.symbol_histograms(res['symbols'].view(n*c, -1), tls._FIRST_RADIUS, first_map=idx_map_exception, map_step=c)
            pieces += [overflow, hist.reshape(-1)]
        host = torch.cat(pieces).cpu().numpy()                                   # the one device -> host copy
        (checks, results_host) = (host[:3], host[3:3 + 4*n*c].reshape(4, n, c))
        costs = None
        if checks[0] == 0 and checks[2] == 0 and not (has_exception and host[3 + 4*n*c:3 + 4*n*c + n].any()):
            costs = {'status': results_host[2], 'stage': results_host[3],
                     'nb_bits': (results_host[0].astype(numpy.int64) + results_host[1].astype(numpy.int64)).astype(numpy.uint32),
                     'exception_hist': host[3 + 4*n*c + n:].reshape(n, -1).astype(numpy.int64) if has_exception else None}
        record.extras[key] = costs
    return record.extras[key]


def exception_map_nb_bits(hist_row, map_size):
    """compression.py:73-74: ceil(h*w*discrete_entropy(map, 1.)) from the map's exact symbol histogram."""
    occupied = numpy.flatnonzero(hist_row)
    cumulated_entropy = map_size*tls._entropy_from_hist(hist_row[occupied[0]:occupied[-1] + 1])
    return numpy.ceil(cumulated_entropy).astype(numpy.uint32)


def exception_maps_nb_bits(hist_rows, map_size):
    """`exception_map_nb_bits` of every row of `hist_rows` (int64 [N, bins]) -> int64 [N], bit for bit: the entropies of all rows in
    one pass (`tools._entropies_from_hist_rows`: the element-wise steps at once, every row's sum in `numpy.sum`'s own order, rows near
    a bound through the verbatim expression so that the reference's ValueError still comes from its own comparison)."""
    return numpy.ceil(map_size*tls._entropies_from_hist_rows(hist_rows)).astype(numpy.uint32).astype(numpy.int64)


# The functions are sorted in alphabetic order.

def compress_lossless_maps(ref_int16, path_to_binary_probabilities, idx_map_exception=-1):
    """Compresses without loss each map of signed integers separately (compression.py:11-82).

    Returns (reconstruction int16 (h, w, nb_maps), coding costs uint32 (nb_maps,)).

    Raises
    ------
    TypeError
        If `ref_int16.dtype` is not equal to `numpy.int16`.
    ValueError
        If `binary_probabilities.ndim` is not equal to 2 or its first dimension is not `ref_int16.shape[2]`.
    """
    if ref_int16.dtype != numpy.int16:
        raise TypeError('`ref_int16.dtype` is not equal to `numpy.int16`.')
    (height_map, width_map, nb_maps) = ref_int16.shape
    binary_probabilities = load_binary_probabilities(path_to_binary_probabilities)
    if binary_probabilities.ndim != 2:
        raise ValueError('`binary_probabilities.ndim` is not equal to 2.')
    if binary_probabilities.shape[0] != nb_maps:
        raise ValueError('`binary_probabilities.shape[0]` is not equal to `ref_int16.shape[2]`.')
    planar = bk.to_device(numpy.ascontiguousarray(ref_int16.reshape(height_map*width_map, nb_maps).T)[None])
    (rec_planar, nb_bits) = code_planar_symbols_device(planar, binary_probabilities, idx_map_exception, want_reconstruction=True)
    nb_bits_each_map = nb_bits[0].copy()
    if 0 <= idx_map_exception < nb_maps:
        (hist, radius) = tls._symbol_histograms(planar[0, idx_map_exception:idx_map_exception + 1].contiguous())
        nb_bits_each_map[idx_map_exception] = exception_map_nb_bits(hist[0], height_map*width_map)
    rec_int16 = numpy.ascontiguousarray(bk.to_host(rec_planar[0]).T).reshape(height_map, width_map, nb_maps)
    return (rec_int16, nb_bits_each_map)


def rescale_compress_lossless_maps(centered_quantized_data, bin_widths_test, path_to_binary_probabilities, idx_map_exception=-1):
    """Rescales and compresses without loss each map of centered-quantized data separately (compression.py:84-154).

    Returns the number of bits in the bitstream (int).

    Raises
    ------
    ValueError
        If `bin_widths_test.ndim` is not equal to 1 or its size is not `centered_quantized_data.shape[2]`.
    AssertionError
        If the lossless compression has altered the centered quantized data.
    """
    if bin_widths_test.ndim != 1:
        raise ValueError('`bin_widths_test.ndim` is not equal to 1.')
    (height_map, width_map, nb_maps) = centered_quantized_data.shape
    if bin_widths_test.size != nb_maps:
        raise ValueError('`bin_widths_test.size` is not equal to `centered_quantized_data.shape[2]`.')
    binary_probabilities = load_binary_probabilities(path_to_binary_probabilities)
    if binary_probabilities.ndim != 2:
        raise ValueError('`binary_probabilities.ndim` is not equal to 2.')
    if binary_probabilities.shape[0] != nb_maps:
        raise ValueError('`binary_probabilities.shape[0]` is not equal to `ref_int16.shape[2]`.')
    batch = tls._image_of_published_batch(centered_quantized_data) if centered_quantized_data.dtype == numpy.float32 else None
    if batch is not None:
        # image j of a batch `tls.quantize_per_map` returned: the whole batch is coded once, on its device copy
        costs = _resident_lossless_costs(batch[0], bin_widths_test, binary_probabilities, idx_map_exception)
        if costs is not None:
            j = batch[1]
            _raise_for_maps(costs['status'][j], costs['stage'][j])
            nb_bits_each_map = costs['nb_bits'][j].copy()
            if costs['exception_hist'] is not None:
                nb_bits_each_map[idx_map_exception] = exception_map_nb_bits(costs['exception_hist'][j], height_map*width_map)
            return numpy.sum(nb_bits_each_map).item()
    # compression.py:142 on the device: int16(round(cq / bw)), map-major; checks[0] is the int16 assertion of
    # tools.py:130-132, checks[2] the final `assert_equal` of compression.py:149-153 (symbol*bw must give cq back).
    res = dev.quantize_maps(bk.to_device(centered_quantized_data[None], numpy.float32), bk.to_device(bin_widths_test, numpy.float32),
                            None, want_symbols=True)
    checks = res['checks'].cpu().tolist()
    if checks[0] != 0:
        raise AssertionError('The rounded array elements cannot be represented as 16-bit signed integers.')
    # encode + decode + compare on the device (the decoded-equals-input half of the assert of compression.py:146-153
    # is status 6 of the coder; checks[2] is its symbol*bw == cq half)
    (_, nb_bits) = code_planar_symbols_device(res['symbols'], binary_probabilities, idx_map_exception)
    nb_bits_each_map = nb_bits[0]
    if 0 <= idx_map_exception < nb_maps:
        (hist, radius) = tls._symbol_histograms(res['symbols'][:, idx_map_exception:idx_map_exception + 1].contiguous())
        nb_bits_each_map[idx_map_exception] = exception_map_nb_bits(hist[0], height_map*width_map)
    if checks[2] != 0:
        raise AssertionError('\nArrays are not equal\nThe lossless compression has altered the centered quantized data.')
    return numpy.sum(nb_bits_each_map).item()
