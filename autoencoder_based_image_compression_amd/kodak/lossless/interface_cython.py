"""Python binding of the host lossless coder; stands where ``lossless/interface_cython.pyx`` stands in the reference.

Reference: kodak_tensorflow/lossless/interface_cython.pyx:13-59 (``compress_lossless_flattened_map``), which calls
``compress_lossless`` (lossless/c++/source/compression.cpp:3-65) through Cython with ``except +``. Here the same
call goes through ctypes into ``eae_coder_compress_lossless`` (include/eae_coder.h) and integer error codes are
mapped back to the Python exceptions Cython would have produced.
"""
import ctypes

import numpy

from ... import _native

_STAGE_MESSAGES = {
    1: 'during the encoding.',                           # compression.cpp:34
    2: 'when stopping the binary arithmetic encoding.',  # compression.cpp:40
    3: 'when starting the binary arithmetic decoding.',  # compression.cpp:54
    4: 'during the decoding.',                           # compression.cpp:60
}


def raise_for_status(status, stage):
    """Maps an ``eae_error_code`` to the exception the reference binding raises (Cython ``except +`` table)."""
    if status == 0:
        return
    if status == -1:
        raise ValueError('One of the three pointers is NULL.')        # std::invalid_argument, compression.cpp:11
    if status == -2:
        raise MemoryError('std::bad_alloc')
    if status == 5:
        raise IndexError('vector::_M_range_check')                    # std::out_of_range, LosslessCoder.cpp:173
    raise RuntimeError('Error of type {0} {1}'.format(status, _STAGE_MESSAGES.get(stage, '')))


def _check_buffer(array, dtype, name):
    # Cython's typed-buffer acquisition (interface_cython.pyx:13-14): wrong type -> TypeError, wrong
    # dtype / ndim -> ValueError.
    if not isinstance(array, numpy.ndarray):
        raise TypeError("Argument '{0}' has incorrect type (expected numpy.ndarray, got {1})".format(name, type(array).__name__))
    if array.ndim != 1:
        raise ValueError('Buffer has wrong number of dimensions (expected 1, got {})'.format(array.ndim))
    if array.dtype != dtype:
        raise ValueError("Buffer dtype mismatch, expected '{0}' but got '{1}'".format(numpy.dtype(dtype).name, array.dtype.name))


def compress_lossless_flattened_map(ref_map_int16, probabilities):
    """Compresses without loss a flattened map of signed integers.

    Parameters
    ----------
    ref_map_int16 : numpy.ndarray
        1D array with data-type `numpy.int16`. Flattened map of signed integers.
    probabilities : numpy.ndarray
        1D array with data-type `numpy.float64`. Its ith element is the probability that the ith binary
        decision is 0 in the truncated unary prefix (at most 255 elements).

    Returns
    -------
    tuple
        (1D `numpy.int16` reconstruction after the compression without loss, coding cost in bits as int).

    Raises
    ------
    Same exceptions as the reference binding: RuntimeError('Error of type N ...'), ValueError, IndexError,
    OverflowError (more than 255 probabilities, pyx:49).
    """
    _check_buffer(ref_map_int16, numpy.int16, 'ref_map_int16')
    _check_buffer(probabilities, numpy.float64, 'probabilities')
    size = ref_map_int16.size
    if probabilities.size > 255:
        raise OverflowError('value too large to convert to numpy.uint8_t')   # pyx:49
    if size == 0 or probabilities.size == 0:
        raise IndexError('Out of bounds on buffer access (axis 0)')           # &ref_map_int16[0], pyx:55-58
    src = numpy.ascontiguousarray(ref_map_int16)
    probs = numpy.ascontiguousarray(probabilities)
    rec_map_int16 = numpy.zeros(size, dtype=numpy.int16)                     # pyx:50
    nb_bits = ctypes.c_uint32(0)
    stage = ctypes.c_int(0)
    status = _native.coder().eae_coder_compress_lossless(
        size, _native.ptr(src, _native.c_i16p), _native.ptr(rec_map_int16, _native.c_i16p),
        probs.size, _native.ptr(probs, _native.c_f64p), ctypes.byref(nb_bits), ctypes.byref(stage))
    raise_for_status(status, stage.value)
    return (rec_map_int16, nb_bits.value)


def encode_flattened_map(ref_map_int16, probabilities):
    """Encode only: returns (bac_bytes, bac_bits, bypass_bytes, bypass_bits) -- the streams the reference discards."""
    _check_buffer(ref_map_int16, numpy.int16, 'ref_map_int16')
    _check_buffer(probabilities, numpy.float64, 'probabilities')
    if probabilities.size > 255:
        raise OverflowError('value too large to convert to numpy.uint8_t')
    lib = _native.coder()
    src = numpy.ascontiguousarray(ref_map_int16)
    probs = numpy.ascontiguousarray(probabilities)
    cap = lib.eae_coder_stream_capacity_bytes(src.size, probs.size) + 16
    bac = numpy.zeros(cap, dtype=numpy.uint8)
    byp = numpy.zeros(cap, dtype=numpy.uint8)
    bac_bits = ctypes.c_uint32(0)
    byp_bits = ctypes.c_uint32(0)
    stage = ctypes.c_int(0)
    dummy = numpy.zeros(1, dtype=numpy.int16)
    status = lib.eae_coder_encode(src.size, _native.ptr(src if src.size else dummy, _native.c_i16p), probs.size,
                                  _native.ptr(probs if probs.size else numpy.zeros(1), _native.c_f64p),
                                  _native.ptr(bac, _native.c_u8p), ctypes.byref(bac_bits),
                                  _native.ptr(byp, _native.c_u8p), ctypes.byref(byp_bits), ctypes.byref(stage))
    raise_for_status(status, stage.value)
    return (bac[:(bac_bits.value + 7)//8].copy(), bac_bits.value, byp[:(byp_bits.value + 7)//8].copy(), byp_bits.value)


def decode_flattened_map(size, probabilities, bac_bytes, bac_bits, bypass_bytes, bypass_bits):
    """Inverse of `encode_flattened_map`."""
    lib = _native.coder()
    probs = numpy.ascontiguousarray(probabilities, dtype=numpy.float64)
    out = numpy.zeros(max(size, 1), dtype=numpy.int16)
    bac = numpy.concatenate([numpy.ascontiguousarray(bac_bytes, dtype=numpy.uint8), numpy.zeros(8, numpy.uint8)])
    byp = numpy.concatenate([numpy.ascontiguousarray(bypass_bytes, dtype=numpy.uint8), numpy.zeros(8, numpy.uint8)])
    stage = ctypes.c_int(0)
    status = lib.eae_coder_decode(size, _native.ptr(out, _native.c_i16p), probs.size,
                                  _native.ptr(probs if probs.size else numpy.zeros(1), _native.c_f64p),
                                  _native.ptr(bac, _native.c_u8p), bac_bits,
                                  _native.ptr(byp, _native.c_u8p), bypass_bits, ctypes.byref(stage))
    raise_for_status(status, stage.value)
    return out[:size]
