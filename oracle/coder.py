"""ctypes view of the two CPU checkers of the lossless coder -- TEST INFRASTRUCTURE ONLY.

* ``CoderLib('oracle')`` -> oracle/_build/liboracle_coder.so : our plain-C restatement (oracle/coder_oracle.c).
* ``CoderLib('ref')``    -> oracle/_ref/libref_coder.so      : the REAL reference classes
  (kodak_tensorflow/lossless/c++/source/*.cpp) behind oracle/ref_shim.cpp. Built only where /root/reference exists;
  the prebuilt file travels to the GPU box.

Both expose the same functions (prefix ``orc_`` / ``ref_``), mirroring the public methods of the reference's
``LosslessCoder`` (LosslessCoder.h:12-169) and ``compress_lossless`` (compression.h:41-45).
"""
import ctypes
import os

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))

ERROR_NAMES = {0: 'success', 1: 'capacity_error', 2: 'resource_error', 3: 'precision_error',
               4: 'probability_error', 5: 'out_of_range'}
STAGE_MESSAGES = {1: 'during the encoding.',
                  2: 'when stopping the binary arithmetic encoding.',
                  3: 'when starting the binary arithmetic decoding.',
                  4: 'during the decoding.'}


def lib_path(kind):
    if kind == 'oracle':
        return os.path.join(_HERE, '_build', 'liboracle_coder.so')
    if kind == 'ref':
        return os.path.join(_HERE, '_ref', 'libref_coder.so')
    raise ValueError(kind)


def available(kind):
    return os.path.isfile(lib_path(kind))


class CoderLib(object):
    def __init__(self, kind):
        self.kind = kind
        self.prefix = 'orc_' if kind == 'oracle' else 'ref_'
        self.lib = ctypes.CDLL(lib_path(kind))
        c = ctypes
        p = self.prefix
        vp = c.c_void_p

        def sig(name, restype, argtypes):
            f = getattr(self.lib, p + name)
            f.restype = restype
            f.argtypes = argtypes
            return f
        self.count_nb_bits = sig('count_nb_bits', c.c_uint8, [c.c_uint32])
        self._new = sig('new', vp, [c.c_uint32, c.c_uint8, c.POINTER(c.c_double)])
        self._free = sig('free', None, [vp])
        for name in ('occupancy_in_bits_bac', 'occupancy_in_bits_bypass', 'written_bits_bac', 'written_bits_bypass'):
            setattr(self, '_' + name, sig(name, c.c_uint32, [vp]))
        self._bytes_bac = sig('bytes_bac', c.POINTER(c.c_uint8), [vp])
        self._bytes_bypass = sig('bytes_bypass', c.POINTER(c.c_uint8), [vp])
        self._write_sign = sig('write_sign', c.c_int, [vp, c.c_int16])
        self._read_sign = sig('read_sign', c.c_int, [vp, c.POINTER(c.c_int16)])
        self._write_eg0 = sig('write_eg0', c.c_int, [vp, c.c_uint16])
        self._read_eg0 = sig('read_eg0', c.c_int, [vp, c.POINTER(c.c_uint16)])
        self._write_tu = sig('write_truncated_unary', c.c_int, [vp, c.c_uint16])
        self._read_tu = sig('read_truncated_unary', c.c_int, [vp, c.POINTER(c.c_uint16)])
        self._write_sueg0 = sig('write_signed_ueg0', c.c_int, [vp, c.c_int16])
        self._read_sueg0 = sig('read_signed_ueg0', c.c_int, [vp, c.POINTER(c.c_int16)])
        self._stop = sig('stop_bac_encoding', c.c_int, [vp])
        self._start = sig('start_bac_decoding', c.c_int, [vp])
        self._bac_enc = sig('bac_encoding', c.c_int, [vp, c.c_uint8, c.c_double])
        self._bac_dec = sig('bac_decoding', c.c_int, [vp, c.POINTER(c.c_uint8), c.c_double])
        if kind == 'oracle':
            self._compress = sig('compress_lossless', c.c_int,
                                 [c.c_uint32, c.POINTER(c.c_int16), c.POINTER(c.c_int16), c.c_uint8,
                                  c.POINTER(c.c_double), c.POINTER(c.c_uint32), c.POINTER(c.c_int),
                                  c.POINTER(c.c_uint8), c.POINTER(c.c_uint32),
                                  c.POINTER(c.c_uint8), c.POINTER(c.c_uint32)])
        else:
            self._compress = sig('compress_lossless', c.c_int,
                                 [c.c_uint32, c.POINTER(c.c_int16), c.POINTER(c.c_int16), c.c_uint8,
                                  c.POINTER(c.c_double), c.POINTER(c.c_uint32), c.c_char_p, c.c_uint32])

    def coder(self, capacity_bits, probabilities):
        return Coder(self, capacity_bits, probabilities)

    def compress_lossless(self, symbols, probabilities, want_streams=False):
        """compress_lossless (compression.cpp:3-65). Returns (reconstruction, nb_bits[, streams]).

        Raises like the Cython binding maps the C++ exceptions (interface_cython.pyx:6-11 `except +`):
        RuntimeError('Error of type N ...') / ValueError (NULL pointer) / IndexError (out_of_range, L == 0).
        """
        symbols = numpy.ascontiguousarray(symbols, dtype=numpy.int16)
        probabilities = numpy.ascontiguousarray(probabilities, dtype=numpy.float64)
        n = symbols.size
        L = probabilities.size
        if L > 255:
            raise OverflowError('value too large to convert to numpy.uint8_t')
        out = numpy.zeros(n, dtype=numpy.int16)
        nb_bits = ctypes.c_uint32(0)
        c = ctypes
        pin = symbols.ctypes.data_as(c.POINTER(c.c_int16))
        pout = out.ctypes.data_as(c.POINTER(c.c_int16))
        pp = probabilities.ctypes.data_as(c.POINTER(c.c_double))
        if self.kind == 'oracle':
            stage = c.c_int(0)
            cap = (n * max(32, L) + 7) // 8 + 8
            bac = numpy.zeros(cap, dtype=numpy.uint8)
            byp = numpy.zeros(cap, dtype=numpy.uint8)
            bac_bits = c.c_uint32(0)
            byp_bits = c.c_uint32(0)
            rc = self._compress(n, pin, pout, L, pp, c.byref(nb_bits), c.byref(stage),
                                bac.ctypes.data_as(c.POINTER(c.c_uint8)), c.byref(bac_bits),
                                byp.ctypes.data_as(c.POINTER(c.c_uint8)), c.byref(byp_bits))
            if rc == 5:
                raise IndexError('vector::_M_range_check')
            if rc > 0:
                raise RuntimeError('Error of type {0} {1}'.format(rc, STAGE_MESSAGES[stage.value]))
            if rc < 0:
                raise ValueError('One of the three pointers is NULL.')
            if want_streams:
                streams = {'bac_bits': bac_bits.value, 'bypass_bits': byp_bits.value,
                           'bac_bytes': bac[:(bac_bits.value + 7) // 8].copy(),
                           'bypass_bytes': byp[:(byp_bits.value + 7) // 8].copy()}
                return out, nb_bits.value, streams
            return out, nb_bits.value
        msg = c.create_string_buffer(256)
        rc = self._compress(n, pin, pout, L, pp, c.byref(nb_bits), msg, 256)
        if rc == 5:
            raise IndexError(msg.value.decode())
        if rc > 0:
            raise RuntimeError(msg.value.decode())
        if rc < 0:
            raise ValueError(msg.value.decode())
        if want_streams:
            raise NotImplementedError('use Coder(...) to dump the reference streams')
        return out, nb_bits.value


class Coder(object):
    """One LosslessCoder instance (LosslessCoder.h:12-169) of either library."""

    def __init__(self, lib, capacity_bits, probabilities):
        self.lib = lib
        self.probabilities = numpy.ascontiguousarray(probabilities, dtype=numpy.float64)
        self.handle = lib._new(capacity_bits, self.probabilities.size,
                               self.probabilities.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        if not self.handle:
            raise MemoryError()

    def close(self):
        if self.handle:
            self.lib._free(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    @staticmethod
    def _check(rc):
        if rc == 5:
            raise IndexError('out_of_range')
        if rc:
            raise RuntimeError('Error of type {}'.format(rc))

    def write_sign(self, v): self._check(self.lib._write_sign(self.handle, v))
    def write_eg0(self, v): self._check(self.lib._write_eg0(self.handle, v))
    def write_truncated_unary(self, v): self._check(self.lib._write_tu(self.handle, v))
    def write_signed_ueg0(self, v): self._check(self.lib._write_sueg0(self.handle, v))
    def bac_encoding(self, bit, p): self._check(self.lib._bac_enc(self.handle, bit, p))
    def stop_bac_encoding(self): self._check(self.lib._stop(self.handle))
    def start_bac_decoding(self): self._check(self.lib._start(self.handle))

    def read_sign(self, magnitude):
        v = ctypes.c_int16(magnitude)
        self._check(self.lib._read_sign(self.handle, ctypes.byref(v)))
        return v.value

    def read_eg0(self):
        v = ctypes.c_uint16(0)
        self._check(self.lib._read_eg0(self.handle, ctypes.byref(v)))
        return v.value

    def read_truncated_unary(self):
        v = ctypes.c_uint16(0)
        self._check(self.lib._read_tu(self.handle, ctypes.byref(v)))
        return v.value

    def read_signed_ueg0(self):
        v = ctypes.c_int16(0)
        self._check(self.lib._read_sueg0(self.handle, ctypes.byref(v)))
        return v.value

    def bac_decoding(self, p, storage=0):
        v = ctypes.c_uint8(storage)
        self._check(self.lib._bac_dec(self.handle, ctypes.byref(v), p))
        return v.value

    def occupancy_bac(self): return self.lib._occupancy_in_bits_bac(self.handle)
    def occupancy_bypass(self): return self.lib._occupancy_in_bits_bypass(self.handle)
    def written_bac(self): return self.lib._written_bits_bac(self.handle)
    def written_bypass(self): return self.lib._written_bits_bypass(self.handle)

    def bytes_bac(self):
        n = (self.written_bac() + 7) // 8
        return numpy.ctypeslib.as_array(self.lib._bytes_bac(self.handle), shape=(max(n, 1),))[:n].copy()

    def bytes_bypass(self):
        n = (self.written_bypass() + 7) // 8
        return numpy.ctypeslib.as_array(self.lib._bytes_bypass(self.handle), shape=(max(n, 1),))[:n].copy()
