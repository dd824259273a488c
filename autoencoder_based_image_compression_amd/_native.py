"""ctypes bindings of the two in-tree C-ABI libraries (include/eae_coder.h, include/eae_hip.h).

There is NO fallback: if a library is missing the import of the symbol fails loudly. Nothing under ``oracle/`` is
ever touched from here.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, 'lib')

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_i16p = ctypes.POINTER(ctypes.c_int16)
c_u16p = ctypes.POINTER(ctypes.c_uint16)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_intp = ctypes.POINTER(ctypes.c_int)


class NativeLibraryMissing(ImportError):
    pass


def _load(name):
    path = os.path.join(LIB_DIR, name)
    if name == 'libeae_coder.so' and os.environ.get('EAE_CODER_LIB'):
        path = os.environ['EAE_CODER_LIB']      # e.g. the sanitizer build (`make -C csrc sanitize-test`); same ABI, same checks
    if name == 'libeae_hip.so' and os.environ.get('EAE_HIP_LIB'):
        # another build of the same ABI: `test` = lib/libeae_hip_test.so (the whole test-suite on the test build), or a path
        # (an instrumented build, scratch/t3_trace.sh)
        path = os.environ['EAE_HIP_LIB']
        if path == 'test':
            path = os.path.join(LIB_DIR, 'libeae_hip_test.so')
    if not os.path.isfile(path):
        raise NativeLibraryMissing(
            '{0} not found. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(or `make -C autoencoder_based_image_compression_amd/csrc`). There is no CPU fallback.'.format(path))
    return ctypes.CDLL(path)


_lock = threading.Lock()
_coder = None
_hip = None
_hip_test = None

# name -> (restype, argtypes); every symbol include/eae_coder.h declares.
CODER_SYMBOLS = {
    'eae_coder_version': (ctypes.c_char_p, []),
    'eae_coder_count_nb_bits': (ctypes.c_uint8, [ctypes.c_uint32]),
    'eae_coder_stream_capacity_bytes': (ctypes.c_uint32, [ctypes.c_uint32, ctypes.c_uint8]),
    'eae_coder_compress_lossless': (ctypes.c_int, [ctypes.c_uint32, c_i16p, c_i16p, ctypes.c_uint8, c_f64p, c_u32p, c_intp]),
    'eae_coder_encode': (ctypes.c_int, [ctypes.c_uint32, c_i16p, ctypes.c_uint8, c_f64p, c_u8p, c_u32p, c_u8p, c_u32p, c_intp]),
    'eae_coder_decode': (ctypes.c_int, [ctypes.c_uint32, c_i16p, ctypes.c_uint8, c_f64p, c_u8p, ctypes.c_uint32, c_u8p, ctypes.c_uint32, c_intp]),
    'eae_coder_compress_maps': (ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint32, c_i16p, c_i16p, ctypes.c_uint8, c_f64p, c_i32p,
                                               c_u32p, c_i32p, c_i32p, ctypes.c_int, ctypes.c_int]),
    'eae_coder_encode_maps': (ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint32, c_i16p, ctypes.c_uint8, c_f64p, c_i32p,
                                             c_u8p, ctypes.c_uint64, c_u32p, c_u32p, c_i32p, c_i32p, ctypes.c_int]),
    'eae_coder_decode_maps': (ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint32, c_i16p, ctypes.c_uint8, c_f64p, c_i32p,
                                             c_u8p, ctypes.c_uint64, c_u32p, c_u32p, c_i32p, c_i32p, ctypes.c_int]),
    'eae_lossless_coder_new': (ctypes.c_void_p, [ctypes.c_uint32, ctypes.c_uint8, c_f64p]),
    'eae_lossless_coder_free': (None, [ctypes.c_void_p]),
    'eae_lossless_coder_occupancy_in_bits_bac': (ctypes.c_uint32, [ctypes.c_void_p]),
    'eae_lossless_coder_occupancy_in_bits_bypass': (ctypes.c_uint32, [ctypes.c_void_p]),
    'eae_lossless_coder_written_bits_bac': (ctypes.c_uint32, [ctypes.c_void_p]),
    'eae_lossless_coder_written_bits_bypass': (ctypes.c_uint32, [ctypes.c_void_p]),
    'eae_lossless_coder_copy_bac': (ctypes.c_uint32, [ctypes.c_void_p, c_u8p, ctypes.c_uint32]),
    'eae_lossless_coder_copy_bypass': (ctypes.c_uint32, [ctypes.c_void_p, c_u8p, ctypes.c_uint32]),
    'eae_lossless_coder_write_sign': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int16]),
    'eae_lossless_coder_read_sign': (ctypes.c_int, [ctypes.c_void_p, c_i16p]),
    'eae_lossless_coder_write_eg0': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint16]),
    'eae_lossless_coder_read_eg0': (ctypes.c_int, [ctypes.c_void_p, c_u16p]),
    'eae_lossless_coder_write_truncated_unary': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint16]),
    'eae_lossless_coder_read_truncated_unary': (ctypes.c_int, [ctypes.c_void_p, c_u16p]),
    'eae_lossless_coder_write_signed_ueg0': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int16]),
    'eae_lossless_coder_read_signed_ueg0': (ctypes.c_int, [ctypes.c_void_p, c_i16p]),
    'eae_lossless_coder_stop_bac_encoding': (ctypes.c_int, [ctypes.c_void_p]),
    'eae_lossless_coder_start_bac_decoding': (ctypes.c_int, [ctypes.c_void_p]),
    'eae_lossless_coder_bac_encoding': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint8, ctypes.c_double]),
    'eae_lossless_coder_bac_decoding': (ctypes.c_int, [ctypes.c_void_p, c_u8p, ctypes.c_double]),
    'eae_coder_count_binary_decisions': (ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint32, c_i16p, ctypes.c_uint8, c_i64p, c_i64p, ctypes.c_int]),
    'eae_coder_pairwise_row_sums': (ctypes.c_int, [c_f64p, c_i64p, ctypes.c_int64, c_f64p]),
    'eae_crc32c': (ctypes.c_uint32, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]),
}


def _bind(lib, table, optional=False):
    for name, (restype, argtypes) in table.items():
        if optional and not hasattr(lib, name):
            continue
        f = getattr(lib, name)  # AttributeError if the symbol is not exported
        f.restype = restype
        f.argtypes = argtypes
    return lib


def coder():
    """libeae_coder.so with prototypes set."""
    global _coder
    if _coder is None:
        with _lock:
            if _coder is None:
                _coder = _bind(_load('libeae_coder.so'), CODER_SYMBOLS)
    return _coder


def hip():
    """libeae_hip.so with prototypes set (see _native_hip.HIP_SYMBOLS)."""
    global _hip
    if _hip is None:
        with _lock:
            if _hip is None:
                from ._native_hip import EXPERIMENTAL_CODER_SYMBOLS, HIP_SYMBOLS, TEST_HOOK_SYMBOLS
                lib = _bind(_load('libeae_hip.so'), HIP_SYMBOLS)
                # the product library has neither section; a build named by EAE_HIP_LIB may
                _bind(lib, EXPERIMENTAL_CODER_SYMBOLS, optional=True)
                _hip = _bind(lib, TEST_HOOK_SYMBOLS, optional=True)
    return _hip


def hip_test():
    """lib/libeae_hip_test.so: the product's sources compiled with -DEAE_TEST_HOOKS -DEAE_EXPERIMENTAL_CODER (csrc/Makefile). For
    tests (tests/conftest.py: the fixtures `launch_options` and `test_library` make it what `hip()` returns while a test that needs
    the hooks or the experimental coder runs); the package itself never calls this."""
    global _hip_test
    if _hip_test is None:
        with _lock:
            if _hip_test is None:
                from ._native_hip import EXPERIMENTAL_CODER_SYMBOLS, HIP_SYMBOLS, TEST_HOOK_SYMBOLS
                lib = _bind(_load('libeae_hip_test.so'), HIP_SYMBOLS)
                _bind(lib, EXPERIMENTAL_CODER_SYMBOLS)
                _hip_test = _bind(lib, TEST_HOOK_SYMBOLS)
    return _hip_test


def has_experimental_coder():
    """Whether the library `hip()` returns holds the experimental coder round trips (include/eae_hip.h)."""
    return hasattr(hip(), 'eae_hip_coder_roundtrip_trailing')


def ptr(array, ctype_pointer):
    """Pointer to a C-contiguous numpy array's buffer."""
    return array.ctypes.data_as(ctype_pointer)
