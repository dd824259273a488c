"""Prototypes of every symbol include/eae_hip.h declares (bound by `_native.hip()`)."""
import ctypes

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64

HIP_SYMBOLS = {
    'eae_hip_version': (ctypes.c_char_p, []),
    'eae_hip_device_info': (_i, [ctypes.c_char_p, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i64)]),
    'eae_hip_conv9x9s4_u8': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'eae_hip_conv5x5s2': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'eae_hip_gdn': (_i, [_vp, _vp, _vp, _i, _vp, _i64, _vp]),
    'eae_hip_tconv5x5s2': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'eae_hip_tconv9x9s4_luma': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'eae_hip_pack_tconv9x9s4_weights': (_i, [_vp, _vp, _vp]),
    'eae_hip_pack_conv_weights': (_i, [_vp, _vp, _i, _vp]),
    'eae_hip_pack_conv9x9s4_weights': (_i, [_vp, _vp, _vp]),
    'eae_hip_pack_tconv_weights': (_i, [_vp, _vp, _i, _vp]),
    'eae_hip_pack_gamma': (_i, [_vp, _vp, _vp]),
    'eae_hip_quantize_maps': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'eae_hip_map_sums': (_i, [_vp, _vp, _i64, _i, _vp]),
    'eae_hip_nonzero_flags': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'eae_hip_cast_int16': (_i, [_vp, _vp, _i64, _vp, _vp]),
    'eae_hip_symbol_histograms': (_i, [_vp, _vp, _i, _vp, _i, _i, _vp]),
    'eae_hip_coder_stream_stride_bytes': (ctypes.c_uint64, [ctypes.c_uint32, ctypes.c_uint8]),
    'eae_hip_coder_compress_maps': (_i, [ctypes.c_uint32, ctypes.c_uint32, _vp, _vp, ctypes.c_uint8, _vp, _vp, _vp, ctypes.c_uint64,
                                         _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'eae_hip_coder_decode_maps': (_i, [ctypes.c_uint32, ctypes.c_uint32, _vp, ctypes.c_uint8, _vp, _vp, _vp, ctypes.c_uint64,
                                       _vp, _vp, _vp, _vp, _i, _vp]),
    'eae_hip_coder_verify_maps': (_i, [ctypes.c_uint32, ctypes.c_uint32, _vp, ctypes.c_uint8, _vp, _vp, _vp, ctypes.c_uint64,
                                       _vp, _vp, _vp, _vp, _i, _vp]),
    'eae_hip_coder_workspace_bytes': (ctypes.c_uint64, [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint8]),
    'eae_hip_coder_encode_batch': (_i, [ctypes.c_uint32, ctypes.c_uint32, _vp, ctypes.c_uint8, _vp, _vp, _vp, ctypes.c_uint64,
                                        _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, _vp]),
    'eae_hip_coder_decode_batch': (_i, [ctypes.c_uint32, ctypes.c_uint32, _vp, _vp, ctypes.c_uint8, _vp, _vp, _vp, ctypes.c_uint64,
                                        _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, _vp]),
    'eae_hip_publish_to_host': (_i, [_vp, _vp, ctypes.c_uint64, _vp]),
    'eae_hip_debug_set_stamp_buffer': (_i, [_vp]),
    'eae_hip_cast_bt601': (_i, [_vp, _vp, _i64, _vp]),
    'eae_hip_sse_u8': (_i, [_vp, _vp, _vp, _i, _i64, _vp]),
    'eae_hip_svhn_dense_f64': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'eae_hip_svhn_preprocess': (_i, [_vp, _vp, ctypes.c_double, _vp, _i, _i, _vp]),
    'eae_hip_svhn_quantize_f64': (_i, [_vp, ctypes.c_double, _vp, _vp, _vp, _i64, _vp]),
    'eae_hip_svhn_symbol_range': (_i, [_vp, _i64, _vp, _vp]),
    'eae_hip_svhn_symbol_histogram': (_i, [_vp, _i64, ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp]),
    'eae_hip_svhn_postprocess': (_i, [_vp, ctypes.c_double, _vp, _vp, _vp, _vp, _i, _i, _vp]),
}
