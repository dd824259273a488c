"""Prototypes of every symbol include/eae_hip.h declares: HIP_SYMBOLS is the product (lib/libeae_hip.so exports exactly these,
bound by `_native.hip()`); EXPERIMENTAL_CODER_SYMBOLS and TEST_HOOK_SYMBOLS are the header's two conditional sections, compiled
into lib/libeae_hip_test.so only (`_native.hip_test()`)."""
import ctypes

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64

HIP_SYMBOLS = {
    'eae_hip_version': (ctypes.c_char_p, []),
    'eae_hip_device_info': (_i, [ctypes.c_char_p, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i64)]),
    'eae_hip_publish_sequence': (_i, [_vp, _vp, _vp]),
    'eae_hip_publish_step': (_i, [_vp, _vp, ctypes.c_uint64, ctypes.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp]),
    'eae_hip_partition_info': (_i, [ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    'eae_hip_model_create': (_i, [_vp, _i, ctypes.POINTER(_vp)]),
    'eae_hip_model_destroy': (None, [_vp]),
    'eae_hip_model_are_bin_widths_learned': (_i, [_vp]),
    'eae_hip_encode_scratch_bytes': (ctypes.c_uint64, [_i, _i, _i]),
    'eae_hip_decode_scratch_bytes': (ctypes.c_uint64, [_i, _i, _i]),
    'eae_hip_encode': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, ctypes.c_uint64, _vp]),
    'eae_hip_decode': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, _vp]),
    'eae_hip_conv9x9s4_u8': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'eae_hip_conv5x5s2': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'eae_hip_gdn': (_i, [_vp, _vp, _vp, _i, _vp, _i64, _vp]),
    'eae_hip_tconv5x5s2': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'eae_hip_conv_workspace_bytes': (ctypes.c_uint64, []),
    'eae_hip_conv_workspace_collect': (_i, [_vp, _vp, _vp]),
    'eae_hip_transform_status': (_i, [_vp, ctypes.POINTER(ctypes.c_uint32), _vp]),
    'eae_hip_conv5x5s2_ws': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    'eae_hip_tconv5x5s2_ws': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    'eae_hip_tconv9x9s4_luma': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'eae_hip_pack_tconv9x9s4_weights': (_i, [_vp, _vp, _vp]),
    'eae_hip_pack_conv_weights': (_i, [_vp, _vp, _i, _vp]),
    'eae_hip_pack_conv9x9s4_weights': (_i, [_vp, _vp, _vp]),
    'eae_hip_pack_tconv_weights': (_i, [_vp, _vp, _i, _vp]),
    'eae_hip_pack_gamma': (_i, [_vp, _vp, _vp]),
    'eae_hip_quantize_maps': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'eae_hip_map_means': (_i, [_vp, _vp, _i64, _i, _vp]),
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
    'eae_hip_map_minmax': (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    'eae_hip_floor_histograms': (_i, [_vp, _vp, _i, _vp, _i64, _i, _vp]),
    'eae_hip_coder_pack_streams': (_i, [ctypes.c_uint32, _vp, ctypes.c_uint64, _vp, _vp, _vp, _vp, _vp]),
    'eae_hip_coder_unpack_streams': (_i, [ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, _vp]),
    'eae_hip_dequantize_maps': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'eae_hip_conv5x5s2_latent': (_i, [_vp]*15 + [_i, _i, _i, _vp, _vp]),
    'eae_hip_latent_stage': (_i, [_vp]*13 + [_i, _i, _vp]),
    'eae_hip_rgb_to_ycbcr': (_i, [_vp, _vp, _vp, _i64, _vp]),
    'eae_hip_publish_to_host': (_i, [_vp, _vp, ctypes.c_uint64, _vp]),
    'eae_hip_symbol_histograms_strided': (_i, [_vp, _vp, _i, _vp, _i, _i, _i64, _i64, _vp]),
    'eae_hip_cast_bt601': (_i, [_vp, _vp, _i64, _vp]),
    'eae_hip_sse_u8': (_i, [_vp, _vp, _vp, _i, _i64, _vp]),
    'eae_hip_svhn_dense_f64': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'eae_hip_svhn_preprocess': (_i, [_vp, _vp, ctypes.c_double, _vp, _i, _i, _vp]),
    'eae_hip_svhn_quantize_f64': (_i, [_vp, ctypes.c_double, _vp, _vp, _vp, _i64, _vp]),
    'eae_hip_svhn_symbol_range': (_i, [_vp, _i64, _vp, _vp]),
    'eae_hip_svhn_symbol_histogram': (_i, [_vp, _i64, ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp]),
    'eae_hip_svhn_postprocess': (_i, [_vp, ctypes.c_double, _vp, _vp, _vp, _vp, _i, _i, _vp]),
}

# include/eae_hip.h, #ifdef EAE_EXPERIMENTAL_CODER
EXPERIMENTAL_CODER_SYMBOLS = {
    'eae_hip_coder_trailing_workspace_bytes': (ctypes.c_uint64, [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint8]),
    'eae_hip_coder_roundtrip_trailing': (_i, [ctypes.c_uint32, ctypes.c_uint32, _vp, ctypes.c_uint8, _vp, _vp, _vp, ctypes.c_uint64,
                                              _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_uint32, _vp]),
    'eae_hip_coder_roundtrip_fused': (_i, [ctypes.c_uint32, ctypes.c_uint32, _vp, ctypes.c_uint8, _vp, _vp, _vp, ctypes.c_uint64,
                                           _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, _vp]),
}

# include/eae_hip.h, #ifdef EAE_TEST_HOOKS
TEST_HOOK_SYMBOLS = {
    'eae_hip_debug_set_stamp_buffer': (_i, [_vp]),
    'eae_hip_debug_reload_launch_options': (_i, []),
    'eae_hip_debug_set_split_mute': (_i, [_i]),
    'eae_hip_debug_check_mid_forms': (_i, [_i, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, _vp, _vp]),
}
