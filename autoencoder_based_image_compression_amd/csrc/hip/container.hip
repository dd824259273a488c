// container.hip -- device ends of a self-describing container for the coded latents.
// The reference never serialises: its streams die inside compress_lossless (lossless/c++/source/compression.cpp:27-64)
// and the decoder side restarts from the float array the encoder side still holds (reconstructing_eae_kodak.py:192-207).
// A standalone decoder needs (1) the streams packed back to back, (2) the inverse of the symbol conversion of
// lossless/compression.py:142: centred-quantised value = symbol * bin width (tools.py:929 computes bw * round(x / bw), and
// the symbol IS that rounded value), de-centred by the map mean (reconstructing_eae_kodak.py:192).
#include "common.h"

namespace {

// Streams of map m: arithmetic-coded bytes at streams[m * stride], bypass bytes at +stride / 2 (include/eae_coder.h).
// offsets[2m], offsets[2m + 1]: byte offsets of the two pieces in the payload. A piece never exceeds its half of the
// region: bit counts come from an untrusted header when unpacking, so the copy is clamped to stride / 2 bytes (the host
// side rejects such a header before it gets here, container.read_header).
template <bool PACK>
__global__ __launch_bounds__(64) void move_streams_kernel(uint8_t* __restrict__ streams, uint64_t stride, uint8_t* __restrict__ payload,
                                                          const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ bac_bits,
                                                          const uint32_t* __restrict__ bypass_bits) {
    const uint32_t m = blockIdx.x;
    for (int piece = 0; piece < 2; ++piece) {
        uint64_t bytes = ((uint64_t)(piece ? bypass_bits[m] : bac_bits[m]) + 7u) >> 3;
        if (bytes > stride / 2) bytes = stride / 2;
        uint8_t* region = streams + (uint64_t)m * stride + (piece ? stride / 2 : 0);
        uint8_t* packed = payload + offsets[2 * m + piece];
        for (uint64_t i = threadIdx.x; i < bytes; i += 64) {
            if (PACK) packed[i] = region[i];
            else region[i] = packed[i];
        }
    }
}

// symbols planar [N][C][hw] int16 -> NHWC float32 (bw[c] * symbol) + mean[c], 64 pixels x 128 maps per block through LDS
constexpr int PIX = 64;
__global__ __launch_bounds__(256) void dequantize_kernel(const int16_t* __restrict__ symbols, const float* __restrict__ bin_widths,
                                                         const float* __restrict__ map_mean, float* __restrict__ cq_out,
                                                         float* __restrict__ shifted_out, int hw, int chunks) {
    __shared__ int16_t tile[EAE_C][PIX + 2];
    const int tid = threadIdx.x;
    const int img = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    for (int i = tid; i < EAE_C * PIX; i += 256) {
        const int ch = i >> 6, px = i & 63;
        const int pix = chunk * PIX + px;
        tile[ch][px] = pix < hw ? symbols[((size_t)img * EAE_C + ch) * hw + pix] : (int16_t)0;
    }
    __syncthreads();
    const int c = tid & 127, half = tid >> 7;
    const float bw = bin_widths[c];
    const float m = map_mean ? map_mean[c] : 0.f;
    for (int i = 0; i < PIX / 2; ++i) {
        const int px = half * (PIX / 2) + i;
        const int pix = chunk * PIX + px;
        if (pix < hw) {
            const float cq = bw * (float)tile[c][px];          // tools.py:929 with round(x / bw) == symbol
            const size_t idx = ((size_t)img * hw + pix) * EAE_C + c;
            if (cq_out) cq_out[idx] = cq;
            if (shifted_out) shifted_out[idx] = cq + m;        // reconstructing_eae_kodak.py:192
        }
    }
}

}  // namespace

extern "C" int eae_hip_coder_pack_streams(uint32_t n_maps, const uint8_t* streams, uint64_t stride, const uint32_t* bac_bits,
                                          const uint32_t* bypass_bits, const uint64_t* offsets, uint8_t* payload, void* stream) {
    if (!streams || !bac_bits || !bypass_bits || !offsets || !payload) return EAE_HIP_BAD_ARGUMENT;
    if (n_maps == 0) return EAE_HIP_OK;
    hipLaunchKernelGGL(move_streams_kernel<true>, dim3(n_maps), dim3(64), 0, (hipStream_t)stream, const_cast<uint8_t*>(streams),
                       stride, payload, offsets, bac_bits, bypass_bits);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_coder_unpack_streams(uint32_t n_maps, const uint8_t* payload, const uint64_t* offsets,
                                            const uint32_t* bac_bits, const uint32_t* bypass_bits, uint8_t* streams,
                                            uint64_t stride, void* stream) {
    if (!streams || !bac_bits || !bypass_bits || !offsets || !payload || stride < 2) return EAE_HIP_BAD_ARGUMENT;
    if (n_maps == 0) return EAE_HIP_OK;
    hipLaunchKernelGGL(move_streams_kernel<false>, dim3(n_maps), dim3(64), 0, (hipStream_t)stream, streams, stride,
                       const_cast<uint8_t*>(payload), offsets, bac_bits, bypass_bits);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_dequantize_maps(const int16_t* symbols_planar, const float* bin_widths, const float* map_mean,
                                       float* cq_out, float* shifted_out, int n, int hw, int c, void* stream) {
    if (!symbols_planar || !bin_widths || (!cq_out && !shifted_out) || n <= 0 || hw <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (c != EAE_C) return EAE_HIP_BAD_SHAPE;
    const int chunks = (hw + PIX - 1) / PIX;
    hipLaunchKernelGGL(dequantize_kernel, dim3(n * chunks), dim3(256), 0, (hipStream_t)stream, symbols_planar, bin_widths,
                       map_mean, cq_out, shifted_out, hw, chunks);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
