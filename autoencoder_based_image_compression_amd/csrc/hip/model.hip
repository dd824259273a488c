// model.hip -- the whole analysis / synthesis transform behind two calls: what `sess.run(entropy_ae.node_y, ...)`
// (kodak_tensorflow/eae/batching.py:96-99) and `sess.run(isolated_decoder.node_reconstruction, ...)` + `tls.cast_bt601`
// (batching.py:49-53) are to the reference. A model object keeps the variables of one trained entropy autoencoder resident
// in HBM in the kernels' layouts (eae_hip_pack_*); encode / decode chain the per-layer launches of this library on the
// caller's stream over a scratch block the caller provides (no allocation, no synchronisation, nothing hidden in globals:
// calls on different streams need different scratch blocks and are otherwise independent). Host code only.
#include <hip/hip_runtime.h>

#include <new>

#include "eae_hip.h"

struct eae_hip_model {
    int learned;
    int has_encoder, has_decoder;
    float* block;                 // one allocation: everything below points into it
    // analysis side
    float *w1, *b1, *g1, *be1, *w2, *b2, *g2, *be2, *w3, *b3, *g3, *be3;
    // synthesis side
    float *g4, *be4, *w4, *b4, *g5, *be5, *w5, *b5, *g6, *be6, *w6;
};

namespace {

constexpr size_t C = EAE_NB_MAPS;
constexpr size_t W1 = 82 * C, W5x5 = 25 * C * C, GAMMA = C * C, VEC = C, W6 = 9 * C * 16;

inline size_t align_up(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

struct ScratchLayout {
    size_t conv_ws, status, a, b, c, total;      // status: one word behind the conv workspace, same place in both layouts
};
// what the launches of one call leave in the conv workspace's error word goes to the block's status word (zeroed with the
// workspace at the start of every call; read by eae_hip_transform_status); the workspace is all zero again afterwards
int collect(char* base, const ScratchLayout& s, void* stream) {
    return eae_hip_conv_workspace_collect(base + s.conv_ws, reinterpret_cast<uint32_t*>(base + s.status), stream);
}
// encode: a = conv_1 output [n][h/4][w/4][128], b = conv_2 output [n][h/8][w/8][128]
ScratchLayout encode_layout(int n, int h, int w) {
    ScratchLayout s{};
    size_t off = align_up(eae_hip_conv_workspace_bytes());
    s.conv_ws = 0;
    s.status = off; off += 256;
    s.a = off; off += align_up((size_t)n * (h / 4) * (w / 4) * C * sizeof(float));
    s.b = off; off += align_up((size_t)n * (h / 8) * (w / 8) * C * sizeof(float));
    s.total = off;
    return s;
}
// decode (h, w = latent size): a = inverse_gdn_4 output [n][h][w][128], b = [n][2h][2w][128], c = [n][4h][4w][128]
ScratchLayout decode_layout(int n, int h, int w) {
    ScratchLayout s{};
    size_t off = align_up(eae_hip_conv_workspace_bytes());
    s.conv_ws = 0;
    s.status = off; off += 256;
    s.a = off; off += align_up((size_t)n * h * w * C * sizeof(float));
    s.b = off; off += align_up((size_t)n * 2 * h * 2 * w * C * sizeof(float));
    s.c = off; off += align_up((size_t)n * 4 * h * 4 * w * C * sizeof(float));
    s.total = off;
    return s;
}

}  // namespace

extern "C" int eae_hip_model_create(const eae_hip_variables* v, int are_bin_widths_learned, eae_hip_model** out) {
    if (!v || !out) return EAE_HIP_BAD_ARGUMENT;
    *out = nullptr;
    const bool fixed = !are_bin_widths_learned;
    // either side may be left out entirely (the reference's IsolatedDecoder holds the decoder's variables only,
    // IsolatedDecoder.py:21-129); a side that is given must be complete
    const float* enc[] = {v->weights_1, v->biases_1, v->gamma_1, v->beta_1, v->weights_2, v->biases_2, v->gamma_2, v->beta_2,
                          v->weights_3, v->biases_3};
    const float* dec[] = {v->weights_4, v->biases_4, v->gamma_5, v->beta_5, v->weights_5, v->biases_5, v->gamma_6, v->beta_6,
                          v->weights_6};
    int enc_given = 0, dec_given = 0;
    for (const float* p : enc) enc_given += p != nullptr;
    for (const float* p : dec) dec_given += p != nullptr;
    if (fixed) { enc_given += (v->gamma_3 != nullptr) + (v->beta_3 != nullptr); dec_given += (v->gamma_4 != nullptr) + (v->beta_4 != nullptr); }
    const int enc_full = 10 + (fixed ? 2 : 0), dec_full = 9 + (fixed ? 2 : 0);
    if ((enc_given != 0 && enc_given != enc_full) || (dec_given != 0 && dec_given != dec_full) || enc_given + dec_given == 0)
        return EAE_HIP_BAD_ARGUMENT;
    const bool has_enc = enc_given != 0, has_dec = dec_given != 0;

    eae_hip_model* m = new (std::nothrow) eae_hip_model();
    if (!m) return (int)hipErrorOutOfMemory;
    m->learned = are_bin_widths_learned ? 1 : 0;
    m->has_encoder = has_enc;
    m->has_decoder = has_dec;
    // resident block: packed layouts, then a staging area of the same size for the TF layouts (freed after packing)
    const size_t floats = W1 + 4 * W5x5 + 6 * GAMMA + 11 * VEC + W6;
    const size_t staging = 81 * C + 4 * W5x5 + 6 * GAMMA + 81 * C;
    float* stage = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->block), floats * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&stage), staging * sizeof(float));
    if (e != hipSuccess) { if (m->block) (void)hipFree(m->block); delete m; return (int)e; }
    float* p = m->block;
    auto take = [&p](size_t count) { float* q = p; p += count; return q; };
    m->w1 = take(W1); m->w2 = take(W5x5); m->w3 = take(W5x5); m->w4 = take(W5x5); m->w5 = take(W5x5); m->w6 = take(W6);
    m->g1 = take(GAMMA); m->g2 = take(GAMMA); m->g3 = take(GAMMA); m->g4 = take(GAMMA); m->g5 = take(GAMMA); m->g6 = take(GAMMA);
    m->b1 = take(VEC); m->be1 = take(VEC); m->b2 = take(VEC); m->be2 = take(VEC); m->b3 = take(VEC); m->be3 = take(VEC);
    m->be4 = take(VEC); m->b4 = take(VEC); m->be5 = take(VEC); m->b5 = take(VEC); m->be6 = take(VEC);
    if (!fixed) { m->g3 = m->be3 = m->g4 = m->be4 = nullptr; }

    int rc = EAE_HIP_OK;
    float* s = stage;
    auto upload = [&](const float* host, size_t count) -> float* {
        float* d = s; s += count;
        if (rc == EAE_HIP_OK) { const hipError_t err = hipMemcpy(d, host, count * sizeof(float), hipMemcpyHostToDevice); if (err != hipSuccess) rc = (int)err; }
        return d;
    };
    auto vec = [&](float* dst, const float* host) {
        if (rc == EAE_HIP_OK && dst) { const hipError_t err = hipMemcpy(dst, host, VEC * sizeof(float), hipMemcpyHostToDevice); if (err != hipSuccess) rc = (int)err; }
    };
    auto step = [&](int status) { if (rc == EAE_HIP_OK && status != EAE_HIP_OK) rc = status; };
    if (has_enc) {
        step(eae_hip_pack_conv9x9s4_weights(upload(v->weights_1, 81 * C), m->w1, nullptr));
        step(eae_hip_pack_conv_weights(upload(v->weights_2, W5x5), m->w2, 25, nullptr));
        step(eae_hip_pack_conv_weights(upload(v->weights_3, W5x5), m->w3, 25, nullptr));
        step(eae_hip_pack_gamma(upload(v->gamma_1, GAMMA), m->g1, nullptr));
        step(eae_hip_pack_gamma(upload(v->gamma_2, GAMMA), m->g2, nullptr));
        if (fixed) step(eae_hip_pack_gamma(upload(v->gamma_3, GAMMA), m->g3, nullptr));
        vec(m->b1, v->biases_1); vec(m->be1, v->beta_1); vec(m->b2, v->biases_2); vec(m->be2, v->beta_2); vec(m->b3, v->biases_3);
        if (fixed) vec(m->be3, v->beta_3);
    }
    if (has_dec) {
        step(eae_hip_pack_tconv_weights(upload(v->weights_4, W5x5), m->w4, 25, nullptr));
        step(eae_hip_pack_tconv_weights(upload(v->weights_5, W5x5), m->w5, 25, nullptr));
        step(eae_hip_pack_tconv9x9s4_weights(upload(v->weights_6, 81 * C), m->w6, nullptr));
        if (fixed) step(eae_hip_pack_gamma(upload(v->gamma_4, GAMMA), m->g4, nullptr));
        step(eae_hip_pack_gamma(upload(v->gamma_5, GAMMA), m->g5, nullptr));
        step(eae_hip_pack_gamma(upload(v->gamma_6, GAMMA), m->g6, nullptr));
        if (fixed) vec(m->be4, v->beta_4);
        vec(m->b4, v->biases_4); vec(m->be5, v->beta_5); vec(m->b5, v->biases_5); vec(m->be6, v->beta_6);
    }
    if (rc == EAE_HIP_OK) { const hipError_t err = hipStreamSynchronize(nullptr); if (err != hipSuccess) rc = (int)err; }
    (void)hipFree(stage);
    if (rc != EAE_HIP_OK) { (void)hipFree(m->block); delete m; return rc; }
    *out = m;
    return EAE_HIP_OK;
}

extern "C" void eae_hip_model_destroy(eae_hip_model* model) {
    if (!model) return;
    (void)hipFree(model->block);
    delete model;
}

extern "C" int eae_hip_model_are_bin_widths_learned(const eae_hip_model* model) { return model ? model->learned : -1; }

extern "C" uint64_t eae_hip_encode_scratch_bytes(int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0 || (h % 16) || (w % 16)) return 0;
    return encode_layout(n, h, w).total;
}

extern "C" uint64_t eae_hip_decode_scratch_bytes(int n, int h_latent, int w_latent) {
    if (n <= 0 || h_latent <= 0 || w_latent <= 0) return 0;
    return decode_layout(n, h_latent, w_latent).total;
}

extern "C" int eae_hip_encode(const eae_hip_model* m, const uint8_t* images, int n, int h, int w, float* latents,
                              void* scratch, uint64_t scratch_bytes, void* stream) {
    if (!m || !m->has_encoder || !images || !latents || !scratch || n <= 0 || h <= 0 || w <= 0) return EAE_HIP_BAD_ARGUMENT;
    if ((h % 16) || (w % 16)) return EAE_HIP_BAD_SHAPE;      // "not divisible by the product of the three strides"
    if ((long)(h / 4) * (w / 4) * 512L > 0x7FFFFFFFL) return EAE_HIP_BAD_SHAPE;   // conv_2's input plane: 32-bit offsets inside an image
    const ScratchLayout s = encode_layout(n, h, w);
    if (scratch_bytes < s.total) return EAE_HIP_BAD_ARGUMENT;
    char* base = static_cast<char*>(scratch);
    hipError_t e = hipMemsetAsync(base + s.conv_ws, 0, s.status + 256, (hipStream_t)stream);   // workspace + status word
    if (e != hipSuccess) return (int)e;
    float* a = reinterpret_cast<float*>(base + s.a);
    float* b = reinterpret_cast<float*>(base + s.b);
    int rc = eae_hip_conv9x9s4_u8(images, m->w1, m->b1, m->g1, m->be1, a, n, h, w, stream);
    if (rc) return rc;
    rc = eae_hip_conv5x5s2_ws(a, m->w2, m->b2, EAE_NORM_GDN, m->g2, m->be2, b, n, h / 4, w / 4, base + s.conv_ws, stream);
    if (rc) return rc;
    if (m->learned) rc = eae_hip_conv5x5s2_ws(b, m->w3, m->b3, EAE_NORM_NONE, nullptr, nullptr, latents, n, h / 8, w / 8, base + s.conv_ws, stream);
    else rc = eae_hip_conv5x5s2_ws(b, m->w3, m->b3, EAE_NORM_GDN, m->g3, m->be3, latents, n, h / 8, w / 8, base + s.conv_ws, stream);
    if (rc) return rc;
    return collect(base, s, stream);
}

extern "C" int eae_hip_transform_status(void* scratch, uint32_t* host_count, void* stream) {
    if (!scratch || !host_count) return EAE_HIP_BAD_ARGUMENT;
    char* word = static_cast<char*>(scratch) + align_up(eae_hip_conv_workspace_bytes());
    hipError_t e = hipMemcpyAsync(host_count, word, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    return (int)e;
}

extern "C" int eae_hip_decode(const eae_hip_model* m, const float* quantized_latents, int n, int h_latent, int w_latent,
                              float* out_f32, uint8_t* out_u8, const uint8_t* ref_u8, uint64_t* sse, void* scratch,
                              uint64_t scratch_bytes, void* stream) {
    if (!m || !m->has_decoder || !quantized_latents || !scratch || n <= 0 || h_latent <= 0 || w_latent <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (!out_f32 && !out_u8 && !ref_u8) return EAE_HIP_BAD_ARGUMENT;
    if (16L * h_latent * w_latent * 512L > 0x7FFFFFFFL) return EAE_HIP_BAD_SHAPE;   // transpose_conv_3's input plane, as in eae_hip_encode
    const ScratchLayout s = decode_layout(n, h_latent, w_latent);
    if (scratch_bytes < s.total) return EAE_HIP_BAD_ARGUMENT;
    char* base = static_cast<char*>(scratch);
    hipError_t e = hipMemsetAsync(base + s.conv_ws, 0, s.status + 256, (hipStream_t)stream);   // workspace + status word
    if (e != hipSuccess) return (int)e;
    float* a = reinterpret_cast<float*>(base + s.a);
    float* b = reinterpret_cast<float*>(base + s.b);
    float* c = reinterpret_cast<float*>(base + s.c);
    const float* t = quantized_latents;
    int rc;
    if (!m->learned) {
        rc = eae_hip_gdn(t, m->g4, m->be4, 1, a, (int64_t)n * h_latent * w_latent, stream);
        if (rc) return rc;
        t = a;
    }
    rc = eae_hip_tconv5x5s2_ws(t, m->w4, m->b4, EAE_NORM_IGDN, m->g5, m->be5, b, n, h_latent, w_latent, base + s.conv_ws, stream);
    if (rc) return rc;
    rc = eae_hip_tconv5x5s2_ws(b, m->w5, m->b5, EAE_NORM_IGDN, m->g6, m->be6, c, n, 2 * h_latent, 2 * w_latent, base + s.conv_ws, stream);
    if (rc) return rc;
    rc = collect(base, s, stream);
    if (rc) return rc;
    return eae_hip_tconv9x9s4_luma(c, m->w6, out_f32, out_u8, ref_u8, sse, n, 4 * h_latent, 4 * w_latent, stream);
}
