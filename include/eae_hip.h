/*
 * eae_hip.h -- C ABI of the gfx950 (MI355X / CDNA4) device path (libeae_hip.so).
 *
 * The reference has no native code for the transforms: its hot loops are the TensorFlow kernels that
 * `sess.run(entropy_ae.node_y)` (kodak_tensorflow/eae/batching.py:96-99) and
 * `sess.run(isolated_decoder.node_reconstruction)` (batching.py:49-52) dispatch, plus numpy helpers
 * (kodak_tensorflow/tools/tools.py). Each entry point below replaces one TF op / numpy helper (or a fused run of
 * them) and cites it. All pointers are DEVICE pointers unless named host_*; `stream` is a hipStream_t passed as
 * void* (NULL = default stream). Layout is the reference's: NHWC, C-contiguous, float32 activations.
 * Every function returns 0 on success, a negative EAE_HIP_* code for bad arguments, or a positive hipError_t.
 * Launches are asynchronous on `stream`.
 *
 * NUMERICS CONTRACT (mirrored by oracle/transforms_oracle.c, checked bit-for-bit in tests):
 *  - every dot product is ONE f32 fused-multiply-add chain in a fixed order (conv / transposed conv: input channels in
 *    blocks of 32 (outer), then the taps row-major, then the channel inside the block; GDN: channel ascending),
 *    started from +0, bias/beta added afterwards -- this is what v_mfma_f32_32x32x2_f32 / 16x16x4_f32 compute
 *    (exact f32 FMA chain, k ascending);
 *  - division, sqrt are correctly rounded; rounding to integer is round-half-to-even (numpy.round).
 */
#ifndef EAE_HIP_H
#define EAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    EAE_HIP_OK = 0,
    EAE_HIP_BAD_ARGUMENT = -1,  /* NULL pointer, non-positive size */
    EAE_HIP_BAD_SHAPE = -2      /* spatial size not accepted by the op (see each op) */
};

enum { EAE_NORM_NONE = 0, EAE_NORM_GDN = 1, EAE_NORM_IGDN = 2 };

#define EAE_NB_MAPS 128 /* eae/graph/constants.py:42-44 (NB_MAPS_1/2/3) */

const char* eae_hip_version(void);
/* Fills name (e.g. "gfx950:sramecc+:xnack-"), CU count; returns 0, or a hipError_t when no device is usable. */
int eae_hip_device_info(char* name, int name_cap, int* compute_units, int* clock_mhz, int64_t* hbm_bytes);
/* The compute partition the current logical device is, as far as it matters here: its compute units, the XCDs that makes on
 * gfx950 (32 CUs each; 0 on another architecture), and whether it is one whole MI355X (256 CUs, mode SPX) -- the shape the
 * conv launches' XCD-aware tile order and the sizing of their cut tiles are built on. On a partition (DPX / QPX / CPX) every
 * result is the same, the launches keep whole tiles, and a note goes to stderr once. */
int eae_hip_partition_info(int* compute_units, int* xcds, int* whole_device);

/* Small results for the host (bit counts, statuses, histograms): a kernel copies `bytes` (multiple of 4) from device
 * memory into PINNED, device-mapped host memory (hipHostMalloc / torch pin_memory) in stream order; the host reads them
 * after synchronising an event recorded behind it. Unlike hipMemcpyAsync this never blocks the calling thread. */
int eae_hip_publish_to_host(const void* src_device, void* dst_host_mapped, uint64_t bytes, void* stream);
/* A step counter for a host thread: in stream order, *counter_device (uint32, device memory) is incremented and the new value is
 * left in *word_host_mapped (uint32, pinned host memory), behind everything the stream did before -- in particular behind an
 * eae_hip_publish_to_host in front of it. The host, which knows how many times it submitted the step, waits for that value with
 * plain loads: no event to record, query or wait on (capturable into a hipGraph: every replay bumps the counter). */
int eae_hip_publish_sequence(void* counter_device, void* word_host_mapped, void* stream);
/* The end of one side of a step in ONE launch (what codec.BatchCodec enqueues behind a batch's coder and behind its synthesis
 * transform; each launch costs the submitting thread ~10 us): in stream order, (1) when `conv_workspace` is given, what
 * eae_hip_conv_workspace_collect does with it and `error_word` (which must lie inside the source block: it is copied in (2); the block
 * is then at most 256 KB); (2) eae_hip_publish_to_host of `bytes` from `src_device`; (3) the source's bytes from `clear_from_byte` on are
 * zeroed -- the accumulators of the next step that uses the block (pass `bytes` to clear nothing); (4) eae_hip_publish_sequence on
 * `counter_device` / `word_host_mapped`, behind the copy. `tickets_device`: a zeroed uint32 in device memory, left zeroed (the copying
 * blocks count themselves in; steps that may run concurrently need their own). */
int eae_hip_publish_step(void* src_device, void* dst_host_mapped, uint64_t bytes, uint64_t clear_from_byte, void* conv_workspace,
                         uint32_t* error_word, void* tickets_device, void* counter_device, void* word_host_mapped, void* stream);

/* ---- whole-path entry points (csrc/hip/model.hip) --------------------------------------------------------------------
 * What `sess.run(entropy_ae.node_y, feed_dict={node_visible_units: batch})` (eae/batching.py:96-99) and
 * `sess.run(isolated_decoder.node_reconstruction, feed_dict={node_quantized_y: batch})` followed by `tls.cast_bt601`
 * (batching.py:49-53) are to the reference: one model object per trained entropy autoencoder, one call per mini-batch.
 * The numpy steps the reference runs between the two (centring, quantiser, statistics: reconstructing_eae_kodak.py:170-192)
 * are eae_hip_latent_stage / eae_hip_quantize_maps below; the per-layer ops that encode / decode chain follow further down.
 *
 * eae_hip_variables: HOST pointers to the variables of `EntropyAutoencoder` / `IsolatedDecoder` in TensorFlow's layouts, as a
 * checkpoint holds them (float32, C-contiguous): weights_1 [9][9][1][128], weights_2/3 [5][5][128 in][128 out],
 * weights_4/5 [5][5][128 out][128 in], weights_6 [9][9][1][128], gamma_i [128][128], biases_i / beta_i [128].
 * gamma_3, beta_3, gamma_4, beta_4 exist only in the fixed-bin-width model (components.py:137-142, 53-58): NULL when
 * are_bin_widths_learned != 0. Either side may be left out entirely (all its pointers NULL): an encoder-only model for
 * eae_hip_encode, a decoder-only one (the reference's IsolatedDecoder, IsolatedDecoder.py:21-129) for eae_hip_decode; a side
 * that is given must be complete, else EAE_HIP_BAD_ARGUMENT.
 * eae_hip_model_create uploads and re-lays them out on the current device (7 MB resident) and returns when that is done.
 * eae_hip_encode:  images uint8 [n][h][w] (device, 4-byte aligned) -> latents f32 [n][h/16][w/16][128] (device). h, w
 *                  multiples of 16.
 * eae_hip_decode:  quantised latents f32 [n][h_latent][w_latent][128] (after the de-centring of
 *   reconstructing_eae_kodak.py:192) -> out_f32 [n][16 h_latent][16 w_latent] (nullable: the float reconstruction),
 *   out_u8 (nullable: its BT.601 cast), and with ref_u8 + sse the squared error per image as in eae_hip_tconv9x9s4_luma.
 * scratch: device memory of eae_hip_{encode,decode}_scratch_bytes(...) bytes for the activations between the layers;
 * contents need not survive between calls, calls that may overlap (different streams) need different blocks. Everything is
 * asynchronous on `stream`; results are bit-identical to the per-layer entry points and to oracle/transforms_oracle.c. */
typedef struct eae_hip_model eae_hip_model;
typedef struct eae_hip_variables {
    const float *weights_1, *biases_1, *gamma_1, *beta_1;
    const float *weights_2, *biases_2, *gamma_2, *beta_2;
    const float *weights_3, *biases_3, *gamma_3, *beta_3;
    const float *gamma_4, *beta_4;
    const float *weights_4, *biases_4, *gamma_5, *beta_5;
    const float *weights_5, *biases_5, *gamma_6, *beta_6;
    const float *weights_6;
} eae_hip_variables;
int eae_hip_model_create(const eae_hip_variables* host_variables, int are_bin_widths_learned, eae_hip_model** model);
void eae_hip_model_destroy(eae_hip_model* model);
int eae_hip_model_are_bin_widths_learned(const eae_hip_model* model);
uint64_t eae_hip_encode_scratch_bytes(int n, int h, int w);                 /* 0 for sizes eae_hip_encode rejects */
uint64_t eae_hip_decode_scratch_bytes(int n, int h_latent, int w_latent);
/* eae_hip_transform_status: the failure word of the MOST RECENT encode / decode call issued on `stream` with this scratch
 * block (the hand-off timeout of a cut launch, see eae_hip_conv_workspace_collect). Waits for `stream`, writes the count to
 * *host_count (0 = the results are valid) and returns 0, or a hipError_t. Not calling it forgoes the check; it never
 * affects later calls (encode / decode zero their workspace and the word on entry). */
int eae_hip_transform_status(void* scratch, uint32_t* host_count, void* stream);
int eae_hip_encode(const eae_hip_model* model, const uint8_t* images, int n, int h, int w, float* latents, void* scratch,
                   uint64_t scratch_bytes, void* stream);
int eae_hip_decode(const eae_hip_model* model, const float* quantized_latents, int n, int h_latent, int w_latent,
                   float* out_f32, uint8_t* out_u8, const uint8_t* ref_u8, uint64_t* sse, void* scratch,
                   uint64_t scratch_bytes, void* stream);

/* ---- analysis transform (eae/graph/components.py:86-142) ---------------------------------------------------------*/

/* conv_1 + bias_add + gdn_1  (components.py:119-125; tf.nn.conv2d 9x9, 1->128, stride 4, 'SAME' = pad 2/3;
 * tfutils.py:393-397). x: uint8 [N][H][W] (the uint8->float32 cast of batching.py:95 is done in-kernel, no offset,
 * no scale); w_packed: [82][128] from eae_hip_pack_conv9x9s4_weights; out: f32 [N][H/4][W/4][128]. H, W multiples
 * of 4, x aligned to 4 bytes (the kernel reads whole 32-bit words; EAE_HIP_BAD_ARGUMENT otherwise). gamma_packed: from
 * eae_hip_pack_gamma; NULL skips the normalisation (plain conv + bias). */
int eae_hip_conv9x9s4_u8(const uint8_t* x, const float* w_packed, const float* bias, const float* gamma_packed,
                         const float* beta, float* out, int n, int h, int w_in, void* stream);

/* conv_2 / conv_3 + bias_add (+ gdn_2 / gdn_3)  (components.py:126-142; tf.nn.conv2d 5x5, 128->128, stride 2,
 * 'SAME' = pad 1/2). x: f32 [N][H][W][128]; w_packed: from eae_hip_pack_conv_weights (HWIO [5][5][128][128] with the
 * output channels in packed order); out: [N][H/2][W/2][128]. H, W even, and H * W * 512 bytes (one image's input plane; the
 * kernels address inside an image with 32-bit byte offsets) below 2 GB: EAE_HIP_BAD_SHAPE otherwise. For the whole path that is
 * an image of at most 67 megapixels (conv_2's input is the 1/16-size plane: 8192 x 8176 passes, 8192 x 8192 does not).
 * norm: EAE_NORM_NONE (learned-bin-width model, components.py:137-138) or EAE_NORM_GDN (gamma_packed, beta). */
int eae_hip_conv5x5s2(const float* x, const float* w_packed, const float* bias, int norm, const float* gamma_packed,
                      const float* beta, float* out, int n, int h, int w_in, void* stream);

/* GDN / IGDN on its own (tfutils.py:363-397 / 480-509): out[r][c] = x[r][c] (/ or *) sqrt(beta[c] + sum_k x[r][k]^2
 * gamma[k][c]); x, out: [rows][128]; gamma_packed from eae_hip_pack_gamma. Used for inverse_gdn #4
 * (components.py:53-58) and as a standalone op. */
int eae_hip_gdn(const float* x, const float* gamma_packed, const float* beta, int inverse, float* out, int64_t rows,
                void* stream);

/* ---- synthesis transform (components.py:11-84) --------------------------------------------------------------------*/

/* transpose_conv_1 / _2 + bias_add + inverse_gdn of the next layer (components.py:63-78; tf.nn.conv2d_transpose 5x5,
 * 128->128, stride 2, 'SAME'). x: [N][h][w][128]; w_packed: from eae_hip_pack_tconv_weights (the TF filter
 * [5][5][out][in] as [25][in][packed out]); out: [N][2h][2w][128] (one image's output plane below 2 GB, as above). */
int eae_hip_tconv5x5s2(const float* x, const float* w_packed, const float* bias, int norm, const float* gamma_packed,
                       const float* beta, float* out, int n, int h, int w_in, void* stream);

/* The same two ops with a workspace that lets the launch cut its last tiles (csrc/hip/conv_gemm_split.hip): identical
 * results, no partly empty last round. A batch gives every SIMD a non-integer number of 32-position tiles (4.5 for conv_2
 * and 1.1 for conv_3 on 24 Kodak images); with the workspace the last tiles of the launch are interrupted at a K-step
 * boundary -- the accumulators parked in the tile's own output pixels -- and finished by a wave dispatched at the very end,
 * so that every SIMD runs dry at the same time. The per-element f32 FMA chain, hence every bit of the result, is the one
 * documented above. workspace: eae_hip_conv_workspace_bytes() bytes of device memory, ALL ZERO on entry and all zero again
 * when the launch has completed (zero it once); launches that may run concurrently need their own. Whether a launch is cut
 * is decided from its shape (only the convolutions, only when the last round would be less than ~97 % full). */
uint64_t eae_hip_conv_workspace_bytes(void);
/* Failure mode of a cut launch, and how it surfaces. The second half of a cut tile waits for the first half's accumulators;
 * the first half is always dispatched earlier (workgroups start in grid order on gfx950, the only device on which launches
 * are cut), so the wait is finite. Should it ever exceed ~1 s, the waiting wave writes NO result for its tile and counts
 * itself in the workspace's error word. eae_hip_conv_workspace_collect, enqueued behind the launch(es), adds that count to
 * *error_word (device memory or device-mapped pinned host memory; non-zero = the outputs of the launches since the last
 * collect are INVALID) and restores the all-zero workspace, so later launches are unaffected. A caller that cuts launches
 * must collect before trusting their outputs (codec.BatchCodec does per batch and raises from Ticket.result()). */
int eae_hip_conv_workspace_collect(void* workspace, uint32_t* error_word, void* stream);
int eae_hip_conv5x5s2_ws(const float* x, const float* w_packed, const float* bias, int norm, const float* gamma_packed,
                         const float* beta, float* out, int n, int h, int w_in, void* workspace, void* stream);
int eae_hip_tconv5x5s2_ws(const float* x, const float* w_packed, const float* bias, int norm, const float* gamma_packed,
                          const float* beta, float* out, int n, int h, int w_in, void* workspace, void* stream);

/* transpose_conv_3 (components.py:79-83; 9x9, 128->1, stride 4, 'SAME', no bias) fused with what follows it on the
 * path: tls.cast_bt601 (tools.py:93: uint8(round_half_even(clip(x,16,235)))) and the squared error of tls.psnr_2d
 * (tools.py:873-875). x: [N][h][w][128]; w_phase: 18432 floats from eae_hip_pack_tconv9x9s4_weights;
 * out_f32 (nullable): [N][4h][4w] float reconstruction; out_u8 (nullable): [N][4h][4w] BT.601 cast;
 * ref_u8 + sse (both nullable): sse[i] += sum over image i of (ref - out_u8)^2, exact uint64 (caller zeroes). */
int eae_hip_tconv9x9s4_luma(const float* x, const float* w_phase, float* out_f32, uint8_t* out_u8,
                            const uint8_t* ref_u8, uint64_t* sse, int n, int h, int w_in, void* stream);

/* TF filter [9][9][1][128] -> per-lane MFMA fragments [4 channel blocks][9 neighbours][64 lanes][8] (zeros where an
 * output phase has no tap). */
int eae_hip_pack_tconv9x9s4_weights(const float* w_tf, float* w_phase, void* stream);

/* Kernel-side layouts, packed once per model on the device. "Packed" channel order: out channel c sits at position
 * (c % 32) * 4 + c / 32 of its row, so that the 4 values one lane needs for its 4 column tiles are one 16-byte load.
 *   eae_hip_pack_conv_weights : HWIO [taps][128 in][128 out]            -> [taps][128 in][packed out]   (conv_2, conv_3)
 *   eae_hip_pack_tconv_weights: TF conv2d_transpose [taps][128 out][128 in] -> [taps][128 in][packed out] (tconv_1, _2)
 *   eae_hip_pack_gamma        : gamma [128 k][128 c]                     -> [128 k][packed c]            (every GDN / IGDN) */
int eae_hip_pack_conv_weights(const float* w_hwio, float* w_packed, int taps, void* stream);
/* conv_1 filter [9][9][1][128] -> [82 taps (81 + one zero row)][packed out] */
int eae_hip_pack_conv9x9s4_weights(const float* w_tf, float* w_packed, void* stream);
int eae_hip_pack_tconv_weights(const float* w_tf, float* w_packed, int taps, void* stream);
int eae_hip_pack_gamma(const float* gamma, float* gamma_packed, void* stream);

/* ---- quantiser, symbols, per-map statistics ------------------------------------------------------------------------
 * One pass over the latents y: [N][h*w][C] f32, C = 128. For every element, with m = map_mean[c] (NULL = 0) and
 * bw = bin_widths[c]:
 *   centered  = y - m                                   (reconstructing_eae_kodak.py:178)
 *   r         = round_half_even(centered / bw)          (tools.py:929)
 *   cq        = bw * r                                  (tools.py:929, `quantize_per_map`)
 *   symbol    = int16(round_half_even(cq / bw))         (lossless/compression.py:142, tools.py:126-133)
 *   shifted   = cq + m                                  (reconstructing_eae_kodak.py:192)
 * Outputs (each nullable):
 *   cq_out, shifted_out : f32 [N][h*w][C]
 *   symbols_planar      : int16 [N][C][h*w]  -- map-major: map (i, c) is `ref_int16[:, :, c].flatten()` of
 *                         compression.py:77 for image i; this is the buffer the single device->host copy moves
 *   nonzero_flags       : uint32 [N][C], set to 1 when some cq of the map is != 0 (so `count_nb_deads`,
 *                         tools.py:318-320, is the number of zero flags); caller zeroes
 *   checks              : uint32 [3], caller zeroes; each += a count of offending elements:
 *       [0] |round(cq/bw)| >= 32768                      -> AssertionError of tools.py:130-132
 *       [1] |bw*round(x/bw) - x| >= 1.5e-10, x = y - m   -> AssertionError "The quantization was omitted."
 *                                                           (tools.py:372-375) when y is passed as already quantised
 *       [2] float(symbol)*bw != x                        -> AssertionError of lossless/compression.py:149-153
 * c = number of maps (last axis). c == 128 takes the tiled kernel; any other c a generic one (same arithmetic). */
int eae_hip_quantize_maps(const float* y, const float* map_mean, const float* bin_widths,
                          float* cq_out, float* shifted_out, int16_t* symbols_planar,
                          uint32_t* nonzero_flags, uint32_t* checks, int n, int hw, int c, void* stream);

/* The whole latent stage of the fixed-bin-width model in one pass (csrc/hip/latent.hip): x = conv_3 + bias (NORM_NONE)
 *   -> gdn_3 (gamma_in_packed/beta_in; both NULL = no normalisation, the learned-bin-width model, components.py:137-138)
 *   -> the quantiser exactly as eae_hip_quantize_maps (symbols_planar, nonzero_flags, checks: same meaning, caller zeroes
 *      flags and checks; map_mean nullable)
 *   -> + map_mean -> inverse_gdn_4 (gamma_out_packed/beta_out; both NULL = none) into t_out, the input of transpose_conv_1.
 * y_out (the latents after gdn_3), shifted_out (quantised + mean) are optional f32 [N][hw][128] outputs.
 * Same arithmetic as eae_hip_gdn + eae_hip_quantize_maps + eae_hip_gdn: identical bits (tests/test_gpu_latent.py). */
int eae_hip_latent_stage(const float* x, const float* gamma_in_packed, const float* beta_in, const float* map_mean,
                         const float* bin_widths, const float* gamma_out_packed, const float* beta_out, float* y_out,
                         float* shifted_out, float* t_out, int16_t* symbols_planar, uint32_t* nonzero_flags, uint32_t* checks,
                         int n, int hw, void* stream);

/* conv_3 + bias_add with the latent stage behind it in one launch: the convolution's register tile goes straight through
 * gdn_3 -> quantiser -> inverse_gdn_4 (eae_hip_latent_stage's arithmetic, same bits) instead of through HBM and a second
 * kernel -- what `codec.BatchCodec` launches between conv_2 and transpose_conv_1. x: gdn_2 output [N][h][w][128]; w_packed /
 * bias: conv_3; gamma_in/beta_in (gdn_3) and gamma_out/beta_out (inverse_gdn_4): both pairs (fixed-bin-width model) or
 * neither (learned: the quantiser only); outputs as eae_hip_latent_stage ([N][h/2 * w/2][128] f32, planar int16 symbols,
 * flags and checks zeroed by the caller); t_out (fixed) / shifted_out (learned) is required: it is the synthesis transform's
 * input. workspace: as eae_hip_conv5x5s2_ws (nullable: then the launch is never cut). Layers too small for the fused kernel
 * run as the convolution followed by eae_hip_latent_stage in place. */
int eae_hip_conv5x5s2_latent(const float* x, const float* w_packed, const float* bias, const float* gamma_in_packed,
                             const float* beta_in, const float* map_mean, const float* bin_widths, const float* gamma_out_packed,
                             const float* beta_out, float* y_out, float* shifted_out, float* t_out, int16_t* symbols_planar,
                             uint32_t* nonzero_flags, uint32_t* checks, int n, int h, int w_in, void* workspace, void* stream);

/* The map means of lossless/stats.py:306, `numpy.mean(y_float32, axis=(0, 1, 2))`, bit for bit: means[c] = (the float32
 * sum of y[row][c] accumulated row by row, rows ascending) / float32(rows) -- the order numpy reduces the leading axes of a
 * C-contiguous float32 array in (tests/test_host_logic.py pins that order against numpy itself). y: [rows][c] f32. */
int eae_hip_map_means(const float* y, float* means, int64_t rows, int c, void* stream);

/* The two device passes of lossless/stats.py:197-241 (find_index_map_exception), whose per-map loop calls
 * compute_probabilities_intervals(map, 1.) (stats.py:70-134): numpy.amin / amax per map, then a histogram over the
 * unit-width intervals [floor(min), ceil(max)].
 *   map_minmax: minmax[0][c] = min, minmax[1][c] = max over rows of y[row][c]; scratch_keys: 2*c uint32 of scratch.
 *   floor_histograms: hist[c][floor(y) + radius] += 1 for |floor(y)| <= radius, else overflow[c] += 1 (caller zeroes both;
 *   hist is [c][2*radius+1]). The closed last interval of numpy.histogram (values equal to the right edge) is folded in
 *   by the host, which also forms the probabilities and the Jensen-Shannon divergences in float64 like the reference. */
int eae_hip_map_minmax(const float* y, float* minmax, uint32_t* scratch_keys, int64_t rows, int c, void* stream);
int eae_hip_floor_histograms(const float* y, uint32_t* hist, int radius, uint32_t* overflow, int64_t rows, int c, void* stream);

/* tls.count_nb_deads (tools.py:294-320) for an arbitrary stack x [N][hw][C]: nonzero_flags[n][c] = 1 when some
 * element of map (n, c) is != 0 (caller zeroes); the number of dead maps of image n is the number of zero flags. */
int eae_hip_nonzero_flags(const float* x, uint32_t* nonzero_flags, int n, int hw, int c, void* stream);

/* tls.cast_float_to_int16 (tools.py:95-133): out = int16(round_half_even(x)); *range_error += number of elements
 * with |round(x)| >= 32768 (the reference raises AssertionError); caller zeroes. */
int eae_hip_cast_int16(const float* x, int16_t* out, int64_t count, uint32_t* range_error, void* stream);

/* Per-map symbol histograms (tls.count_symbols, tools.py:376-388, is this histogram restricted to [min, max]; it feeds
 * discrete_entropy :523-537, rate_3d :977-989 and stats.count_binary_decisions, lossless/stats.py:179-195).
 * symbols_planar: int16 [n_maps][map_size]; hist: uint32 [n_maps][2*hist_radius+1], bin = symbol + hist_radius;
 * symbols outside the radius are counted in overflow[n_maps] instead (re-run with a larger radius; 32768 covers
 * int16). hist and overflow are ACCUMULATED into: the caller zeroes them. */
int eae_hip_symbol_histograms(const int16_t* symbols_planar, uint32_t* hist, int hist_radius, uint32_t* overflow,
                              int n_maps, int map_size, void* stream);
/* Same over every map_step-th map starting at first_map (row i of hist = map first_map + i * map_step): the exception
 * map of every image of a batch (first_map = idx_map_exception, map_step = 128) without gathering it first. */
int eae_hip_symbol_histograms_strided(const int16_t* symbols_planar, uint32_t* hist, int hist_radius, uint32_t* overflow,
                                      int n_maps, int map_size, int64_t first_map, int64_t map_step, void* stream);

/* ---- lossless coder on the device: one feature map per lane --------------------------------------------------------
 * Replaces the per-map compress_lossless loop of lossless/compression.py:76-81 (and, underneath it,
 * lossless/c++/source/compression.cpp:3-65) for a whole batch of images whose symbols are already in HBM
 * (symbols_planar of eae_hip_quantize_maps). Same arguments and stream layout as the HOST entry points
 * eae_coder_compress_maps / eae_coder_encode_maps / eae_coder_decode_maps of include/eae_coder.h, every pointer a
 * DEVICE pointer; same source for the arithmetic (csrc/coder/coder_core.h), so bits, bytes, bit counts and error codes
 * are identical to the host library's and therefore to the reference's.
 *   symbols [n_maps][map_size] int16; probs [rows][L] float64; prob_row[m] = row for map m, < 0 skips the map
 *   (exception map: 0 bits, verbatim copy when `reconstruction` is given), NULL = row m;
 *   streams: n_maps regions of `stride` bytes, BAC bytes at +0, bypass bytes at +stride/2 (stride from
 *   eae_hip_coder_stream_stride_bytes; 8-byte aligned base); bac_bits/bypass_bits/status/stage: per map.
 *   mode: 0 encode + decode into `reconstruction`; 1 encode only; 2 encode + decode + compare in registers
 *   (status 6 = EAE_ROUNDTRIP_MISMATCH). lanes_per_wave: 1..64 = that many maps per 64-thread block, one per lane
 *   (fewer lanes = more blocks over more CUs and less divergence per wave, but more wave-instructions in total);
 *   <= 0 = one wavefront per map with wave-uniform state (scalar registers, ~20 VGPRs: shortest latency per map).
 * Returns 0, -1 (NULL argument / bad mode), 1 (stride too small: EAE_CAPACITY_ERROR) or a hipError_t. Per-map failures
 * are reported in status[] (the launch is asynchronous). */
uint64_t eae_hip_coder_stream_stride_bytes(uint32_t map_size, uint8_t truncated_unary_length);
int eae_hip_coder_compress_maps(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, int16_t* reconstruction,
                                uint8_t truncated_unary_length, const double* probabilities, const int32_t* prob_row,
                                uint8_t* streams, uint64_t stream_stride_bytes, uint32_t* bac_bits, uint32_t* bypass_bits,
                                int32_t* status, int32_t* stage, int mode, int lanes_per_wave, void* stream);
int eae_hip_coder_decode_maps(uint32_t n_maps, uint32_t map_size, int16_t* symbols_out, uint8_t truncated_unary_length,
                              const double* probabilities, const int32_t* prob_row, const uint8_t* streams,
                              uint64_t stream_stride_bytes, const uint32_t* bac_bits, const uint32_t* bypass_bits,
                              int32_t* status, int32_t* stage, int lanes_per_wave, void* stream);
/* The second half of compress_lossless as its own launch (so that it can run on another stream, concurrently with the
 * next batch's encode): decodes every map from its streams and compares with `expected` (the symbols that were
 * encoded) in registers. status[m] = 6 (EAE_ROUNDTRIP_MISMATCH) or a decoder error; maps whose status is already
 * non-zero (failed encode) and skipped maps are left alone. */
int eae_hip_coder_verify_maps(uint32_t n_maps, uint32_t map_size, const int16_t* expected, uint8_t truncated_unary_length,
                              const double* probabilities, const int32_t* prob_row, const uint8_t* streams,
                              uint64_t stream_stride_bytes, const uint32_t* bac_bits, const uint32_t* bypass_bits,
                              int32_t* status, int32_t* stage, int lanes_per_wave, void* stream);

/* ---- the same coder, 64 maps per wavefront in step (csrc/hip/coder_simd.hip) -----------------------------------------
 * Same arguments, stream layout and results as the entry points above, organised for the machine. Only the interval
 * arithmetic of the binary arithmetic coder is serial (one map per lane, csrc/coder/lean_step.h); everything around it is
 * data-parallel over the symbols:
 *   encode_batch: binarise (decisions, the bypass stream of signs / Exp-Golomb suffixes) -> the serial core, which leaves
 *     one 32-bit record per decision (the bits that may leave, E1/E2 and E3 counts) -> emit (prefix sums over the records
 *     place the bits and the pending-E3 runs in the stream);
 *   decode_batch: the serial core reads the stream through a small ring in LDS and leaves one byte per symbol (its
 *     truncated-unary prefix) -> debinarise (signs, suffixes from the bypass stream) -> compare with `expected`.
 * Every map the fast kernels cannot finish (any error, streams that outgrow their region, exotic lengths) is recoded by
 * the general kernel above, so statuses, stages, bit counts and bytes are identical in every case. These are the launches
 * bench.py times.
 *   streams / stream_stride_bytes: as above, with every map's two streams on 16-byte boundaries and 16 spare bytes per
 *   stream (the stride of eae_hip_coder_stream_stride_bytes on a 16-byte aligned base does that): encode_batch returns
 *   1 (hipErrorInvalidValue) for any other layout, decode_batch hands such streams to the general kernel, map by map.
 *   L == 0 and L > 32 go to the general kernel too. Same results in every case.
 *   workspace: device scratch of eae_hip_coder_workspace_bytes(n_maps, map_size, L) bytes, private to the call chain
 *   (encode_batch followed by decode_batch of the same maps may share it; concurrent batches need their own). It holds the
 *   decisions and records of the batch: about 4 * (L + 1) + 12 bytes per symbol for the records alone.
 *   encode_batch: symbols -> streams + bac_bits/bypass_bits/status/stage (all written for every map).
 *   decode_batch: streams -> symbols_out (expected == NULL; status/stage written for every map; skipped maps are left
 *   untouched), or, with expected != NULL, decode into the workspace (symbols_out may be NULL) and compare: maps whose
 *   status is already non-zero are left alone, a difference gives status 6 (EAE_ROUNDTRIP_MISMATCH). */
uint64_t eae_hip_coder_workspace_bytes(uint32_t n_maps, uint32_t map_size, uint8_t truncated_unary_length);
int eae_hip_coder_encode_batch(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, uint8_t truncated_unary_length,
                               const double* probabilities, const int32_t* prob_row, uint8_t* streams,
                               uint64_t stream_stride_bytes, uint32_t* bac_bits, uint32_t* bypass_bits, int32_t* status,
                               int32_t* stage, void* workspace, uint64_t workspace_bytes, void* stream);
int eae_hip_coder_decode_batch(uint32_t n_maps, uint32_t map_size, int16_t* symbols_out, const int16_t* expected,
                               uint8_t truncated_unary_length, const double* probabilities, const int32_t* prob_row,
                               const uint8_t* streams, uint64_t stream_stride_bytes, const uint32_t* bac_bits,
                               const uint32_t* bypass_bits, int32_t* status, int32_t* stage, void* workspace,
                               uint64_t workspace_bytes, void* stream);

/* ==== EXPERIMENTAL (-DEAE_EXPERIMENTAL_CODER): NOT exported by the product library lib/libeae_hip.so ==========================
 * Two byte-exact alternatives for the coder's round trip on small batches, both measured slower than the encode_batch +
 * decode_batch pair on this runtime. They are compiled into lib/libeae_hip_test.so only
 * (csrc/Makefile: `make hip-test`), where their parity tests run (tests/test_coder_device.py). */
#ifdef EAE_EXPERIMENTAL_CODER
/* The round trip of lossless/c++/source/compression.cpp:27-64 (encode every map, decode it back, compare) for SMALL batches -- one or
 * two images, two to four wavefronts of maps -- where what it costs is the length of its serial chains one after the other. The
 * chains are cut into `chunks` launches (2..16; 4 is a good value): while the encoder core runs chunk c + 1 the emit pass assembles
 * the stream words of chunk c and the decoder works through those of chunk c - 1, on two side streams of the library's own joined
 * to `stream` by events (nothing is polled: capturable into a hipGraph). Results as eae_hip_coder_encode_batch followed by
 * eae_hip_coder_decode_batch(expected = symbols): same stream bytes, bit counts, statuses (6 on a difference) and stages, same
 * fall-back to the general kernel per map. workspace: eae_hip_coder_trailing_workspace_bytes. chunks < 2: the two calls one after
 * the other. For large batches the transforms of the batches in flight hide the coder anyway and the extra launches only cost. */
uint64_t eae_hip_coder_trailing_workspace_bytes(uint32_t n_maps, uint32_t map_size, uint8_t truncated_unary_length);
int eae_hip_coder_roundtrip_trailing(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, uint8_t truncated_unary_length,
                                     const double* probabilities, const int32_t* prob_row, uint8_t* streams,
                                     uint64_t stream_stride_bytes, uint32_t* bac_bits, uint32_t* bypass_bits, int32_t* status,
                                     int32_t* stage, void* workspace, uint64_t workspace_bytes, uint32_t chunks, void* stream);

/* The same round trip with the three serial stages of every group of 64 maps in ONE workgroup: the encoder core, a bit writer
 * (the emit pass, one map per lane) and the decoder core are three wavefronts that hand records and stream words to each other
 * through rings in LDS, so the decoder runs a few hundred bits behind the encoder instead of after it, and the records never go
 * to memory. One stream, five launches (binarise, the pipeline, the general kernel for what the pipeline hands over, debinarise,
 * compare). Same stream bytes, bit counts, statuses and stages as encode_batch + decode_batch(expected = symbols).
 * workspace: eae_hip_coder_trailing_workspace_bytes. */
int eae_hip_coder_roundtrip_fused(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, uint8_t truncated_unary_length,
                                  const double* probabilities, const int32_t* prob_row, uint8_t* streams,
                                  uint64_t stream_stride_bytes, uint32_t* bac_bits, uint32_t* bypass_bits, int32_t* status,
                                  int32_t* stage, void* workspace, uint64_t workspace_bytes, void* stream);

#endif /* EAE_EXPERIMENTAL_CODER */

/* ---- container support (the reference never serialises, compression.cpp:27-64) ------------------
 * pack_streams: gathers the valid bytes of every map's two streams (layout above) into `payload`: the arithmetic-coded
 * bytes of map m at offsets[2m], its bypass bytes at offsets[2m+1] (uint64 byte offsets, device memory, chosen by the
 * caller from the bit counts); unpack_streams is the inverse, into a stream region a decoder can read.
 * dequantize_maps: the inverse of the symbol conversion of lossless/compression.py:142 followed by the de-centring of
 * reconstructing_eae_kodak.py:192: symbols_planar [N][128][hw] int16 -> cq_out = bin_widths[c] * symbol (the exact value
 * tools.py:929 produced) and shifted_out = cq + map_mean[c], both f32 [N][hw][128], each nullable. */
int eae_hip_coder_pack_streams(uint32_t n_maps, const uint8_t* streams, uint64_t stream_stride_bytes, const uint32_t* bac_bits,
                               const uint32_t* bypass_bits, const uint64_t* offsets, uint8_t* payload, void* stream);
int eae_hip_coder_unpack_streams(uint32_t n_maps, const uint8_t* payload, const uint64_t* offsets, const uint32_t* bac_bits,
                                 const uint32_t* bypass_bits, uint8_t* streams, uint64_t stream_stride_bytes, void* stream);
int eae_hip_dequantize_maps(const int16_t* symbols_planar, const float* bin_widths, const float* map_mean, float* cq_out,
                            float* shifted_out, int n, int hw, int c, void* stream);

/* ==== TEST HOOKS (-DEAE_TEST_HOOKS): NOT exported by the product library lib/libeae_hip.so ====================================
 * Four entry points the test-suite needs and a deployment must not have (they change what later launches do, or only
 * exist to prove something about the kernels). Compiled into lib/libeae_hip_test.so only; tests/test_abi.py checks that
 * the product library exports no symbol of this section (nor any other `debug` symbol). */
#ifdef EAE_TEST_HOOKS
/* Diagnostic hook (not part of the path): when given a device buffer of grid * waves * 8 uint64, the conv GEMM kernel
 * records s_memtime stamps per wave (start, loop start, loop end, GDN end, end, K-steps, XCC id, HW id). NULL disables. */
int eae_hip_debug_set_stamp_buffer(uint64_t* device_buffer);

/* Kernel-form overrides (EAE_HIP_GEMM, EAE_HIP_SPLIT_WAVES, EAE_HIP_FORCE_TILE, EAE_HIP_FORCE_NT, EAE_HIP_LATENT: README.md) are
 * read from the environment ONCE, when the library is loaded -- no launch reads the environment, and a hipGraph captures what
 * every later launch would have done anyway. The parity tests of every kernel form change the environment inside one process and
 * then call this to have it read again. Not part of the path. */
int eae_hip_debug_reload_launch_options(void);
/* Fault injection for the cut-tile hand-off of the conv launches (tests/test_gpu_conv_split.py, tests/test_gpu_codec.py): with
 * on != 0 the heads of cut tiles never publish and the tails give up after ~1 ms, so the launch reports the hand-off failure
 * (eae_hip_conv_workspace_collect, eae_hip_transform_status). Reachable through this entry point only -- no environment variable
 * can make a deployment drop tiles. */
int eae_hip_debug_set_split_mute(int on);
/* Proof harness of the normalisations' mid-range forms (csrc/hip/common.h: sqrt_mid, div_mid -- hipcc's correctly rounded sqrtf and
 * `/` without the scaling and fix-up steps that only extreme operands need). mode 0: every float with bits in [first, first + count)
 * through sqrt_mid and sqrtf; mode 1: `count` pseudo-random operand pairs of the guarded range (all exponents, random and extreme
 * mantissas) through div_mid and `/`. out2_device[0] += results whose bits differ, out2_device[1] = one such operand (pair). */
int eae_hip_debug_check_mid_forms(int mode, uint64_t first, uint64_t count, uint64_t seed, uint64_t* out2_device, void* stream);
#endif /* EAE_TEST_HOOKS */

/* tls.cast_bt601 (tools.py:61-93) on its own: u8 = uint8(round_half_even(clip(x, 16, 235))). */
int eae_hip_cast_bt601(const float* x, uint8_t* out, int64_t count, void* stream);

/* tls.rgb_to_ycbcr (tools.py:1019-1083; how the dataset builders turn RGB photographs into the luminance images of the
 * path, datasets/kodak/kodak.py:70): rgb uint8 [count][3] -> ycbcr uint8 [count][3] and / or luma uint8 [count] (each
 * nullable), ITU-R BT.601 in float64, round half to even. Exact for all 2^24 inputs (tests/test_gpu_kernels.py). */
int eae_hip_rgb_to_ycbcr(const uint8_t* rgb, uint8_t* ycbcr, uint8_t* luma, int64_t count, void* stream);

/* Squared error per image for tls.psnr_2d (tools.py:873-875): sse[i] += sum (a - b)^2 over pixels_per_image. */
int eae_hip_sse_u8(const uint8_t* a, const uint8_t* b, uint64_t* sse, int n, int64_t pixels_per_image, void* stream);

/* ---- SVHN path, BASELINE.json configs[0] (svhn/eae/EntropyAutoencoder.py, svhn/eae/utils.py) ------------------------
 * The reference's pure-numpy FLOAT64 fully connected autoencoder 3072 -> 300 -> 200 -> 300 -> 3072, LeakyReLU(0.1).
 * All tensors row-major float64 on the device. Every dot product is one float64 FMA chain, k ascending from +0, bias
 * added afterwards (oracle/svhn_oracle.c runs the same chain; numpy.dot's BLAS order is unspecified). */

/* out[n][m] = act(x[n][:] . w[:][m] + b[m]); act = LeakyReLU(0.1) when leaky_relu != 0 (svhn/tools/tools.py:676-694).
 * One call per layer of `encoder` (EntropyAutoencoder.py:239-246) / `decoder` (:270-277). x: [n][k], w: [k][m]. */
int eae_hip_svhn_dense_f64(const double* x, const double* w, const double* b, double* out, int n, int k, int m,
                           int leaky_relu, void* stream);
/* preprocess_svhn (svhn/svhn/svhn.py:210): (uint8 - mean[j]) / std. images: [n][d] uint8, mean: [d]. */
int eae_hip_svhn_preprocess(const uint8_t* images, const double* mean, double std_training, double* out, int n, int d,
                            void* stream);
/* tls.quantization (svhn/tools/tools.py:1095): q = bw * round_half_even(y / bw) with ONE scalar bin width, plus the
 * int32 symbols round(q / bw) behind tls.count_symbols / discrete_entropy (:214-231, :289-). q and symbols nullable.
 * checks[2] (caller zeroes): [0] symbols outside int32, [1] |q - y| >= 1.5e-10 ("The quantization was omitted.",
 * tools.py:214-217, when y is passed as already quantised). */
int eae_hip_svhn_quantize_f64(const double* y, double bin_width, double* q, int32_t* symbols, uint32_t* checks,
                              int64_t count, void* stream);
/* minmax[0] = min(minmax[0], min symbols), minmax[1] = max(...): caller initialises to (INT32_MAX, INT32_MIN). */
int eae_hip_svhn_symbol_range(const int32_t* symbols, int64_t count, int32_t* minmax, void* stream);
/* hist[s - lowest] += 1 for lowest <= s < lowest + nb_bins, *overflow += 1 otherwise (caller zeroes both). */
int eae_hip_svhn_symbol_histogram(const int32_t* symbols, int64_t count, int32_t lowest, int32_t nb_bins, uint32_t* hist,
                                  uint32_t* overflow, void* stream);
/* utils.py:71-74: uint8(round_half_even(clip(rec*std + mean[j], 0, 255))) (tools.py:166) and, with ref_u8 + sse, the
 * squared error per image behind tls.mean_psnr (tools.py:857-859). sse[i] is overwritten. */
int eae_hip_svhn_postprocess(const double* reconstruction, double std_training, const double* mean, uint8_t* out_u8,
                             const uint8_t* ref_u8, uint64_t* sse, int n, int d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EAE_HIP_H */
