/* bert_hip_test.h — op-level test hooks of the MI355X engine: standalone kernel entry points (host buffers in, host
 * buffers out) that the parity tests call through ctypes.  They live in libbert_test.so (libbert.so plus these), NOT in
 * the product library.
 */
#ifndef BERT_HIP_TEST_H
#define BERT_HIP_TEST_H

#include "bert_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Standalone kernel entry points for op-level tests (host buffers in, host buffers out).
 * C[M][N] = epilogue(A[M][K] (f16 bits) x W[N][K]^T + bias); W given in file layout of `wtype`
 * (row-major f32 / f16 / block_q4_0 / block_q4_1 bytes).  epilogue: 0 bias, 1 bias+GELU(tanh),
 * 2 bias+residual.  impl: 0 tiled MFMA kernel (gemm.hip; q4 blocks dequantised in the tile load), 1 naive, 3 the 256 x 256 tile
 * kernel (gemm256.hip; -2 unless f16/f32 weights and N % 256 == 0).  Output f16 bits.  Returns 0 on success.                  */
BERT_API int32_t bert_hip_test_gemm(int32_t M, int32_t N, int32_t K, const uint16_t *A, const void *W,
                                    int32_t wtype, const float *bias, const uint16_t *resid,
                                    int32_t epilogue, int32_t impl, uint16_t *C);

/* LayerNorm folded into the mat-muls around it (kernels.h GemmLnFold; gemm256.hip, f16 weights, H and N2 multiples of 256, M is
 * padded to 256): the pair a layer runs —
 *   u   = A1[M][K1] W1[H][K1]^T + b1 + R,   R = r[M][H] itself (rg == NULL) or LayerNorm(r; rg, rb) rebuilt per element from r's row
 *         statistics (computed here on the host), written UN-normalised with per-row partial statistics;
 *   out = epi2( LayerNorm(u; g, be) W2[N2][H]^T + b2 ),  epi2 0 = bias, 1 = bias + GELU: reads u itself, gamma folded into W2, one
 *         statistics k-step, rows scaled by 1 / std.
 * u_out [M][H], out2 [M][N2] f16 bits, rows_out [M][4] f32 {rstd, -mean rstd, -mean, std} of u.  Returns 0 on success.          */
BERT_API int32_t bert_hip_test_gemm_lnfold(int32_t M, int32_t K1, int32_t H, int32_t N2, const uint16_t *A1, const uint16_t *W1,
                                           const float *b1, const uint16_t *r, const float *rg, const float *rb,
                                           const uint16_t *W2, const float *b2, const float *g, const float *be, int32_t epi2,
                                           uint16_t *u_out, uint16_t *out2, float *rows_out);

/* qkv[T][3H] f16 bits (Q | K | V per row), packed sentences -> ctx[T][H] f16 bits.             */
BERT_API int32_t bert_hip_test_attention(int32_t n_sentences, const int32_t *cu_seqlens, int32_t n_head,
                                         int32_t d_head, const uint16_t *qkv, int32_t impl, uint16_t *out);

/* Q|K|V projection + attention: x[T][H] f16 bits, Wqkv [3H][H] (Q rows, K rows, V rows) in file layout of `wtype`,
 * bias[3H] -> ctx[T][H] f16 bits (reference bert.cpp:822-856).  fused: 0 = GEMM kernel + attention kernel; the window kernel
 * (qkv_attention2.hip; -2 if the shape is not supported) with 2 = next-fit windows built on the host, 3 = the uniform
 * placement rule, 4 = next-fit windows built on the device.                                                               */
BERT_API int32_t bert_hip_test_qkv_attention(int32_t n_sentences, const int32_t *cu_seqlens, int32_t n_head,
                                             int32_t d_head, const uint16_t *x, const void *Wqkv, int32_t wtype,
                                             const float *bias, int32_t fused, uint16_t *out);

/* Everything of a layer after the attention (reference bert.cpp:859-901):
 *   y = LayerNorm(ctx Wo^T + bo + x) * g1 + be1;  out = LayerNorm(gelu(y W1^T + b1) W2^T + b2 + y) * g2 + be2
 * ctx, x, out [M][H] f16 bits; Wo [H][H], W1 [I][H], W2 [H][I] in file layout of `wtype`.
 * impl: 0 = GEMM + LayerNorm kernels, 1 = the one-launch kernel with specialist wave pairs (layer_tail.hip; -2 if the shape is
 * not supported).  q4 `wtype`: both keep the blocks 4-bit on the device and dequantise in the tile load.             */
BERT_API int32_t bert_hip_test_layer_tail(int32_t M, int32_t H, int32_t I, const uint16_t *ctx, const uint16_t *x,
                                          const void *Wo, const void *W1, const void *W2, int32_t wtype,
                                          const float *bo, const float *g1, const float *be1, const float *b1,
                                          const float *b2, const float *g2, const float *be2, int32_t impl,
                                          uint16_t *out);

/* Embedding gather-sum + LayerNorm (reference bert.cpp:796-814): tables in the file layout of `table_type` (0 f32, 1 f16,
 * 2 q4_0, 3 q4_1), word [n_vocab][H], type [2][H], pos [n_pos][H]; packed sentences; out [T][H] f16 bits.               */
BERT_API int32_t bert_hip_test_embed_ln(int32_t table_type, int32_t H, int32_t n_vocab, int32_t n_pos, const void *word,
                                        const void *type, const void *pos, const float *gamma, const float *beta,
                                        const bert_vocab_id *tokens, const int32_t *cu_seqlens, int32_t n_sentences,
                                        uint16_t *out);
/* Mean-pool + L2 normalise (reference bert.cpp:904-913) of x [T][H] f16 bits -> out [n_sentences][H] f32; *status receives
 * the device status word (1 if a sentence length is outside [1, max_len]: its row is NaN).                              */
BERT_API int32_t bert_hip_test_pool_normalize(int32_t H, const uint16_t *x, const int32_t *cu_seqlens, int32_t n_sentences,
                                              int32_t max_len, float *out, int32_t *status);

/* Parses a model file (no GPU): returns the number of tensors (negative on error, message on stderr), whether the file uses the
 * legacy 20 / 24-byte q4 blocks, and a digest of every tensor's name, type and bytes AFTER conversion to the current layout.   */
BERT_API int32_t bert_hip_test_model_digest(const char *fname, int32_t *legacy_q4, uint64_t *digest);

/* Host logic of the multi-GPU layer and of the sentence windows, callable without a GPU:
 * shard bounds [n_shards + 1] of a packed batch (multi_device.h), and the {first, count} windows of 128 token slots
 * (engine.h build_windows; returns their number, `windows` holds 2 ints per window, capacity n_sentences).        */
BERT_API void bert_hip_test_shard_bounds(const int32_t *cu_seqlens, int32_t n_sentences, int32_t n_shards, int32_t *bounds);
BERT_API int32_t bert_hip_test_build_windows(const int32_t *cu_seqlens, int32_t n_sentences, int32_t *windows);
/* Upper bound of the number of windows used to size the grid of the fused attention kernel when the windows are built on the
 * device (a function of the sentence and token counts only).                                                             */
BERT_API int32_t bert_hip_test_max_windows(int32_t n_sentences, int32_t n_tokens);
/* the windows' place granularity in THIS library (16, or 8: BERT_HIP_WINDOW_SLOTS / option "window_slots"); returns the value now in force */
BERT_API int32_t bert_hip_test_set_window_slots(int32_t slots);
/* The same windows from the device-side builder the asynchronous device API uses (needs a GPU; -1 on a HIP error).        */
BERT_API int32_t bert_hip_test_build_windows_device(const int32_t *cu_seqlens, int32_t n_sentences, int32_t *windows);
/* The multi-device dispatcher (shards, a persistent worker thread per shard beyond the first, results straight into the
 * caller's rows) driven with a stub evaluator instead of GPUs: row b of `out` [n_sentences][H] becomes f(sentence b) =
 * {sum of ids, length, shard, ...}.  H < 0: the stub throws inside every shard (the exception must come back as -9).      */
BERT_API int32_t bert_hip_test_dispatch(const bert_vocab_id *tokens, const int32_t *cu_seqlens, int32_t n_sentences,
                                        int32_t n_shards, int32_t H, float *out);
/* Threads this process has created for shard work so far (ShardWorkers): repeated calls must not create threads.          */
BERT_API int64_t bert_hip_test_shard_threads_created(void);

#ifdef __cplusplus
}
#endif

#endif /* BERT_HIP_TEST_H */
