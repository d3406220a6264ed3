/* bert_hip.h — extensions of the MI355X engine beyond the reference's bert.h.
 *
 * Plain C ABI (pointers + sizes, no torch / HIP types in the signatures; a stream is passed as a
 * `void *` that must be a hipStream_t or NULL for the default stream).  None of these exist in
 * skeskinen/bert.cpp; each entry names the reference code it generalises.
 */
#ifndef BERT_HIP_H
#define BERT_HIP_H

#include "bert.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tokenizer-only context: header + vocab of a model file, no weights, no GPU needed.
 * bert_tokenize / bert_vocab_id_to_token / bert_n_max_tokens work; eval prints an error.
 * (Tokenization is CPU-only in the reference as well: bert.cpp:199-325.)                        */
BERT_API struct bert_ctx *bert_hip_load_tokenizer(const char *fname);

/* bert_tokenize for many texts on up to n_threads host threads (what bert_encode_batch does before it evaluates):
 * tokens[i * bert_n_max_tokens(ctx) ..] receives the ids of texts[i], n_tokens[i] their count.  Works on
 * tokenizer-only contexts.  Returns 0, negative on bad arguments.                                             */
BERT_API int32_t bert_hip_tokenize_batch(struct bert_ctx *ctx, int32_t n_threads, int32_t n_inputs, const char **texts,
                                         bert_vocab_id *tokens, int32_t *n_tokens);

/* Model facts from the file header (reference bert.cpp:361-367).                                */
BERT_API int32_t bert_hip_n_layer(struct bert_ctx *ctx);
BERT_API int32_t bert_hip_n_head(struct bert_ctx *ctx);
BERT_API int32_t bert_hip_n_intermediate(struct bert_ctx *ctx);
BERT_API int32_t bert_hip_n_vocab(struct bert_ctx *ctx);
BERT_API int32_t bert_hip_ftype(struct bert_ctx *ctx);        /* 0 f32, 1 f16, 2 q4_0, 3 q4_1 */
BERT_API int32_t bert_hip_device(struct bert_ctx *ctx);       /* HIP device ordinal, -1 if none */

/* Packed, variable-length batch evaluation — the engine's native entry point; bert_eval_batch
 * (reference bert.cpp:730-941) is a wrapper that packs the per-sentence host pointers.
 *   tokens      [n_tokens_total] ids of all sentences back to back
 *   cu_seqlens  [n_sentences + 1] exclusive prefix sums of the sentence lengths (cu[0] = 0)
 *   embeddings  [n_sentences * bert_n_embd] row-major
 * Returns 0 on success, negative on error (message on stderr, outputs untouched).               */
BERT_API int32_t bert_hip_eval_packed(struct bert_ctx *ctx, const bert_vocab_id *tokens,
                                      const int32_t *cu_seqlens, int32_t n_sentences, float *embeddings);

/* Same computation with every buffer already resident in HBM on the context's device; work is
 * enqueued on `stream` and NOT synchronised (the caller owns the stream).  `max_len` must be >=
 * the longest sentence (it selects the attention kernel variant); n_tokens_total = cu[n].       */
BERT_API int32_t bert_hip_eval_packed_device(struct bert_ctx *ctx, const bert_vocab_id *d_tokens,
                                             const int32_t *d_cu_seqlens, int32_t n_sentences,
                                             int32_t n_tokens_total, int32_t max_len,
                                             float *d_embeddings, void *stream);

/* Hidden-state tap for parity tests: one sentence, writes hidden[(n_layer+1)][n_tokens][n_embd]
 * f32 (after the embedding LayerNorm and after every encoder layer, reference bert.cpp:806-901)
 * and the final embedding.  Either output may be NULL.                                          */
BERT_API int32_t bert_hip_eval_hidden(struct bert_ctx *ctx, const bert_vocab_id *tokens, int32_t n_tokens,
                                      float *hidden, float *embedding);

/* Per-kernel timing with HIP events on the launch stream.  While enabled every kernel launch of
 * the forward pass is bracketed by events (this serialises nothing but adds event overhead, so
 * never enable it inside a throughput measurement).  bert_hip_profile_report writes one line per
 * kernel: "<name> <launches> <total_ms> <flops_per_launch_avg>\n" and returns the number of bytes
 * it needed (excluding NUL); it synchronises the device first and resets the counters.          */
BERT_API void    bert_hip_profile_enable(struct bert_ctx *ctx, int32_t on);
BERT_API int32_t bert_hip_profile_report(struct bert_ctx *ctx, char *buf, int32_t buf_len);

/* Engine knobs (also settable through the environment before bert_load_from_file):
 *   BERT_HIP_DEVICE        device ordinal (default: current device)
 *   BERT_HIP_CHUNK_TOKENS  max tokens evaluated per device pass by the host API (default 262144)
 *   BERT_HIP_GEMM          "mfma" (default) | "naive"  — kernel family for the weight mat-muls
 *   BERT_HIP_ATTN          "mfma" (default) | "naive"
 *   BERT_HIP_Q4            "expand" (default) | "fused" — q4_0 / q4_1 weight matrices are expanded to f16 images in HBM once
 *                          at load (same values, fastest kernels) or stay 4-bit and are dequantised inside the GEMM kernels
 *   BERT_HIP_TAIL          1 (default) | 0 — token-owning-waves kernel for out-projection + LN + FFN + LN (f16 weights)
 *   BERT_HIP_QKV_ATT       1 (default) | 0 — fused projection + attention kernel for batches of long sentences
 *   BERT_HIP_QUIET         1 = no progress text on stdout during load, no "unknown token" lines on stderr from bert_tokenize                            */
BERT_API void bert_hip_set_option(struct bert_ctx *ctx, const char *key, const char *value);

/* Standalone kernel entry points for op-level tests (host buffers in, host buffers out).
 * C[M][N] = epilogue(A[M][K] (f16 bits) x W[N][K]^T + bias); W given in file layout of `wtype`
 * (row-major f32 / f16 / block_q4_0 / block_q4_1 bytes).  epilogue: 0 bias, 1 bias+GELU(tanh),
 * 2 bias+residual.  impl: 0 tiled MFMA kernel, 1 naive, 2 row-panel kernel (epilogue 0 only; -2 if the
 * shape is not supported).  Output f16 bits.  Returns 0 on success.                             */
BERT_API int32_t bert_hip_test_gemm(int32_t M, int32_t N, int32_t K, const uint16_t *A, const void *W,
                                    int32_t wtype, const float *bias, const uint16_t *resid,
                                    int32_t epilogue, int32_t impl, uint16_t *C);

/* out = LayerNorm(A W^T + bias + resid) * gamma + beta (reference bert.cpp:859-875).  fused: 1 = single
 * row-panel kernel (-2 if unsupported), 0 = GEMM + LayerNorm kernels.                            */
BERT_API int32_t bert_hip_test_proj_ln(int32_t M, int32_t N, int32_t K, const uint16_t *A, const void *W, int32_t wtype,
                                       const float *bias, const uint16_t *resid, const float *gamma,
                                       const float *beta, int32_t fused, uint16_t *out);

/* Whole feed-forward block: out = LayerNorm(gelu(y W1^T + b1) W2^T + b2 + y) * gamma + beta, y [M][H] f16 bits,
 * W1 [I][H] and W2 [H][I] in file layout of `wtype`.  fused: 1 = single fused kernel (returns -2 if the
 * shape is not supported by it), 0 = the three-kernel path (GEMM+GELU, GEMM+residual, LayerNorm).   */
BERT_API int32_t bert_hip_test_ffn(int32_t M, int32_t H, int32_t I, const uint16_t *y, const void *W1, const void *W2,
                                   int32_t wtype, const float *b1, const float *b2, const float *gamma,
                                   const float *beta, int32_t fused, uint16_t *out);

/* qkv[T][3H] f16 bits (Q | K | V per row), packed sentences -> ctx[T][H] f16 bits.             */
BERT_API int32_t bert_hip_test_attention(int32_t n_sentences, const int32_t *cu_seqlens, int32_t n_head,
                                         int32_t d_head, const uint16_t *qkv, int32_t impl, uint16_t *out);

/* Q|K|V projection + attention: x[T][H] f16 bits, Wqkv [3H][H] (Q rows, K rows, V rows) in file layout of `wtype`,
 * bias[3H] -> ctx[T][H] f16 bits (reference bert.cpp:822-856).  fused: 1 = one kernel per sentence
 * (qkv_attention.hip; -2 if the shape is not supported), 0 = GEMM kernel + attention kernel.       */
BERT_API int32_t bert_hip_test_qkv_attention(int32_t n_sentences, const int32_t *cu_seqlens, int32_t n_head,
                                             int32_t d_head, const uint16_t *x, const void *Wqkv, int32_t wtype,
                                             const float *bias, int32_t fused, uint16_t *out);

/* Everything of a layer after the attention (reference bert.cpp:859-901):
 *   y = LayerNorm(ctx Wo^T + bo + x) * g1 + be1;  out = LayerNorm(gelu(y W1^T + b1) W2^T + b2 + y) * g2 + be2
 * ctx, x, out [M][H] f16 bits; Wo [H][H], W1 [I][H], W2 [H][I] in file layout of `wtype`.
 * impl: 0 = GEMM + LayerNorm kernels, 1 = token-owning-waves kernel (layer_tail.hip), 2 = panel kernel
 * (ffn_fused.hip with the leading projection phase); -2 if the shape is not supported by the chosen kernel.   */
BERT_API int32_t bert_hip_test_layer_tail(int32_t M, int32_t H, int32_t I, const uint16_t *ctx, const uint16_t *x,
                                          const void *Wo, const void *W1, const void *W2, int32_t wtype,
                                          const float *bo, const float *g1, const float *be1, const float *b1,
                                          const float *b2, const float *g2, const float *be2, int32_t impl,
                                          uint16_t *out);

/* Average milliseconds of `iters` launches of the fused feed-forward kernel on device-resident random data
 * (tuning helper of tools/bench_ffn.py; negative on error).                                              */
BERT_API float bert_hip_bench_ffn(int32_t M, int32_t H, int32_t I, int32_t iters);

BERT_API const char *bert_hip_version(void);

#ifdef __cplusplus
}
#endif

#endif /* BERT_HIP_H */
