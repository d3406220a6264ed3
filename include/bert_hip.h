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
BERT_API int32_t bert_hip_device(struct bert_ctx *ctx);       /* HIP ordinal of the context's first device, -1 if none */
BERT_API int32_t bert_hip_n_devices(struct bert_ctx *ctx);    /* GPUs the context spreads its batches over */

/* bert_encode_batch with a result: the number of inputs encoded (all of them, or the inputs in front of the first one
 * that could not be evaluated — later embeddings stay untouched), negative on an internal error.                     */
BERT_API int32_t bert_hip_encode_batch(struct bert_ctx *ctx, int32_t n_threads, int32_t n_inputs, const char **texts,
                                       float **embeddings);

/* Packed, variable-length batch evaluation — the engine's native entry point; bert_eval_batch
 * (reference bert.cpp:730-941) is a wrapper that packs the per-sentence host pointers.
 *   tokens      [n_tokens_total] ids of all sentences back to back
 *   cu_seqlens  [n_sentences + 1] exclusive prefix sums of the sentence lengths (cu[0] = 0)
 *   embeddings  [n_sentences * bert_n_embd] row-major
 * Returns 0 on success, negative on error (message on stderr, outputs untouched).               */
BERT_API int32_t bert_hip_eval_packed(struct bert_ctx *ctx, const bert_vocab_id *tokens,
                                      const int32_t *cu_seqlens, int32_t n_sentences, float *embeddings);

/* Multi-GPU contexts (BERT_HIP_DEVICES=all or a list; default: the calling thread's current device only): bert_eval_batch / bert_encode_batch /
 * bert_hip_eval_packed cut a call into contiguous shards with near-equal token counts, one per device (weights are
 * replicated, each device has its own host thread and stream), and every shard writes its embeddings straight into
 * the caller's host rows.  bert_hip_eval_packed_gather keeps the results on the devices instead and runs the path's one
 * exchange step — an RCCL all-gather over xGMI (librccl.so is loaded, and the communicator made, when the model is loaded) — so that afterwards EVERY device
 * holds the whole [n_sentences][n_embd] f32 matrix: d_embeddings[d] receives the pointer on device d (owned by the
 * context, valid until the next call; bert_hip_n_devices entries).  Blocking.  Per-sentence results are the same bits
 * whatever the number of devices.  Returns 0, negative on error.                                                      */
BERT_API int32_t bert_hip_eval_packed_gather(struct bert_ctx *ctx, const bert_vocab_id *tokens, const int32_t *cu_seqlens,
                                             int32_t n_sentences, float **d_embeddings);

/* Same computation with every buffer already resident in HBM on the context's FIRST device; work is
 * enqueued on `stream` and NOT synchronised (the caller owns the stream).  `max_len` must be >=
 * the longest sentence (it selects and sizes the attention kernels); n_tokens_total = cu[n].
 * Rules of the asynchronous entry point:
 *   - a context has ONE workspace: a forward pass waits (on its own stream, hipStreamWaitEvent) for the previous pass
 *     of the context, whatever stream that ran on — passes never overlap, callers need no extra ordering;
 *   - the workspace grows on demand, and growing allocates (synchronises the device, illegal under stream capture):
 *     call bert_hip_reserve once with the largest batch first;
 *   - lengths are validated on the device: a sentence longer than max_len (or empty) yields a NaN embedding and sets a
 *     status word that bert_hip_check returns (and clears) after synchronising.  A batch shaped like full windows
 *     (n_tokens_total = 128 n_sentences, max_len = 128) is evaluated 128-token block by block: if it has that shape only by
 *     the sum of its lengths (an over-long sentence, a shorter one), every sentence that is not exactly its block gets a
 *     NaN row as well — the other rows are the bits they always have;
 *   - short sentences are packed several to a 128-slot attention window by a kernel of the pass itself (the lengths
 *     exist only in HBM here): results are the bits of the host entry points.                                        */
BERT_API int32_t bert_hip_eval_packed_device(struct bert_ctx *ctx, const bert_vocab_id *d_tokens,
                                             const int32_t *d_cu_seqlens, int32_t n_sentences,
                                             int32_t n_tokens_total, int32_t max_len,
                                             float *d_embeddings, void *stream);

BERT_API int32_t bert_hip_reserve(struct bert_ctx *ctx, int32_t n_tokens, int32_t n_sentences);
BERT_API int32_t bert_hip_check(struct bert_ctx *ctx);        /* 0 ok, 1 a batch broke its max_len promise, < 0 error */

/* Hidden-state tap for parity tests: one sentence, writes hidden[(n_layer+1)][n_tokens][n_embd]
 * f32 (after the embedding LayerNorm and after every encoder layer, reference bert.cpp:806-901)
 * and the final embedding.  Either output may be NULL.                                          */
BERT_API int32_t bert_hip_eval_hidden(struct bert_ctx *ctx, const bert_vocab_id *tokens, int32_t n_tokens,
                                      float *hidden, float *embedding);

/* Per-kernel timing with HIP events on the launch stream.  While enabled every kernel launch of
 * the forward pass carries an event pair (hipExtLaunchKernelGGL start / stop events: a timed launch runs behind system-scope
 * fences and reads up to 8 % long for sub-millisecond kernels, so never enable it inside a throughput measurement).  bert_hip_profile_report writes one line per
 * kernel: "<name> <launches> <total_ms> <flops_per_launch_avg>\n" and returns the number of bytes
 * it needed (excluding NUL); it synchronises the device first and resets the counters.  Behind the kernels it lists which
 * mat-mul kernel family served the weight GEMMs of the pass: "family:gemm256_f16" / "family:gemm256_q4" (256 x 256 tiles, f16
 * image / 4-bit planes dequantised in the tile load), "family:gemm_mfma_f16" / "_q4" (128 x 128 tiles), "family:gemm_naive",
 * each "<name> <launches> 0 0".  With bert_hip_set_option("profile_replay", "<kernel>:<K>") a pass runs untimed and the
 * first launch of <kernel> is followed by K repeats of itself between ONE event pair (the pair's cost spread over K launches;
 * in-place kernels then run on their own output: such a pass's results are not to be used); "" restores a pair per launch. */
BERT_API void    bert_hip_profile_enable(struct bert_ctx *ctx, int32_t on);
BERT_API int32_t bert_hip_profile_report(struct bert_ctx *ctx, char *buf, int32_t buf_len);

/* Environment, read by bert_load_from_file (eight switches):
 *   BERT_HIP_DEVICES       "all" or a comma-separated list of HIP ordinals without repeats: the GPUs of the context
 *                          (default: the calling thread's current device — one context, one GPU, unless asked otherwise);
 *                          BERT_HIP_DEVICE=<n>, the spelling of the first builds, is read as a list of one when this is unset
 *   BERT_HIP_KERNELS       "fused" (default): two launches per layer where the shape allows it — projection + attention of a
 *                          128-slot window (qkv_attention2.hip), everything behind the attention (layer_tail.hip) —, tiled kernels
 *                          elsewhere | "tiled": GEMM, attention and LayerNorm kernels only (Q|K|V and the intermediate through
 *                          HBM).  ("naive" — the generic kernels the parity tests compare against — is a route of libbert_test.so only;
 *                          libbert.so prints a note and ignores it.)
 *   BERT_HIP_Q4            "expand" (default) | "fused" — q4_0 / q4_1 weight matrices are expanded to f16 images in HBM once
 *                          at load, or stay 4-bit in HBM and are dequantised in the tile loads of the same kernels (same values,
 *                          same bits on the fused kernels; a quarter of the weight bytes)
 *   BERT_HIP_LATENCY       1 (default) | 0 | n — calls of at most n tokens (default 768: one sentence per call, the reference's callers,
 *                          and the small batches of a polling server) take the latency route: every mat-mul of a layer split by
 *                          output features and token blocks over many workgroups (skinny.hip); same bits as the batch route
 *                          (220 us per 128-token sentence, 330 us for 16 sentences of 25 tokens, host to host)
 *   BERT_HIP_F32           "exact" (default) | "f16" — f32 model files run in f32 arithmetic like the reference's (f32 activations,
 *                          v_mfma_f32_32x32x2_f32: f32_route.hip), or with their matrices rounded to f16 through the f16 kernels
 *   BERT_HIP_WINDOW_SLOTS  16 (default) | 8 — PROCESS-WIDE: sentences start at multiples of this many slots inside the 128-slot windows of the
 *                          fused attention kernels.  8 packs mean-25-token batches into about an eighth fewer windows; the price is
 *                          that a sentence's embedding then depends, in its last bits, on where it sits in its window — 16 slots are
 *                          one k-step of the P.V MFMAs, 8 are not (tools/ubench/mfma_shift.hip) — so "the same sentence gives the
 *                          same bits in any batch" holds only with 16.  Cosines against the CPU are unchanged.
 *   BERT_HIP_CHUNK_TOKENS  max tokens evaluated per device pass by the host API (default 262144)
 *   BERT_HIP_LN_FOLD       1 (default) | 0 — models on the 256 x 256-tile mat-mul route (H = 768): the LayerNorms folded into the mat-muls around
 *                          them (no LayerNorm launch but the last; roundings differ from the un-folded sequence in the last bits), or a
 *                          LayerNorm kernel per LayerNorm
 *   BERT_HIP_QUIET         1 = no progress text on stdout during load, no "unknown token" lines on stderr from bert_tokenize
 * bert_hip_set_option (after load; tests and tuning): "qkv2" / "tail" / "gemm256" / "latency" = "0" | "1" switch single kernels
 * of the fused family, "one_launch" = "0" | "1" (default: all layers in one launch for well-filled windows) | "2" (whenever the
 * kernel takes the batch), "gemm" / "attn" = "mfma" | "naive" (libbert_test.so), "ln_fold" = "0" | "1", "f32" = "exact" | "f16", "latency_tokens" = n, "window_slots" = "16" | "8" (process-wide default, read once
 * per forward pass), "chunk_tokens" = n, "gather_super_tokens" = n (bert_hip_eval_packed_gather: tokens per device and super-batch, 0 =
 * four device chunks), "stage_kernel" = "0" | "1" (host API: staged blocks of at most 256 KiB travel by a kernel that reads the mapped
 * pinned memory instead of the copy engine), "profile_replay" (above).                                                                 */
BERT_API void bert_hip_set_option(struct bert_ctx *ctx, const char *key, const char *value);

BERT_API const char *bert_hip_version(void);

#ifdef __cplusplus
}
#endif

#endif /* BERT_HIP_H */
