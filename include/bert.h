/* bert.h — the drop-in boundary of the MI355X-native engine.
 *
 * This header declares, symbol for symbol, the C API of skeskinen/bert.cpp (reference bert.h:27-82)
 * so that anything that links or dlopen()s the reference's `libbert.so` — the ctypes callers
 * (reference examples/sample_dylib.py:19-34, benchmarks/run_mteb.py:34-49), the dlsym skeleton
 * (examples/dylib.cpp:14-16), main.cpp / server.cpp — can be pointed at this library unchanged.
 * Only the implementation behind it differs: the tokenizer and model-file loader are plain C++,
 * and the bert_eval forward pass runs as hand-written HIP kernels for gfx950 on truly batched,
 * variable-length packed inputs.  Extensions that the reference does not have
 * (device-resident evaluation, per-kernel timing, hidden-state taps) live in bert_hip.h.
 *
 * Conventions kept from the reference:
 *   - no error codes: the loader returns NULL and explains on stderr; eval / encode return void and
 *     on failure (too many tokens, device error) print to stderr and leave the outputs untouched;
 *   - every token / embedding buffer is caller-allocated HOST memory; batch entry points take an
 *     array of per-sentence pointers;
 *   - calls are blocking: outputs are complete on return; a bert_ctx is not thread-safe;
 *   - `n_threads` has no meaning for a GPU engine: accepted and ignored;
 *   - `n_batch_size` is a hint whose value never changes results.
 *
 * Arithmetic by file type (reference: ggml's mat-mul per weight type):
 *   - f16 files: f16 operands, f32 accumulation — the reference's own class for these files;
 *   - q4_0 / q4_1 files: blocks dequantised to f16 ((q - 8) d, q d + m), f16 activations, f32 accumulation (the reference
 *     also quantises the activations to 8 bits per 32-block: this engine is the closer one to exact arithmetic);
 *   - f32 files: f32 weights, f32 activations, f32 accumulation on the matrix cores' f32 form (v_mfma_f32_32x32x2_f32), f32
 *     softmax / GELU / LayerNorm — the reference's pure-f32 mat-mul for this file type (bert.cpp:825 with GGML_TYPE_F32);
 *     max-abs 2e-5 per embedding component against f32 arithmetic on the CPU (tests).  BERT_HIP_F32=f16 (bert_hip.h) asks for
 *     the faster f16-operand kernels instead (weights rounded to f16 once at load; cosine >= 1 - 1e-4).
 */
#ifndef BERT_H
#define BERT_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#if defined(_WIN32)
#  define BERT_API __declspec(dllexport)
#else
#  define BERT_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

struct bert_ctx;                 /* opaque; owns host vocab + device-resident weights          */
typedef int32_t bert_vocab_id;   /* reference bert.h:31                                          */

/* Command-line parameters of the example programs (reference bert.h:18-25).  C++-only, exactly
 * as in the reference: default member initialisers and a reference parameter below.            */
#ifdef __cplusplus
struct bert_params {
    int32_t     n_threads = 6;
    int32_t     port      = 8080;                                           /* server mode */
    const char *model     = "models/all-MiniLM-L6-v2/ggml-model-q4_0.bin";
    const char *prompt    = "test prompt";
};
/* reference bert.h:27, bert.cpp:157-193: -t/--threads, -p/--prompt, --port, -m/--model, -h/--help;
 * an unknown argument prints the usage text and exit(0)s, like the reference.                  */
BERT_API bool bert_params_parse(int argc, char **argv, bert_params &params);
#endif

/* ---- lifetime (reference bert.h:33-34, bert.cpp:331-694, 715-718) ---------------------------- */
/* Parses a ggml-model-{f32,f16,q4_0,q4_1}.bin file, builds the tokenizer tables and uploads /
 * repacks all weights into HBM.  NULL + stderr message on any failure (including: no HIP device). */
BERT_API struct bert_ctx *bert_load_from_file(const char *fname);
BERT_API void             bert_free(struct bert_ctx *ctx);

/* ---- text in, embedding out (reference bert.h:38-52, bert.cpp:943-1022) ---------------------- */
/* embeddings: bert_n_embd(ctx) floats, L2-normalised mean-pooled sentence embedding.            */
BERT_API void bert_encode(struct bert_ctx *ctx, int32_t n_threads, const char *texts, float *embeddings);

/* texts[n_inputs], embeddings[n_inputs] (each bert_n_embd floats).  All inputs are tokenized and
 * evaluated as real device batches; results equal n_inputs independent bert_encode calls.       */
BERT_API void bert_encode_batch(struct bert_ctx *ctx, int32_t n_threads, int32_t n_batch_size,
                                int32_t n_inputs, const char **texts, float **embeddings);

/* ---- separate tokenization and evaluation (reference bert.h:56-77) --------------------------- */
/* tokens must hold n_max_tokens ids; writes [CLS] pieces... [SEP], *n_tokens <= n_max_tokens.
 * Bit-exact with reference bert.cpp:252-325 including its quirks.        */
BERT_API void bert_tokenize(struct bert_ctx *ctx, const char *text, bert_vocab_id *tokens,
                            int32_t *n_tokens, int32_t n_max_tokens);

/* One sentence: token ids -> embedding (reference bert.cpp:720-728).  embeddings == NULL selects
 * the reference's memory-probe mode, which is a no-op here.                                     */
BERT_API void bert_eval(struct bert_ctx *ctx, int32_t n_threads, bert_vocab_id *tokens,
                        int32_t n_tokens, float *embeddings);

/* B sentences of arbitrary lengths n_tokens[b] <= bert_n_max_tokens (reference bert.cpp:730-941).
 * No padding, no mask: every sentence is evaluated at its true length, token_type 0, positions
 * 0..N-1, mean-pool over all N tokens, L2 normalise.  The reference's note "the longest input
 * must be first" is accepted but not required.                                                  */
BERT_API void bert_eval_batch(struct bert_ctx *ctx, int32_t n_threads, int32_t n_batch_size,
                              bert_vocab_id **batch_tokens, int32_t *n_tokens, float **batch_embeddings);

/* ---- accessors (reference bert.h:79-82, bert.cpp:111-134) ------------------------------------ */
BERT_API int32_t     bert_n_embd(struct bert_ctx *ctx);
BERT_API int32_t     bert_n_max_tokens(struct bert_ctx *ctx);
BERT_API const char *bert_vocab_id_to_token(struct bert_ctx *ctx, bert_vocab_id id);

#ifdef __cplusplus
}
#endif

#endif /* BERT_H */
