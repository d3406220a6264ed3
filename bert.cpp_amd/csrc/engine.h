// engine.h — device side of a bert_ctx: HBM-resident weights, workspace, and the launch sequence
// of the forward pass.  Replaces the reference's ggml arena + per-sentence graph build + execute
// (reference bert.cpp:730-941, sizing :680-713) with a fixed kernel sequence (2 + 2*L launches on the fused path) over a packed
// variable-length batch.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"
#include "model_file.h"

namespace bert_hip {

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf();
    bool alloc(size_t n, std::string &err);                       // zero-filled
    bool upload(const void *src, size_t n, std::string &err);     // alloc + H2D
    bool ensure(size_t n, std::string &err);                      // grow only
    template <class T> T *as() const { return (T *)p; }
};

// Owns the HBM image of one weight matrix in the layouts kernels.h describes.
struct GemmWeightStore {
    GemmWeight w;
    DevBuf w16, w16p, qs, sc, naive16, w32;
    bool mfma_ok = false;
    // rows: list of (file tensor) stacked along N (one entry, or q|k|v).  All share type and K.
    // want_kperm: also build GemmWeight::w16p (f16 images only); expand_q4: q4_0 / q4_1 tensors become an f16 image at
    // load (default for the engine) instead of the nibble / scale planes of the fused-dequant kernels
    // want_f32: f32 tensors also keep their own f32 rows (GemmWeight::w32, the f32 route)
    bool build(const std::vector<const HostTensor *> &rows, bool want_naive, std::string &err, bool want_kperm = false,
               bool expand_q4 = false, bool want_f32 = false);
    // LayerNorm folded into a mat-mul that consumes LayerNorm(u; gamma, beta) (kernels.h GemmLnFold): the f16 image of
    // W diag(gamma) (this store) and the weight side of the statistics k-step, [N][16] f16: s_hi s_lo s_hi c_hi c_lo c_hi 0..
    // with s[n] = sum_k W'[n][k], c[n] = sum_k beta[k] W[n][k] + bias[n]
    bool build_ln_fold(const std::vector<const HostTensor *> &rows, const float *gamma, const float *beta, const float *bias, DevBuf &waug, std::string &err);
};

struct LayerWeights {
    GemmWeightStore qkv, o, ffi, ffo;
    // q4 files with the default BERT_HIP_Q4=expand: the stacked Q | K | V matrix ALSO as 4-bit planes when its f16 image
    // (3 H x H x 2 bytes) cannot stay in an XCD's 4 MiB L2 beside the activation tiles in flight — gemm256's persistent walk
    // then re-fetches the f16 image every round (2.40 GB per launch at bert-base dims against 0.81 GB with the planes, which
    // give the same bits and are 1-3 % faster on that launch: DESIGN.md §3).  Used by the QKV mat-mul of the gemm256 route only.
    GemmWeightStore qkv_q4;
    DevBuf qkv_b, o_b, ffi_b, ffo_b, ln_att_w, ln_att_b, ln_out_w, ln_out_b;
    // LayerNorm folded into the H = 768 mat-muls (kernels.h GemmLnFold): the up-projection with this layer's attention LayerNorm
    // folded in, the Q|K|V projection with the PREVIOUS layer's output LayerNorm (layers >= 1), their statistics columns, and
    // the packed (gamma, beta + bias) pairs of the two residual mat-muls (attention output: the previous layer's output LayerNorm)
    GemmWeightStore ffi_fold, qkv_fold;
    DevBuf ffi_waug, qkv_waug, o_gb, ffo_gb;
    bool fold_ok = false;
};

struct KernelStat { int launches = 0; double ms = 0.0; double flops = 0.0; };

class Engine {
public:
    // device: HIP ordinal the weights are uploaded to and every launch runs on
    static Engine *create(const ModelFile &mf, int device, std::string &err);
    ~Engine();

    // host-resident packed batch (validated by the caller), blocking
    // embeddings: host destination [n_sentences][H]; d_embeddings (optional, instead): destination in this device's memory
    int eval_packed_host(const int32_t *tokens, const int32_t *cu_seqlens, int n_sentences, float *embeddings,
                         std::string &err, float *d_embeddings = nullptr);
    // device-resident, asynchronous on `stream`
    // d_windows / n_windows: optional sentence windows of the fused projection+attention kernel (build_windows), in
    // device memory; without them the same windows are built on the device (launch_build_windows) when the batch is
    // short enough on average for packing to pay, else sentences are placed by the uniform rule of qkv_attention2.hip
    int eval_packed_device(const int32_t *d_tokens, const int32_t *d_cu, int n_sentences, int n_tokens, int max_len,
                           float *d_out, hipStream_t stream, float *d_hidden, std::string &err,
                           const int2 *d_windows = nullptr, int n_windows = 0, int window_slots = 0);      // window_slots: the place granularity d_windows was built with (0: read it now)
    // next-fit packing of whole sentences (in order, each starting at a multiple of 16 slots) into windows of 128 token
    // slots: {first sentence, count} per window.  Sentences longer than a window get one of their own (the fused kernel is
    // not used for such batches).
    static void build_windows(const int32_t *cu_seqlens, int n_sentences, std::vector<int2> &windows, int slots);
    int eval_hidden(const int32_t *tokens, int n_tokens, float *hidden, float *embedding, std::string &err);

    // sizes the workspace for batches of up to n_tokens tokens / n_sentences sentences now, so that later calls of
    // eval_packed_device never allocate (allocation synchronises the device and breaks stream capture)
    bool reserve(int n_tokens, int n_sentences, std::string &err);
    // device-side validation (sentence lengths vs max_len): synchronises, returns and clears the status word
    int check(std::string &err);
    void set_option(const std::string &key, const std::string &value);
    void profile_enable(bool on);
    std::string profile_report();

    const HParams &hparams() const { return hp_; }
    int device() const { return device_; }
    hipStream_t stream() const { return stream_; }

private:
    Engine() = default;
    bool ensure_workspace(int t_pad, int n_sentences, std::string &err);
    // f32 files in f32 arithmetic (f32_route.hip): the whole pass, one launch per operation
    int forward_f32(const int32_t *d_tokens, const int32_t *d_cu, int B, int T, int max_len, float *d_out, hipStream_t s, float *d_hidden,
                    std::string &err);
    template <class F> void timed(const char *name, double flops, hipStream_t s, F &&f);

    HParams hp_;
    int device_ = 0;
    int table_type_ = 0;
    DevBuf word_emb_, type_emb_, pos_emb_, ln_e_w_, ln_e_b_;
    std::vector<LayerWeights *> layers_;

    // workspace (grow-only)
    DevBuf x_, qkv_, ctx_, y_, ff_, v32_, d_tokens_, d_cu_, d_out_, d_hidden_, status_, windows_;
    DevBuf ln_stats1_, ln_stats2_, ln_rows1_, ln_rows2_;      // LayerNorm folding: per-row partial statistics / finalized rows of the two LayerNorms of a layer
    hipStream_t stream_ = nullptr;
    // the workspace serves ONE forward pass at a time: every pass waits for the previous one's event on its own stream
    hipEvent_t busy_ = nullptr;
    // host path: two sets of pinned staging + device id / embedding buffers, so that the host stages chunk i+1 and
    // unpacks chunk i-1 while the GPU computes chunk i (eval_packed_host)
    struct HostSlot {
        // ONE pinned staging block per chunk — ids | cu_seqlens | windows, each part 16-byte aligned — and one device image of
        // it: a single H2D copy per chunk.  The embeddings come back without a copy: the pooling kernel writes them straight
        // into h_out (pinned, mapped into the device's address space as d_out_host).
        char *h_in = nullptr, *d_in_host = nullptr;          // (d_in_host: h_in as the device sees it)
        float *h_out = nullptr, *d_out_host = nullptr;
        size_t h_in_cap = 0, h_out_cap = 0;
        DevBuf d_in, d_out;
        hipEvent_t done = nullptr;
    } slot_[2];

    // options
    bool gemm_naive_ = false, attn_naive_ = false, qkv2_ = true, gemm256_ = true, ln_fold_ = true, tail_ = true, latency_ = true, q4_expand_ = true;
    bool f32_file_ = false;           // every matrix and table of the file is f32: the f32 route can take it
    bool f32_exact_ = true;           // ... and takes it unless BERT_HIP_F32=f16 / set_option("f32", "f16")
    int one_launch_ = 1;              // all layers in one launch: 0 never, 1 when it pays (well-filled windows), 2 whenever the kernel takes the batch
    int chunk_tokens_ = 262144;
    // Calls of at most this many tokens take the latency route (skinny.hip).  A call of T tokens keeps ceil(T / 128) CUs busy on
    // the fused kernels — 615-685 us for anything from 129 to 3000 tokens of all-MiniLM-L6-v2 — while the route's time grows with T
    // from 220 us: 252 us at 172 tokens (8 sentences), 330 at 363 (16), 376 at 512, 527 at 716, 606 at 1024 (round 5, same bits).
    int latency_tokens_ = 768;
    bool stage_kernel_ = true;        // small staged blocks come in by a kernel that reads the mapped pinned block, not by the copy engine

    // profiling
    bool profiling_ = false;
    std::vector<hipEvent_t> ev_pool_;
    struct Pending { const char *name; hipEvent_t a, b; double flops; int launches; };
    std::string replay_name_;         // "profile_replay": time K repeats of this kernel between one event pair (engine.hip timed())
    int replay_k_ = 10;
    bool replay_done_ = false;
    std::vector<Pending> pending_;
    std::map<std::string, KernelStat> stats_;
    std::map<std::string, int> families_;       // profile: mat-mul launches per kernel family ("family:gemm256_q4" ...)
};

}  // namespace bert_hip
