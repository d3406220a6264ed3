// bert_api.cpp — the C ABI of libbert.so: bert.h (the reference's API, symbol for symbol) and the
// bert_hip.h extensions.  Host-side glue only; every FLOP of bert_eval runs in the HIP kernels.
//
// Reference behaviour mirrored here (reference file:line):
//   bert_load_from_file  bert.cpp:331-694    bert_free          bert.cpp:715-718
//   bert_tokenize        bert.cpp:252-325    bert_eval          bert.cpp:720-728
//   bert_eval_batch      bert.cpp:730-941    bert_encode        bert.cpp:943-950
//   bert_encode_batch    bert.cpp:952-1022   accessors          bert.cpp:111-134
//   bert_params_parse    bert.cpp:140-193
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bert.h"
#include "../../include/bert_hip.h"
#include <exception>
#include <new>
#include <system_error>
#include <utility>

#include "engine.h"
#include "model_file.h"
#include "multi_device.h"
#include "tokenizer.h"

using namespace bert_hip;

struct bert_ctx {
    HParams hp;
    Tokenizer tok;
    // one engine (weight replica + stream + workspace) per GPU; empty for tokenizer-only contexts.  Devices:
    // BERT_HIP_DEVICES ("all" or a comma-separated list without repeats), else the
    // calling thread's CURRENT device — one context = one GPU unless the caller asks for more, like the reference's one
    // context = one compute arena (eight torch.distributed ranks that each load a model must not build 64 replicas)
    std::vector<std::unique_ptr<Engine>> engines;
    // host threads of the devices beyond the first, created once at load (multi_device.h)
    std::unique_ptr<ShardWorkers> workers;
    // host threads of the batch tokenizer (bert_encode_batch / bert_hip_tokenize_batch), created at the first call that asks for
    // them and kept: a group of 4096 texts tokenizes in about a millisecond, sixteen thread starts cost a third of that
    mutable std::unique_ptr<ShardWorkers> tok_workers;
    mutable int tok_workers_asked = 0;
    // bert_encode_batch's two groups of tokenized texts (one on the GPU, one being tokenized): kept between calls, grown only,
    // never zero-filled (the tokenizer writes what is read; 16384 texts x n_max_tokens ids are 32 MiB of pages to touch otherwise)
    struct EncodeGroup {
        std::unique_ptr<bert_vocab_id[]> ids, packed;      // [n][n_max_tokens] as tokenized; the same ids back to back
        size_t ids_cap = 0, packed_cap = 0;
        std::vector<int32_t> n_tokens, cu;
        int32_t n_ok = 0;                                   // texts in front of the first one the engine cannot take (= all of them)
    } enc_group[2];
    // test knobs (bert_hip_set_option "test_inject_bad_alloc" / "test_rccl_single"): the ABI's catch-all; the exchange step
    // through a 1-rank communicator on a single device
    bool inject_bad_alloc = false, rccl_single = false;
    Engine *engine() const { return engines.empty() ? nullptr : engines[0].get(); }
    // device-resident results of bert_hip_eval_packed_gather: shard buffers (two per device: super-batch k + 1 is computed
    // into one while the exchange of super-batch k reads the other) and the gathered matrix, per device
    std::vector<std::unique_ptr<DevBuf>> shard_out, gathered;
    // the exchange's own stream per device and an event per (device, shard buffer): "the exchange that read this buffer is done"
    std::vector<hipStream_t> xstream;
    std::vector<hipEvent_t> xdone;
    int gather_super_tokens = 0;        // option "gather_super_tokens": tokens per device and super-batch (0: four device chunks)
    RcclGather rccl;
    ~bert_ctx() {
        for (size_t d = 0; d < xstream.size(); ++d) {
            if (d < engines.size()) (void)hipSetDevice(engines[d]->device());
            if (xstream[d]) { (void)hipStreamSynchronize(xstream[d]); (void)hipStreamDestroy(xstream[d]); }
        }
        for (hipEvent_t e : xdone) if (e) (void)hipEventDestroy(e);
    }
};

namespace {

bool quiet_env() {
    const char *q = getenv("BERT_HIP_QUIET");
    return q && *q && *q != '0';
}

// No exception may cross the C ABI: every extern "C" entry runs its body through one of these; an
// exception (std::bad_alloc from a staging vector, std::system_error from a thread, ...) becomes the reference's error
// convention — a line on stderr and an early return with the outputs untouched.
template <class F>
auto guarded(const char *name, decltype(std::declval<F>()()) on_error, F &&body) -> decltype(body()) {
    try {
        return body();
    } catch (const std::exception &e) {
        fprintf(stderr, "%s: %s\n", name, e.what());
    } catch (...) {
        fprintf(stderr, "%s: unknown exception\n", name);
    }
    return on_error;
}
template <class F>
void guarded_void(const char *name, F &&body) {
    try {
        body();
    } catch (const std::exception &e) {
        fprintf(stderr, "%s: %s\n", name, e.what());
    } catch (...) {
        fprintf(stderr, "%s: unknown exception\n", name);
    }
}

// the devices a context spreads over (see bert_ctx)
bool context_devices(std::vector<int> &devs, std::string &err) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        err = "no HIP device available (this library needs an AMD GPU; there is no CPU fallback)";
        return false;
    }
    const char *list = getenv("BERT_HIP_DEVICES");
    // (BERT_HIP_DEVICE=<n>, the single-device spelling of earlier builds: an alias for a list of one)
    if (!list || !*list) list = getenv("BERT_HIP_DEVICE");
    if (list && !*list) list = nullptr;
    devs.clear();
    if (list && *list && strcmp(list, "all") != 0) {
        for (const char *p = list; *p;) {
            char *end = nullptr;
            const long d = strtol(p, &end, 10);
            if (end == p) { err = std::string("BERT_HIP_DEVICES: cannot parse '") + list + "'"; return false; }
            if (d < 0 || d >= ndev) { err = "BERT_HIP_DEVICES: ordinal " + std::to_string(d) + " out of range"; return false; }
            devs.push_back((int)d);
            p = *end == ',' ? end + 1 : end;
        }
    } else if (list && strcmp(list, "all") == 0) {
        for (int d = 0; d < ndev; ++d) devs.push_back(d);
    } else {
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess || cur < 0 || cur >= ndev) cur = 0;
        devs.push_back(cur);
    }
    if (devs.empty()) { err = "BERT_HIP_DEVICES names no device"; return false; }
    for (size_t i = 0; i < devs.size(); ++i)
        for (size_t j = 0; j < i; ++j)
            if (devs[i] == devs[j]) { err = "BERT_HIP_DEVICES lists device " + std::to_string(devs[i]) + " twice"; return false; }
    return true;
}

bert_ctx *load_impl(const char *fname, bool tokenizer_only) {
    const bool quiet = quiet_env();
    if (!quiet) printf("%s: loading model from '%s' - please wait ...\n", "bert_load_from_file", fname);
    ModelFile mf;
    std::string err;
    if (!mf.load(fname, tokenizer_only, err)) {
        fprintf(stderr, "%s: %s\n", "bert_load_from_file", err.c_str());
        return nullptr;
    }
    if (!quiet) {
        printf("%s: n_vocab = %d\n%s: n_max_tokens   = %d\n%s: n_embd  = %d\n%s: n_intermediate  = %d\n"
               "%s: n_head  = %d\n%s: n_layer = %d\n%s: f16     = %d\n",
               "bert_load_from_file", mf.hp.n_vocab, "bert_load_from_file", mf.hp.n_max_tokens, "bert_load_from_file",
               mf.hp.n_embd, "bert_load_from_file", mf.hp.n_intermediate, "bert_load_from_file", mf.hp.n_head,
               "bert_load_from_file", mf.hp.n_layer, "bert_load_from_file", mf.hp.f16);
    }
    if (mf.legacy_q4 && !quiet)
        printf("%s: legacy q4 layout (f32 block scales, 20 / 24-byte blocks): re-blocked at load, scales rounded to f16\n", "bert_load_from_file");
    std::unique_ptr<bert_ctx> ctx(new bert_ctx);
    ctx->hp = mf.hp;
    ctx->tok.build(std::move(mf.vocab));
    ctx->tok.quiet = quiet;           // BERT_HIP_QUIET also drops the reference's per-byte "unknown token" stderr lines
    if (!tokenizer_only) {
        std::vector<int> devs;
        if (!context_devices(devs, err)) {
            fprintf(stderr, "%s: %s\n", "bert_load_from_file", err.c_str());
            return nullptr;
        }
        int caller_device = 0;
        const bool have_caller_device = hipGetDevice(&caller_device) == hipSuccess;
        // the replicas are built side by side (each upload is host-bound: repacking + H2D), one thread per extra device
        std::vector<Engine *> made(devs.size(), nullptr);
        std::vector<std::string> errs(devs.size());
        {
            std::vector<std::thread> builders;
            auto build = [&](size_t i) {
                try { made[i] = Engine::create(mf, devs[i], errs[i]); }
                catch (const std::exception &e) { errs[i] = e.what(); }
                catch (...) { errs[i] = "unknown exception"; }
            };
            for (size_t i = 1; i < devs.size(); ++i) {
                try { builders.emplace_back(build, i); } catch (const std::system_error &) { build(i); }
            }
            build(0);
            for (auto &th : builders) th.join();
        }
        if (have_caller_device) (void)hipSetDevice(caller_device);   // loading leaves the caller's current device alone
        bool ok = true;
        for (size_t i = 0; i < devs.size(); ++i) {
            if (made[i]) ctx->engines.emplace_back(made[i]);
            else if (ok) { fprintf(stderr, "%s: %s\n", "bert_load_from_file", errs[i].c_str()); ok = false; }
        }
        if (!ok) return nullptr;                              // (the engines made so far are freed with the context)
        if (devs.size() > 1) {
            ctx->workers.reset(new ShardWorkers((int)devs.size() - 1));
            // the communicator of the embedding gather is made now, not inside the first timed call
            if (!ctx->rccl.init(devs, err) && !quiet)
                fprintf(stderr, "%s: RCCL is not available (%s): bert_hip_eval_packed_gather will fail, everything else works\n", "bert_load_from_file", err.c_str());
            if (have_caller_device) (void)hipSetDevice(caller_device);
        }
        if (!quiet)
            printf("%s: model size = %8.2f MB / num tensors = %zu (HBM-resident on %zu HIP device%s, first %d)\n", "bert_load_from_file",
                   mf.total_tensor_bytes / 1024.0 / 1024.0, mf.tensors.size(), devs.size(), devs.size() == 1 ? "" : "s", devs[0]);
    }
    return ctx.release();
}

// Validates sentence b; returns false (after the reference's stderr message) if it cannot be evaluated.
bool sentence_ok(const bert_ctx *ctx, const bert_vocab_id *toks, int32_t n) {
    if (n > ctx->hp.n_max_tokens) {
        fprintf(stderr, "Too many tokens, maximum is %d\n", ctx->hp.n_max_tokens);   // reference bert.cpp:765-769
        return false;
    }
    if (n <= 0 || !toks) {
        fprintf(stderr, "bert_eval_batch: empty input\n");
        return false;
    }
    for (int32_t i = 0; i < n; ++i)
        if (toks[i] < 0 || toks[i] >= ctx->hp.n_vocab) {
            fprintf(stderr, "bert_eval_batch: token id %d out of range [0, %d)\n", toks[i], ctx->hp.n_vocab);
            return false;
        }
    return true;
}

// One packed batch over all devices of the context: contiguous token-balanced shards, one host thread per device, every
// shard's embeddings written straight into the caller's rows (or, d_dst: into the shard's device buffer).  Batches of
// fewer than MIN_SHARD_TOKENS tokens per device stay on the first device (a launch sequence costs ~50 us whatever the size).
constexpr long long MIN_SHARD_TOKENS = 2048;
int eval_packed_all_devices(bert_ctx *ctx, const int32_t *tokens, const int32_t *cu, int B, float *embeddings, std::string &err,
                            std::vector<int> *bounds_out = nullptr, float *const *d_dst = nullptr) {
    const int H = ctx->hp.n_embd;
    int n_dev = (int)ctx->engines.size();
    const long long total = (long long)cu[B] - cu[0];
    if (!d_dst)
        while (n_dev > 1 && total < MIN_SHARD_TOKENS * n_dev) --n_dev;
    std::vector<int> bounds;
    shard_bounds(cu, B, n_dev, bounds);
    if (bounds_out) *bounds_out = bounds;
    std::vector<std::string> errs((size_t)n_dev);
    auto eval = [&](int r, int b0, int b1) {
        // eval_packed_host takes the global token array and a window of the prefix sums
        return ctx->engines[r]->eval_packed_host(tokens, cu + b0, b1 - b0, embeddings ? embeddings + (size_t)b0 * H : nullptr, errs[r],
                                                 d_dst ? d_dst[r] : nullptr);
    };
    int rc;
    if (n_dev == 1 || !ctx->workers) {
        rc = 0;
        for (int r = 0; r < n_dev && rc == 0; ++r)
            if (bounds[r + 1] > bounds[r]) rc = eval(r, bounds[r], bounds[r + 1]);
    } else {
        rc = ctx->workers->run(bounds, eval, &err);           // (a worker's exception arrives here as rc -9 + message)
    }
    if (rc != 0 && err.empty())
        for (auto &e : errs)
            if (!e.empty()) { err = e; break; }
    return rc;
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------------------------------------
// bert.h
// ------------------------------------------------------------------------------------------------
bool bert_params_parse(int argc, char **argv, bert_params &params) {
    auto usage = [&]() {
        fprintf(stderr, "usage: %s [options]\n\noptions:\n", argv[0]);
        fprintf(stderr, "  -h, --help            show this help message and exit\n");
        fprintf(stderr, "  -t N, --threads N     accepted for compatibility, ignored by the GPU engine (default: %d)\n", params.n_threads);
        fprintf(stderr, "  -p PROMPT, --prompt PROMPT\n                        text to embed (default: %s)\n", params.prompt);
        fprintf(stderr, "  --port p              port to bind in server mode (default: %d)\n", params.port);
        fprintf(stderr, "  -m FNAME, --model FNAME\n                        model path (default: %s)\n\n", params.model);
    };
    for (int i = 1; i < argc; ++i) {
        const char *arg = argv[i];
        const bool has_value = i + 1 < argc;
        auto is = [&](const char *a, const char *b) { return strcmp(arg, a) == 0 || strcmp(arg, b) == 0; };
        if (is("-t", "--threads") && has_value) params.n_threads = atoi(argv[++i]);
        else if (is("-p", "--prompt") && has_value) params.prompt = argv[++i];
        else if (strcmp(arg, "--port") == 0 && has_value) params.port = atoi(argv[++i]);
        else if (is("-m", "--model") && has_value) params.model = argv[++i];
        else {
            if (!is("-h", "--help")) fprintf(stderr, "error: unknown argument: %s\n", arg);
            usage();
            exit(0);   // the reference exits with status 0 on both paths (bert.cpp:180-189)
        }
    }
    return true;
}

struct bert_ctx *bert_load_from_file(const char *fname) {
    return guarded("bert_load_from_file", (bert_ctx *)nullptr, [&] { return load_impl(fname, false); });
}

void bert_free(struct bert_ctx *ctx) {
    guarded_void("bert_free", [&] { delete ctx; });
}

int32_t bert_n_embd(struct bert_ctx *ctx) { return ctx->hp.n_embd; }
int32_t bert_n_max_tokens(struct bert_ctx *ctx) { return ctx->hp.n_max_tokens; }
const char *bert_vocab_id_to_token(struct bert_ctx *ctx, bert_vocab_id id) { return ctx->tok.id_to_token(id); }

void bert_tokenize(struct bert_ctx *ctx, const char *text, bert_vocab_id *tokens, int32_t *n_tokens, int32_t n_max_tokens) {
    guarded_void("bert_tokenize", [&] { ctx->tok.tokenize(text, tokens, n_tokens, n_max_tokens); });
}

// B validated sentences, packed, into the caller's rows: B, or -1 on a device error
static int32_t eval_packed_rows(struct bert_ctx *ctx, const bert_vocab_id *packed, const int32_t *cu, int32_t B, float *const *batch_embeddings) {
    const int H = ctx->hp.n_embd;
    std::string err;
    // the caller's rows are usually the rows of ONE matrix (NumPy rows through ctypes: reference examples/sample_dylib.py:50-51):
    // then the engine writes them in place; scattered rows go through a matrix of our own.  (A device error half way through a
    // call of several chunks leaves the rows of the finished chunks written in the first case, nothing in the second.)
    bool rows_of_one_matrix = true;
    for (int32_t b = 1; b < B && rows_of_one_matrix; ++b) rows_of_one_matrix = batch_embeddings[b] == batch_embeddings[0] + (size_t)b * H;
    if (rows_of_one_matrix) {
        if (eval_packed_all_devices(ctx, packed, cu, B, batch_embeddings[0], err) != 0) {
            fprintf(stderr, "bert_eval_batch: %s\n", err.c_str());
            return -1;
        }
        return B;
    }
    std::vector<float> out((size_t)B * H);
    if (eval_packed_all_devices(ctx, packed, cu, B, out.data(), err) != 0) {
        fprintf(stderr, "bert_eval_batch: %s\n", err.c_str());
        return -1;
    }
    for (int32_t b = 0; b < B; ++b) memcpy(batch_embeddings[b], out.data() + (size_t)b * H, sizeof(float) * H);
    return B;
}

// returns the number of sentences evaluated (stops in front of the first one it cannot handle), -1 on a device error
static int32_t eval_batch_impl(struct bert_ctx *ctx, int32_t n_batch_size, bert_vocab_id *const *batch_tokens,
                               const int32_t *n_tokens, float *const *batch_embeddings) {
    if (!ctx->engine()) { fprintf(stderr, "bert_eval_batch: this context has no device weights (tokenizer-only)\n"); return -1; }
    if (n_batch_size <= 0) return 0;
    if (ctx->inject_bad_alloc) throw std::bad_alloc();           // test knob: the path an exhausted host takes
    // The reference evaluates sentences in order and stops at the first one it cannot handle,
    // leaving later outputs untouched; keep that observable behaviour.
    int32_t B = 0;
    for (; B < n_batch_size; ++B)
        if (!sentence_ok(ctx, batch_tokens[B], n_tokens[B])) break;
    if (B == 0) return 0;
    std::vector<int32_t> cu(B + 1, 0);
    for (int32_t b = 0; b < B; ++b) cu[b + 1] = cu[b] + n_tokens[b];
    std::vector<int32_t> packed((size_t)cu[B]);
    for (int32_t b = 0; b < B; ++b) memcpy(packed.data() + cu[b], batch_tokens[b], sizeof(int32_t) * n_tokens[b]);
    return eval_packed_rows(ctx, packed.data(), cu.data(), B, batch_embeddings);
}

void bert_eval_batch(struct bert_ctx *ctx, int32_t /*n_threads*/, int32_t n_batch_size, bert_vocab_id **batch_tokens,
                     int32_t *n_tokens, float **batch_embeddings) {
    if (!batch_embeddings) return;   // the reference's memory-probe mode (bert.cpp:739); nothing to size here
    guarded_void("bert_eval_batch", [&] { (void)eval_batch_impl(ctx, n_batch_size, batch_tokens, n_tokens, batch_embeddings); });
}

void bert_eval(struct bert_ctx *ctx, int32_t n_threads, bert_vocab_id *tokens, int32_t n_tokens, float *embeddings) {
    bert_eval_batch(ctx, n_threads, 1, &tokens, &n_tokens, embeddings ? &embeddings : nullptr);
}

// Tokenizes n_inputs texts into tokens[i * n_max_tokens ..] on up to n_threads host threads (the tokenizer itself is const
// and re-entrant; inputs are handed out in blocks of 16 from a shared counter).  The persistent worker pool belongs to the
// context and is created / grown here: like every entry point of a bert_ctx this function is NOT re-entrant on one context
// (calls are serialised by the caller; bert_encode_batch's tokenize-ahead thread is the only caller while it runs).
static void tokenize_many(const bert_ctx *ctx, int32_t n_threads, int32_t n_inputs, const char **texts,
                          bert_vocab_id *tokens, int32_t *n_tokens) {
    const int32_t N = ctx->hp.n_max_tokens;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    int nt = std::min<int>({n_threads > 0 ? n_threads : 1, (int)hw, (n_inputs + 31) / 32});
    if (nt <= 1) {
        for (int32_t i = 0; i < n_inputs; ++i) ctx->tok.tokenize(texts[i], tokens + (size_t)i * N, &n_tokens[i], N);
        return;
    }
    std::atomic<int32_t> next{0};
    auto work = [&] {
        for (;;) {
            const int32_t i0 = next.fetch_add(16);
            if (i0 >= n_inputs) break;
            const int32_t i1 = std::min(n_inputs, i0 + 16);
            for (int32_t i = i0; i < i1; ++i) ctx->tok.tokenize(texts[i], tokens + (size_t)i * N, &n_tokens[i], N);
        }
    };
    // persistent workers (a thread that could not be started is not an error: the others, at least the caller, take its share)
    // (rebuilt only when MORE threads are asked for than were ever asked for: a pool that came up short — a thread that could
    // not be started — is kept, not torn down and recreated on every call)
    if (!ctx->tok_workers || ctx->tok_workers_asked < nt - 1) { ctx->tok_workers.reset(new ShardWorkers(nt - 1)); ctx->tok_workers_asked = nt - 1; }
    std::vector<int> each((size_t)nt + 1);
    for (int k = 0; k <= nt; ++k) each[k] = k;                   // "shard" k = worker k's turn at the shared counter
    std::string err;
    if (ctx->tok_workers->run(each, [&](int, int, int) { work(); return 0; }, &err) != 0) throw std::runtime_error("tokenizer worker: " + err);
}

// number of inputs encoded (stops at the first failure, later outputs untouched)
static int32_t encode_batch_impl(struct bert_ctx *ctx, int32_t n_threads, int32_t n_inputs, const char **texts, float **embeddings) {
    if (n_inputs <= 0) return 0;
    const int32_t N = ctx->hp.n_max_tokens;
    // Tokenize on n_threads host threads (at 10^6 sentences/s on the GPU the tokenizer is the stage in front of the
    // path that has to keep up) and evaluate as packed device batches (the reference sorts by length and loops with
    // batch size 1, bert.cpp:960-1020; per-sentence results do not depend on batching).  Inputs go through in groups:
    // group g+1 is tokenized AND PACKED while group g is on the GPU (cu_seqlens and the ids back to back, what
    // bert_eval_batch would do first thing with the GPU idle: 0.1 us per text, 1.7 ms for 16384 — round 5's trace of this
    // function), and the id buffers stay bounded for any n_inputs.  The groups GROW — 2048, 4096, 8192, then 16384 texts: the
    // first one is all a caller waits for with an idle GPU, later ones amortise the fixed costs of a blocking evaluation
    // and fill the GPU better (1.17 M texts/s at 2048 texts of 25 tokens, 1.37 M at 16384); a remainder of less than a quarter
    // of a group joins the last one.  Tokenizing a group of twice the size still fits under its predecessor's evaluation.
    if (!ctx->engine()) { fprintf(stderr, "bert_encode_batch: this context has no device weights (tokenizer-only)\n"); return -1; }
    if (ctx->inject_bad_alloc) throw std::bad_alloc();           // test knob: the path an exhausted host takes
    auto group_size = [](int k, int32_t left) {
        const int32_t g = (int32_t)(2048 << std::min(k, 3));
        return left - g < g / 4 ? left : g;
    };
    bert_ctx::EncodeGroup *groups = ctx->enc_group;
    auto tokenize_group = [&](bert_ctx::EncodeGroup &g, int32_t i0, int32_t n) {
        const size_t need = (size_t)N * n;
        if (g.ids_cap < need) { g.ids.reset(new bert_vocab_id[need]); g.ids_cap = need; }
        g.n_tokens.resize(n);
        tokenize_many(ctx, n_threads, n, texts + i0, g.ids.get(), g.n_tokens.data());
        g.cu.resize((size_t)n + 1);
        g.cu[0] = 0;
        g.n_ok = n;
        for (int32_t i = 0; i < n; ++i) {
            // (the tokenizer's ids are in range by construction; its counts are 2 .. n_max_tokens)
            if (g.n_tokens[i] <= 0 || g.n_tokens[i] > N) { g.n_ok = i; break; }
            g.cu[i + 1] = g.cu[i] + g.n_tokens[i];
        }
        const size_t T = (size_t)g.cu[g.n_ok];
        if (g.packed_cap < T) { g.packed.reset(new bert_vocab_id[T + T / 4]); g.packed_cap = T + T / 4; }
        for (int32_t i = 0; i < g.n_ok; ++i) memcpy(g.packed.get() + g.cu[i], g.ids.get() + (size_t)i * N, sizeof(bert_vocab_id) * g.n_tokens[i]);
    };
#ifdef BERT_HIP_HOST_TRACE
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
#endif
    tokenize_group(groups[0], 0, group_size(0, n_inputs));
#ifdef BERT_HIP_HOST_TRACE
    fprintf(stderr, "[encode] first group tokenized in %.3f ms\n", now() - t_begin);
#endif
    int32_t total = 0;
    for (int32_t i0 = 0, k = 0; i0 < n_inputs; ++k) {
        const int32_t n = group_size(k, n_inputs - i0), n_next = n_inputs - i0 - n > 0 ? group_size(k + 1, n_inputs - i0 - n) : 0;
        std::thread ahead;
        std::exception_ptr ahead_error;
        if (n_next > 0) {
            auto job = [&, k, i0, n, n_next] {
                try { tokenize_group(groups[(k + 1) & 1], i0 + n, n_next); } catch (...) { ahead_error = std::current_exception(); }
            };
            try { ahead = std::thread(job); } catch (const std::system_error &) { job(); }     // no thread: tokenize in line
        }
        int32_t done = -1;
        std::exception_ptr eval_error;
#ifdef BERT_HIP_HOST_TRACE
        const double t_e0 = now();
#endif
        try {
            bert_ctx::EncodeGroup &g = groups[k & 1];
            if (g.n_ok < n) fprintf(stderr, "bert_encode_batch: input %d cannot be evaluated (%d tokens)\n", i0 + g.n_ok, g.n_tokens[g.n_ok]);
            done = g.n_ok > 0 ? eval_packed_rows(ctx, g.packed.get(), g.cu.data(), g.n_ok, embeddings + i0) : 0;
        } catch (...) { eval_error = std::current_exception(); }
#ifdef BERT_HIP_HOST_TRACE
        const double t_e1 = now();
#endif
        if (ahead.joinable()) ahead.join();                   // never leave the scope with a running thread
#ifdef BERT_HIP_HOST_TRACE
        fprintf(stderr, "[encode] group %d: %d texts, eval %.3f ms, then waited %.3f ms for the tokenizer\n", k, n, t_e1 - t_e0, now() - t_e1);
#endif
        if (eval_error) std::rethrow_exception(eval_error);
        if (ahead_error) std::rethrow_exception(ahead_error);
        total += done > 0 ? done : 0;
        if (done != n) break;                                 // outputs after the failure stay untouched
        i0 += n;
    }
    return total;
}

void bert_encode_batch(struct bert_ctx *ctx, int32_t n_threads, int32_t /*n_batch_size*/, int32_t n_inputs,
                       const char **texts, float **embeddings) {
    guarded_void("bert_encode_batch", [&] { (void)encode_batch_impl(ctx, n_threads, n_inputs, texts, embeddings); });
}

void bert_encode(struct bert_ctx *ctx, int32_t n_threads, const char *texts, float *embeddings) {
    bert_encode_batch(ctx, n_threads, 1, 1, &texts, &embeddings);
}

// ------------------------------------------------------------------------------------------------
// bert_hip.h
// ------------------------------------------------------------------------------------------------
struct bert_ctx *bert_hip_load_tokenizer(const char *fname) {
    return guarded("bert_hip_load_tokenizer", (bert_ctx *)nullptr, [&] { return load_impl(fname, true); });
}

int32_t bert_hip_encode_batch(struct bert_ctx *ctx, int32_t n_threads, int32_t n_inputs, const char **texts, float **embeddings) {
    return guarded("bert_hip_encode_batch", (int32_t)-1, [&] { return encode_batch_impl(ctx, n_threads, n_inputs, texts, embeddings); });
}

int32_t bert_hip_tokenize_batch(struct bert_ctx *ctx, int32_t n_threads, int32_t n_inputs, const char **texts,
                                bert_vocab_id *tokens, int32_t *n_tokens) {
    if (!ctx || n_inputs < 0 || (n_inputs > 0 && (!texts || !tokens || !n_tokens))) return -1;
    return guarded("bert_hip_tokenize_batch", (int32_t)-1, [&] { tokenize_many(ctx, n_threads, n_inputs, texts, tokens, n_tokens); return (int32_t)0; });
}

int32_t bert_hip_n_layer(struct bert_ctx *ctx) { return ctx->hp.n_layer; }
int32_t bert_hip_n_head(struct bert_ctx *ctx) { return ctx->hp.n_head; }
int32_t bert_hip_n_intermediate(struct bert_ctx *ctx) { return ctx->hp.n_intermediate; }
int32_t bert_hip_n_vocab(struct bert_ctx *ctx) { return ctx->hp.n_vocab; }
int32_t bert_hip_ftype(struct bert_ctx *ctx) { return ctx->hp.f16; }
int32_t bert_hip_device(struct bert_ctx *ctx) { return ctx->engine() ? ctx->engine()->device() : -1; }
int32_t bert_hip_n_devices(struct bert_ctx *ctx) { return (int32_t)ctx->engines.size(); }

int32_t bert_hip_eval_packed(struct bert_ctx *ctx, const bert_vocab_id *tokens, const int32_t *cu_seqlens,
                             int32_t n_sentences, float *embeddings) {
    return guarded("bert_hip_eval_packed", (int32_t)-4, [&]() -> int32_t {
        if (!ctx->engine()) { fprintf(stderr, "bert_hip_eval_packed: tokenizer-only context\n"); return -1; }
        if (n_sentences <= 0) return 0;
        for (int32_t b = 0; b < n_sentences; ++b)
            if (!sentence_ok(ctx, tokens + cu_seqlens[b], cu_seqlens[b + 1] - cu_seqlens[b])) return -2;
        std::string err;
        if (eval_packed_all_devices(ctx, tokens, cu_seqlens, n_sentences, embeddings, err) != 0) {
            fprintf(stderr, "bert_hip_eval_packed: %s\n", err.c_str());
            return -3;
        }
        return 0;
    });
}

int32_t bert_hip_eval_packed_gather(struct bert_ctx *ctx, const bert_vocab_id *tokens, const int32_t *cu_seqlens,
                                    int32_t n_sentences, float **d_embeddings) {
    return guarded("bert_hip_eval_packed_gather", (int32_t)-4, [&]() -> int32_t {
        const char *me = "bert_hip_eval_packed_gather";
        if (!ctx->engine()) { fprintf(stderr, "%s: tokenizer-only context\n", me); return -1; }
        if (n_sentences <= 0) return 0;
        for (int32_t b = 0; b < n_sentences; ++b)
            if (!sentence_ok(ctx, tokens + cu_seqlens[b], cu_seqlens[b + 1] - cu_seqlens[b])) return -2;
        const int n_dev = (int)ctx->engines.size(), H = ctx->hp.n_embd;
        const bool exchange = n_dev > 1 || ctx->rccl_single;      // (test_rccl_single: a single device runs the step on a 1-rank communicator)
        std::string err;
        // (a failure after the first exchange has been issued must not return while earlier exchanges still write the gathered
        // matrices and read the shard buffers: drain every exchange and engine stream first)
        bool issued = false;
        auto fail = [&](const std::string &what) {
            fprintf(stderr, "%s: %s\n", me, what.c_str());
            if (issued)
                for (int d = 0; d < (int)ctx->engines.size(); ++d) {
                    if (hipSetDevice(ctx->engines[d]->device()) != hipSuccess) continue;
                    (void)hipStreamSynchronize(ctx->engines[d]->stream());
                    if (d < (int)ctx->xstream.size() && ctx->xstream[d]) (void)hipStreamSynchronize(ctx->xstream[d]);
                }
            return (int32_t)-3;
        };
        // per device: two shard buffers and the gathered [n_sentences][H] matrix (grow-only, owned by the context)
        if (ctx->shard_out.empty())
            for (int d = 0; d < 2 * n_dev; ++d) { ctx->shard_out.emplace_back(new DevBuf); if (d < n_dev) ctx->gathered.emplace_back(new DevBuf); }
        // SUPER-BATCHES ("one gather per super-batch, overlapped with the next super-batch's compute"): the call is
        // cut into runs of sentences of about `super` tokens per device; every run is sharded over the devices by token count
        // like a call of its own, and its exchange is issued on the devices' EXCHANGE streams as soon as its shards are
        // computed — it runs under the next run's compute.  Rows land at their global positions, so the result does not depend
        // on the cut.  A call that fits one run is one shard per device and one exchange, as before.
        const long long super = ctx->gather_super_tokens > 0 ? ctx->gather_super_tokens : 4ll * 262144;
        std::vector<int> runs{0};
        if (exchange) {
            const long long per_run = super * n_dev;
            for (int b = 1; b <= n_sentences; ++b)
                if (b == n_sentences || (long long)cu_seqlens[b + 1] - cu_seqlens[runs.back()] > per_run) runs.push_back(b);
        } else {
            runs.push_back(n_sentences);
        }
        const int n_runs = (int)runs.size() - 1;
        // every fallible preparation happens before any rank enters RCCL: a rank that fails between its peers' collectives leaves
        // them waiting in a collective that never completes
        std::vector<int> devs;
        for (auto &e : ctx->engines) devs.push_back(e->device());
        size_t max_rows = 1;
        std::vector<std::vector<int>> run_bounds((size_t)n_runs);
        for (int k = 0; k < n_runs; ++k) {
            shard_bounds(cu_seqlens + runs[k], runs[k + 1] - runs[k], n_dev, run_bounds[k]);
            for (int d = 0; d < n_dev; ++d) max_rows = std::max(max_rows, (size_t)(run_bounds[k][d + 1] - run_bounds[k][d]));
        }
        if (exchange && (int)ctx->xstream.size() < n_dev) { ctx->xstream.resize(n_dev, nullptr); ctx->xdone.resize(2 * n_dev, nullptr); }
        std::vector<float *> dst((size_t)n_dev);
        for (int d = 0; d < n_dev; ++d) {
            if (hipSetDevice(devs[d]) != hipSuccess) return fail("hipSetDevice failed");
            for (int sl = 0; sl < (n_runs > 1 ? 2 : 1); ++sl)
                if (!ctx->shard_out[2 * d + sl]->ensure(max_rows * H * 4, err)) return fail(err);
            if (exchange && !ctx->gathered[d]->ensure((size_t)n_sentences * H * 4, err)) return fail(err);
            // (no exchange: the one device's shard buffer IS the result)
            dst[d] = exchange ? ctx->gathered[d]->as<float>() : ctx->shard_out[2 * d]->as<float>();
            if (exchange) {
                if (!ctx->xstream[d] && hipStreamCreateWithFlags(&ctx->xstream[d], hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
                for (int sl = 0; sl < 2; ++sl)
                    if (!ctx->xdone[2 * d + sl] && hipEventCreateWithFlags(&ctx->xdone[2 * d + sl], hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate failed");
            }
        }
        if (exchange && !ctx->rccl.init(devs, err)) return fail(err);
        const bool threaded = exchange && ctx->workers && ctx->workers->n_threads() == n_dev - 1;
        for (int k = 0; k < n_runs; ++k) {
            const int sl = k & 1, b0 = runs[k], nb = runs[k + 1] - b0;
            std::vector<float *> src((size_t)n_dev);
            for (int d = 0; d < n_dev; ++d) {
                src[d] = ctx->shard_out[2 * d + sl]->as<float>();
                // the exchange of run k - 2 read this buffer
                if (exchange && k >= 2 && (hipSetDevice(devs[d]) != hipSuccess || hipEventSynchronize(ctx->xdone[2 * d + sl]) != hipSuccess))
                    return fail("waiting for an exchange failed");
            }
            // (blocking: the shards are complete in src when this returns)
            if (eval_packed_all_devices(ctx, tokens, cu_seqlens + b0, nb, nullptr, err, nullptr, src.data()) != 0) return fail(err);
            if (!exchange) break;
            // this run's exchange step (RCCL over xGMI): every device receives every other device's shard of the run, at rows
            // b0 + bounds of its matrix; not waited for here
            const std::vector<int> &bounds = run_bounds[k];
            bool ok;
            if (threaded) {
                // every device's call from the host thread that serves the device (worker d - 1, the caller for device 0)
                std::vector<int> each((size_t)n_dev + 1);
                for (int d = 0; d <= n_dev; ++d) each[d] = d;
                std::vector<std::string> errs((size_t)n_dev);
                const int rc = ctx->workers->run(each, [&](int d, int, int) {
                    return ctx->rccl.exchange_on(d, src[d], dst[d] + (size_t)b0 * H, bounds, H, ctx->xstream[d], errs[d]) ? 0 : -3; }, &err);
                for (auto &e : errs) if (err.empty() && !e.empty()) err = e;
                ok = rc == 0;
            } else {
                std::vector<float *> at((size_t)n_dev);
                for (int d = 0; d < n_dev; ++d) at[d] = dst[d] + (size_t)b0 * H;
                ok = ctx->rccl.all_gather(src.data(), at.data(), bounds, H, ctx->xstream.data(), err);
            }
            issued = true;                                    // (even a failed attempt may have queued part of the step)
            if (!ok) return fail(err);
            for (int d = 0; d < n_dev; ++d)
                if (hipSetDevice(devs[d]) != hipSuccess || hipEventRecord(ctx->xdone[2 * d + sl], ctx->xstream[d]) != hipSuccess) return fail("hipEventRecord failed");
        }
        for (int d = 0; d < n_dev; ++d) {
            if (hipSetDevice(devs[d]) != hipSuccess || hipStreamSynchronize(ctx->engines[d]->stream()) != hipSuccess ||
                (exchange && hipStreamSynchronize(ctx->xstream[d]) != hipSuccess))
                return fail("synchronisation failed");
            d_embeddings[d] = dst[d];
        }
        return 0;
    });
}

int32_t bert_hip_eval_packed_device(struct bert_ctx *ctx, const bert_vocab_id *d_tokens, const int32_t *d_cu_seqlens,
                                    int32_t n_sentences, int32_t n_tokens_total, int32_t max_len, float *d_embeddings,
                                    void *stream) {
    return guarded("bert_hip_eval_packed_device", (int32_t)-4, [&]() -> int32_t {
        if (!ctx->engine()) { fprintf(stderr, "bert_hip_eval_packed_device: tokenizer-only context\n"); return -1; }
        if (max_len > ctx->hp.n_max_tokens) { fprintf(stderr, "Too many tokens, maximum is %d\n", ctx->hp.n_max_tokens); return -2; }
        if (max_len <= 0 || n_tokens_total > (long long)n_sentences * max_len) {
            fprintf(stderr, "bert_hip_eval_packed_device: max_len = %d cannot hold %d tokens in %d sentences\n", max_len, n_tokens_total, n_sentences);
            return -2;
        }
        std::string err;
        if (ctx->engine()->eval_packed_device(d_tokens, d_cu_seqlens, n_sentences, n_tokens_total, max_len, d_embeddings,
                                              (hipStream_t)stream, nullptr, err) != 0) {
            fprintf(stderr, "bert_hip_eval_packed_device: %s\n", err.c_str());
            return -3;
        }
        return 0;
    });
}

int32_t bert_hip_reserve(struct bert_ctx *ctx, int32_t n_tokens, int32_t n_sentences) {
    return guarded("bert_hip_reserve", (int32_t)-4, [&]() -> int32_t {
        std::string err;
        for (auto &e : ctx->engines)
            if (!e->reserve(n_tokens, n_sentences, err)) { fprintf(stderr, "bert_hip_reserve: %s\n", err.c_str()); return -3; }
        return 0;
    });
}

int32_t bert_hip_check(struct bert_ctx *ctx) {
    return guarded("bert_hip_check", (int32_t)-4, [&]() -> int32_t {
        int32_t st = 0;
        std::string err;
        for (auto &e : ctx->engines) {
            const int s = e->check(err);
            if (s < 0) { fprintf(stderr, "bert_hip_check: %s\n", err.c_str()); return -3; }
            st |= s;
        }
        if (st) fprintf(stderr, "bert_hip_check: a device batch held a sentence longer than the max_len it was called with (or an empty one): its embeddings are NaN\n");
        return st;
    });
}

int32_t bert_hip_eval_hidden(struct bert_ctx *ctx, const bert_vocab_id *tokens, int32_t n_tokens, float *hidden,
                             float *embedding) {
    return guarded("bert_hip_eval_hidden", (int32_t)-4, [&]() -> int32_t {
        if (!ctx->engine()) { fprintf(stderr, "bert_hip_eval_hidden: tokenizer-only context\n"); return -1; }
        if (!sentence_ok(ctx, tokens, n_tokens)) return -2;
        std::string err;
        if (ctx->engine()->eval_hidden(tokens, n_tokens, hidden, embedding, err) != 0) {
            fprintf(stderr, "bert_hip_eval_hidden: %s\n", err.c_str());
            return -3;
        }
        return 0;
    });
}

void bert_hip_profile_enable(struct bert_ctx *ctx, int32_t on) {
    guarded_void("bert_hip_profile_enable", [&] { for (auto &e : ctx->engines) e->profile_enable(on != 0); });
}

int32_t bert_hip_profile_report(struct bert_ctx *ctx, char *buf, int32_t buf_len) {
    return guarded("bert_hip_profile_report", (int32_t)0, [&]() -> int32_t {
        if (!ctx->engine()) return 0;
        const std::string r = ctx->engine()->profile_report();      // (the first device's kernels)
        for (size_t d = 1; d < ctx->engines.size(); ++d) (void)ctx->engines[d]->profile_report();
        if (buf && buf_len > 0) {
            const size_t n = std::min((size_t)buf_len - 1, r.size());
            memcpy(buf, r.data(), n);
            buf[n] = 0;
        }
        return (int32_t)r.size();
    });
}

void bert_hip_set_option(struct bert_ctx *ctx, const char *key, const char *value) {
    guarded_void("bert_hip_set_option", [&] {
        if (!key || !value) return;
        if (strcmp(key, "test_inject_bad_alloc") == 0) ctx->inject_bad_alloc = *value == '1';
        else if (strcmp(key, "test_rccl_single") == 0) ctx->rccl_single = *value == '1';
        else if (strcmp(key, "gather_super_tokens") == 0) ctx->gather_super_tokens = std::max(0, atoi(value));
        else
            for (auto &e : ctx->engines) e->set_option(key, value);
    });
}

const char *bert_hip_version(void) { return "bert.cpp_amd 0.5 (gfx950)"; }

}  // extern "C"
