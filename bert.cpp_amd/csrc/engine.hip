// engine.hip — see engine.h.
#include "engine.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <system_error>
#include <thread>

namespace bert_hip {

#define HIP_OK(expr, errvar, ret)                                                                       \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess) {                                                                        \
            errvar = std::string(#expr) + ": " + hipGetErrorString(e__);                                \
            return ret;                                                                                 \
        }                                                                                               \
    } while (0)

// ------------------------------------------------------------------------------------------------
// DevBuf
// ------------------------------------------------------------------------------------------------
DevBuf::~DevBuf() {
    if (p) (void)hipFree(p);
}
bool DevBuf::alloc(size_t n, std::string &err) {
    if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    if (n == 0) n = 16;
    HIP_OK(hipMalloc(&p, n), err, false);
    bytes = n;
    HIP_OK(hipMemset(p, 0, n), err, false);
    // the fill runs on the null stream and returns early; the engine's streams are non-blocking (not ordered
    // against it), so a kernel writing this buffer could be overtaken by the fill
    HIP_OK(hipDeviceSynchronize(), err, false);
    return true;
}
bool DevBuf::upload(const void *src, size_t n, std::string &err) {
    if (!alloc(n, err)) return false;
    if (n) HIP_OK(hipMemcpy(p, src, n, hipMemcpyHostToDevice), err, false);
    return true;
}
bool DevBuf::ensure(size_t n, std::string &err) {
    if (n <= bytes) return true;
    return alloc(n + n / 8, err);
}

// ------------------------------------------------------------------------------------------------
// weight repacking (host) -> HBM layouts of kernels.h
// ------------------------------------------------------------------------------------------------
namespace {

inline float h2f(uint16_t bits) { _Float16 h; memcpy(&h, &bits, 2); return (float)h; }

// row `r` of a file tensor dequantised to f16 (exactly representable for f16 files; nearest for f32 / q4)
void row_to_f16(const HostTensor &t, int64_t r, _Float16 *dst) {
    const int64_t K = t.ne0;
    const uint8_t *src = t.data + wtype_row_bytes(t.type, K) * (size_t)r;
    if (t.type == W_F32) {
        const float *f = (const float *)src;
        for (int64_t k = 0; k < K; ++k) dst[k] = (_Float16)f[k];
    } else if (t.type == W_F16) {
        memcpy(dst, src, (size_t)K * 2);
    } else {
        const int bs = t.type == W_Q4_0 ? 18 : 20;
        for (int64_t b = 0; b < K / 32; ++b) {
            const uint8_t *blk = src + b * bs;
            uint16_t dbits; memcpy(&dbits, blk, 2);
            const float d = h2f(dbits);
            float m = 0.f;
            const uint8_t *qs = blk + 2;
            if (t.type == W_Q4_1) { uint16_t mb; memcpy(&mb, blk + 2, 2); m = h2f(mb); qs = blk + 4; }
            for (int j = 0; j < 16; ++j) {
                const int q0 = qs[j] & 0x0F, q1 = qs[j] >> 4;
                if (t.type == W_Q4_0) {
                    dst[b * 32 + j] = (_Float16)((float)(q0 - 8) * d);
                    dst[b * 32 + j + 16] = (_Float16)((float)(q1 - 8) * d);
                } else {
                    dst[b * 32 + j] = (_Float16)((double)q0 * d + m);          // exact in double: one rounding, like an f16 fma
                    dst[b * 32 + j + 16] = (_Float16)((double)q1 * d + m);
                }
            }
        }
    }
}

}  // namespace

bool GemmWeightStore::build(const std::vector<const HostTensor *> &rows, bool want_naive, std::string &err, bool want_kperm,
                            bool expand_q4, bool want_f32) {
    const int64_t K = rows[0]->ne0;
    const int ftype = rows[0]->type;
    int64_t N = 0;
    for (auto *t : rows) {
        if (t->ne0 != K || t->type != ftype) { err = "stacked weights disagree in shape/type"; return false; }
        N += t->ne1;
    }
    w.N = (int)N; w.K = (int)K;
    w.N_pad = (int)((N + GEMM_BN - 1) / GEMM_BN * GEMM_BN);
    mfma_ok = (K % GEMM_BK == 0) && (N % 8 == 0);
    auto src_row = [&](int64_t n, const HostTensor *&t, int64_t &r) {
        for (auto *c : rows) { if (n < c->ne1) { t = c; r = n; return; } n -= c->ne1; }
        t = nullptr; r = 0;
    };
    // expand_q4: the 4-bit blocks become an f16 image here, once (same values the fused-dequant kernels build in
    // registers on every tile); the matrix then runs on the f16 kernels
    const bool quant = (ftype == W_Q4_0 || ftype == W_Q4_1) && !expand_q4;
    if (mfma_ok && !quant) {
        w.type = GW_F16;
        std::vector<_Float16> img((size_t)w.N_pad * K, (_Float16)0);
        for (int64_t n = 0; n < N; ++n) { const HostTensor *t; int64_t r; src_row(n, t, r); row_to_f16(*t, r, img.data() + (size_t)n * K); }
        if (!w16.upload(img.data(), img.size() * 2, err)) return false;
        w.w16 = w16.as<half_t>();
        if (want_kperm && K % 16 == 0) {
            std::vector<_Float16> pimg(img.size());
            for (size_t base = 0; base < img.size(); base += 16)
                for (int j = 0; j < 16; ++j) {
                    // stored position j of a group <- k offset: [0-3, 8-11, 4-7, 12-15]
                    const int src = (j & 3) + ((j >> 2) & 1) * 8 + (j >> 3) * 4;
                    pimg[base + j] = img[base + src];
                }
            if (!w16p.upload(pimg.data(), pimg.size() * 2, err)) return false;
            w.w16p = w16p.as<half_t>();
        }
    } else if (mfma_ok) {
        w.type = ftype == W_Q4_0 ? GW_Q4_0 : GW_Q4_1;
        const int bs = ftype == W_Q4_0 ? 18 : 20, scb = ftype == W_Q4_0 ? 2 : 4;
        const int64_t nkt = K / GEMM_BK, ntn = w.N_pad / GEMM_BN;
        const size_t nblk = (size_t)ntn * nkt * 256;
        std::vector<uint8_t> q(nblk * 16, 0), s(nblk * scb, 0);
        for (int64_t nt = 0; nt < ntn; ++nt)
            for (int64_t kt = 0; kt < nkt; ++kt)
                for (int row = 0; row < 128; ++row) {
                    const int64_t n = nt * 128 + row;
                    if (n >= N) continue;
                    const HostTensor *t; int64_t r; src_row(n, t, r);
                    const uint8_t *rowp = t->data + wtype_row_bytes(ftype, K) * (size_t)r;
                    for (int kb = 0; kb < 2; ++kb) {
                        const uint8_t *blk = rowp + (size_t)(kt * 2 + kb) * bs;
                        const size_t bi = ((size_t)(nt * nkt + kt) * 128 + row) * 2 + kb;
                        memcpy(s.data() + bi * scb, blk, scb);               // d  or  {d, m}
                        memcpy(q.data() + bi * 16, blk + scb, 16);           // 32 nibbles
                    }
                }
        if (!qs.upload(q.data(), q.size(), err) || !sc.upload(s.data(), s.size(), err)) return false;
        w.qs = qs.as<uint4>();
        w.sc = sc.p;
    }
    if (want_f32 && ftype == W_F32) {
        std::vector<uint8_t> all;
        for (auto *t : rows) all.insert(all.end(), t->data, t->data + t->nbytes);
        if (all.size() != (size_t)N * K * 4) { err = "f32 tensor size mismatch"; return false; }
        if (!w32.upload(all.data(), all.size(), err)) return false;
        w.w32 = w32.as<float>();
    }
    if (!mfma_ok || want_naive) {
        std::vector<_Float16> img((size_t)N * K);
        for (int64_t n = 0; n < N; ++n) { const HostTensor *t; int64_t r; src_row(n, t, r); row_to_f16(*t, r, img.data() + (size_t)n * K); }
        if (!naive16.upload(img.data(), img.size() * 2, err)) return false;
        w.naive16 = naive16.as<half_t>();
    }
    return true;
}

bool GemmWeightStore::build_ln_fold(const std::vector<const HostTensor *> &rows, const float *gamma, const float *beta, const float *bias,
                                    DevBuf &waug, std::string &err) {
    const int64_t K = rows[0]->ne0;
    int64_t N = 0;
    for (auto *t : rows) { if (t->ne0 != K) { err = "stacked weights disagree in shape"; return false; } N += t->ne1; }
    w.N = (int)N; w.K = (int)K; w.N_pad = (int)((N + GEMM_BN - 1) / GEMM_BN * GEMM_BN);
    mfma_ok = (K % GEMM_BK == 0) && (N % 8 == 0);
    if (!mfma_ok) return true;                                // (shapes the MFMA kernels do not take are never folded)
    w.type = GW_F16;
    std::vector<_Float16> img((size_t)w.N_pad * K, (_Float16)0), row((size_t)K), aug((size_t)N * 16, (_Float16)0);
    int64_t n = 0;
    for (auto *t : rows)
        for (int64_t r = 0; r < t->ne1; ++r, ++n) {
            row_to_f16(*t, r, row.data());                    // (the values the un-folded f16 image holds)
            double s = 0.0, c = bias ? (double)bias[n] : 0.0;
            for (int64_t k = 0; k < K; ++k) {
                const _Float16 wf = (_Float16)((float)row[k] * gamma[k]);
                img[(size_t)n * K + k] = wf;
                s += (double)(float)wf;
                c += (double)beta[k] * (double)(float)row[k];
            }
            const _Float16 s_h = (_Float16)(float)s, s_l = (_Float16)(float)(s - (double)(float)s_h);
            const _Float16 c_h = (_Float16)(float)c, c_l = (_Float16)(float)(c - (double)(float)c_h);
            _Float16 *a = aug.data() + (size_t)n * 16;
            a[0] = s_h; a[1] = s_l; a[2] = s_h; a[3] = c_h; a[4] = c_l; a[5] = c_h;
        }
    if (!w16.upload(img.data(), img.size() * 2, err) || !waug.upload(aug.data(), aug.size() * 2, err)) return false;
    w.w16 = w16.as<half_t>();
    return true;
}

// packed (f16 gamma | f16 (beta + bias) << 16) per feature: what a residual mat-mul needs to rebuild LayerNorm(resid) per element
static bool upload_gamma_beta_bias(DevBuf &b, const float *gamma, const float *beta, const float *bias, int64_t n, std::string &err) {
    std::vector<uint32_t> v((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const _Float16 g = (_Float16)gamma[i], bb = (_Float16)(beta[i] + bias[i]);
        uint16_t gu, bu;
        memcpy(&gu, &g, 2); memcpy(&bu, &bb, 2);
        v[(size_t)i] = (uint32_t)gu | ((uint32_t)bu << 16);
    }
    return b.upload(v.data(), v.size() * 4, err);
}

// ------------------------------------------------------------------------------------------------
// Engine
// ------------------------------------------------------------------------------------------------
static bool upload_f32(DevBuf &b, const HostTensor *t, std::string &err) { return b.upload(t->data, t->nbytes, err); }

// an embedding table of a q4 file as f32 values: (q - 8) d / q d + m, the numbers the gather kernel dequantises on the fly
// (q d is exact in f32, so the host's multiply-add and the device's fma round alike); 6.4x the bytes, but the f32 form is
// read by the 16-byte-run kernel (embed_ln_rows_kernel) instead of element by element
static bool upload_table_f32(DevBuf &b, const HostTensor *t, std::string &err) {
    const int64_t K = t->ne0, N = t->ne1;
    const int bs = t->type == W_Q4_0 ? 18 : 20;
    std::vector<float> img((size_t)N * K);
    for (int64_t r = 0; r < N; ++r) {
        const uint8_t *src = t->data + wtype_row_bytes(t->type, K) * (size_t)r;
        float *dst = img.data() + (size_t)r * K;
        for (int64_t blk_i = 0; blk_i < K / 32; ++blk_i) {
            const uint8_t *blk = src + blk_i * bs;
            uint16_t dbits; memcpy(&dbits, blk, 2);
            const float d = h2f(dbits);
            float m = 0.f;
            const uint8_t *qs = blk + 2;
            if (t->type == W_Q4_1) { uint16_t mb; memcpy(&mb, blk + 2, 2); m = h2f(mb); qs = blk + 4; }
            for (int j = 0; j < 16; ++j) {
                const int q0 = qs[j] & 0x0F, q1 = qs[j] >> 4;
                dst[blk_i * 32 + j] = t->type == W_Q4_0 ? (float)(q0 - 8) * d : (float)q0 * d + m;
                dst[blk_i * 32 + j + 16] = t->type == W_Q4_0 ? (float)(q1 - 8) * d : (float)q1 * d + m;
            }
        }
    }
    return b.upload(img.data(), img.size() * sizeof(float), err);
}

static bool concat_upload(DevBuf &b, std::initializer_list<const HostTensor *> ts, std::string &err) {
    std::vector<uint8_t> all;
    for (auto *t : ts) all.insert(all.end(), t->data, t->data + t->nbytes);
    return b.upload(all.data(), all.size(), err);
}

Engine *Engine::create(const ModelFile &mf, int device, std::string &err) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        err = "no HIP device available (this library needs an AMD GPU; there is no CPU fallback)";
        return nullptr;
    }
    if (device < 0 || device >= ndev) { err = "HIP device ordinal " + std::to_string(device) + " out of range"; return nullptr; }
    Engine *e = new Engine;
    e->hp_ = mf.hp;
    e->device_ = device;
    if (hipSetDevice(e->device_) != hipSuccess) { err = "hipSetDevice failed"; delete e; return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, e->device_) != hipSuccess) { err = "hipGetDeviceProperties failed"; delete e; return nullptr; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        err = std::string("unsupported GPU architecture '") + prop.gcnArchName + "' (kernels are built for gfx950 / MI355X only)";
        delete e; return nullptr;
    }
    // BERT_HIP_KERNELS = fused (default) | tiled (GEMM + attention + LayerNorm kernels, Q|K|V and the intermediate through HBM);
    // finer switches: bert_hip_set_option.  The GENERIC kernels (any shape, row-major f16 images) are what shapes outside the MFMA
    // kernels' reach fall back to; running a whole model on them is a cross-check for the tests, not a route of the product:
    // "naive" is understood by libbert_test.so only (this file compiled with -DBERT_HIP_TEST_ROUTES).
    if (const char *k = getenv("BERT_HIP_KERNELS")) {
#ifdef BERT_HIP_TEST_ROUTES
        if (strcmp(k, "naive") == 0) e->gemm_naive_ = e->attn_naive_ = true;
        else
#else
        if (strcmp(k, "naive") == 0) fprintf(stderr, "BERT_HIP_KERNELS=naive: a test cross-check (libbert_test.so), not a route of libbert.so; ignored\n");
        else
#endif
        if (strcmp(k, "tiled") == 0) e->qkv2_ = e->tail_ = e->latency_ = false, e->one_launch_ = 0;
    }
    // (the cap of the latency route: measured on H = 384; a window of an H = 128 model costs the fused kernels less than five
    // launches cost the route, so such models keep the one-window cap)
    if (mf.hp.n_embd < 256) e->latency_tokens_ = 128;
    // BERT_HIP_LATENCY: 0 = no latency route; 1 = the default cap; n >= 32: calls of at most n tokens take it
    if (const char *f = getenv("BERT_HIP_LATENCY")) {
        e->latency_ = strcmp(f, "0") != 0;
        if (atoi(f) >= 32) e->latency_tokens_ = atoi(f);
    }
    if (const char *f = getenv("BERT_HIP_Q4")) e->q4_expand_ = strcmp(f, "fused") != 0;
    if (const char *f = getenv("BERT_HIP_LN_FOLD")) e->ln_fold_ = strcmp(f, "0") != 0;        // (tuning: 0 = LayerNorm kernels of their own at H = 768)
    if (const char *c = getenv("BERT_HIP_CHUNK_TOKENS")) { const int v = atoi(c); if (v > 0) e->chunk_tokens_ = v; }
    if (const char *c = getenv("BERT_HIP_WINDOW_SLOTS")) set_window_slots(atoi(c));
    // f32 files: f32 arithmetic like the reference's (f32_route.hip) unless BERT_HIP_F32=f16 asks for f16 operands and the fused kernels
    if (const char *f = getenv("BERT_HIP_F32")) e->f32_exact_ = strcmp(f, "f16") != 0;
    if (mf.hp.n_embd % 2 != 0) { err = "n_embd must be even"; delete e; return nullptr; }

    auto T = [&](const std::string &n) { return mf.find(n); };
    bool ok = true;
    e->table_type_ = mf.hp.f16;
    {
        const HostTensor *tw = T("embeddings.word_embeddings.weight"), *tt = T("embeddings.token_type_embeddings.weight"),
                         *tp = T("embeddings.position_embeddings.weight");
        const bool q4_tables = tw && tt && tp && (tw->type == W_Q4_0 || tw->type == W_Q4_1) && tt->type == tw->type && tp->type == tw->type;
        if (q4_tables && e->q4_expand_ && mf.hp.n_embd % 32 == 0) {
            e->table_type_ = 0;
            ok = ok && upload_table_f32(e->word_emb_, tw, err) && upload_table_f32(e->type_emb_, tt, err) && upload_table_f32(e->pos_emb_, tp, err);
        } else {
            ok = ok && upload_f32(e->word_emb_, tw, err) && upload_f32(e->type_emb_, tt, err) && upload_f32(e->pos_emb_, tp, err);
        }
    }
    ok = ok && upload_f32(e->ln_e_w_, T("embeddings.LayerNorm.weight"), err);
    ok = ok && upload_f32(e->ln_e_b_, T("embeddings.LayerNorm.bias"), err);
    const bool want_naive = e->gemm_naive_;
    // the f32 route needs every matrix of every layer and the three tables as f32 tensors (a file has one ftype, but check)
    e->f32_file_ = mf.hp.f16 == 0;
    for (const char *n : {"embeddings.word_embeddings.weight", "embeddings.token_type_embeddings.weight", "embeddings.position_embeddings.weight"})
        e->f32_file_ = e->f32_file_ && T(n) && T(n)->type == W_F32;
    for (int i = 0; e->f32_file_ && i < mf.hp.n_layer; ++i) {
        const std::string p = "encoder.layer." + std::to_string(i) + ".";
        for (const char *n : {"attention.self.query.weight", "attention.self.key.weight", "attention.self.value.weight", "attention.output.dense.weight",
                              "intermediate.dense.weight", "output.dense.weight"})
            e->f32_file_ = e->f32_file_ && T(p + n) && T(p + n)->type == W_F32;
    }
    const bool want_f32 = e->f32_file_;
    // the k-permuted second image of the FFN weights is only read by layer_tail_kernel (H = 256 / 384)
    const bool want_kperm = mf.hp.n_embd % 128 == 0 && mf.hp.n_embd >= 256 && mf.hp.n_embd <= 384;
    for (int i = 0; ok && i < mf.hp.n_layer; ++i) {
        const std::string p = "encoder.layer." + std::to_string(i) + ".";
        auto *L = new LayerWeights;
        e->layers_.push_back(L);
        ok = ok && L->qkv.build({T(p + "attention.self.query.weight"), T(p + "attention.self.key.weight"),
                                 T(p + "attention.self.value.weight")}, want_naive, err, false, e->q4_expand_, want_f32);
        {
            const HostTensor *tq = T(p + "attention.self.query.weight");
            const bool q4_file = tq && (tq->type == W_Q4_0 || tq->type == W_Q4_1);
            if (ok && q4_file && e->q4_expand_ && L->qkv.mfma_ok && (size_t)L->qkv.w.N * L->qkv.w.K * 2 > ((size_t)3 << 20))
                ok = L->qkv_q4.build({tq, T(p + "attention.self.key.weight"), T(p + "attention.self.value.weight")}, false, err, false, false);
        }
        ok = ok && concat_upload(L->qkv_b, {T(p + "attention.self.query.bias"), T(p + "attention.self.key.bias"),
                                            T(p + "attention.self.value.bias")}, err);
        ok = ok && L->o.build({T(p + "attention.output.dense.weight")}, want_naive, err, false, e->q4_expand_, want_f32);
        ok = ok && upload_f32(L->o_b, T(p + "attention.output.dense.bias"), err);
        ok = ok && upload_f32(L->ln_att_w, T(p + "attention.output.LayerNorm.weight"), err);
        ok = ok && upload_f32(L->ln_att_b, T(p + "attention.output.LayerNorm.bias"), err);
        ok = ok && L->ffi.build({T(p + "intermediate.dense.weight")}, want_naive, err, want_kperm, e->q4_expand_, want_f32);
        ok = ok && upload_f32(L->ffi_b, T(p + "intermediate.dense.bias"), err);
        ok = ok && L->ffo.build({T(p + "output.dense.weight")}, want_naive, err, want_kperm, e->q4_expand_, want_f32);
        ok = ok && upload_f32(L->ffo_b, T(p + "output.dense.bias"), err);
        ok = ok && upload_f32(L->ln_out_w, T(p + "output.LayerNorm.weight"), err);
        ok = ok && upload_f32(L->ln_out_b, T(p + "output.LayerNorm.bias"), err);
    }
    // LayerNorm folding (kernels.h GemmLnFold; the route of models the fused H <= 384 kernels do not take): images for every layer
    // whose four matrices run on gemm256's f16 form
    for (int i = 0; ok && i < mf.hp.n_layer; ++i) {
        LayerWeights &L = *e->layers_[i];
        const std::string p = "encoder.layer." + std::to_string(i) + ".";
        const int H = mf.hp.n_embd;
        auto f16_256 = [](const GemmWeightStore &s) { return s.mfma_ok && s.w.type == GW_F16 && s.w.N % 256 == 0 && s.w.K % 64 == 0 && s.w.K >= 128; };
        if (!(H > 384 && H % 256 == 0 && f16_256(L.qkv) && f16_256(L.o) && f16_256(L.ffi) && f16_256(L.ffo)) || e->f32_file_) continue;
        auto F = [&](const std::string &n) { const HostTensor *t = T(n); return t ? (const float *)t->data : nullptr; };
        const float *g1 = F(p + "attention.output.LayerNorm.weight"), *b1 = F(p + "attention.output.LayerNorm.bias");
        const float *bi = F(p + "intermediate.dense.bias"), *bo2 = F(p + "output.dense.bias"), *bo = F(p + "attention.output.dense.bias");
        if (!g1 || !b1 || !bi || !bo2 || !bo) continue;
        ok = L.ffi_fold.build_ln_fold({T(p + "intermediate.dense.weight")}, g1, b1, bi, L.ffi_waug, err) &&
             upload_gamma_beta_bias(L.ffo_gb, g1, b1, bo2, H, err);
        if (ok && i >= 1) {
            const std::string q = "encoder.layer." + std::to_string(i - 1) + ".";
            const float *g2 = F(q + "output.LayerNorm.weight"), *b2 = F(q + "output.LayerNorm.bias");
            std::vector<float> qb((size_t)3 * H);
            const char *names[3] = {"attention.self.query.bias", "attention.self.key.bias", "attention.self.value.bias"};
            bool have = g2 && b2;
            for (int k = 0; have && k < 3; ++k) { const float *b = F(p + names[k]); have = b != nullptr; if (have) memcpy(qb.data() + (size_t)k * H, b, (size_t)H * 4); }
            if (!have) continue;
            ok = L.qkv_fold.build_ln_fold({T(p + "attention.self.query.weight"), T(p + "attention.self.key.weight"), T(p + "attention.self.value.weight")},
                                          g2, b2, qb.data(), L.qkv_waug, err) &&
                 upload_gamma_beta_bias(L.o_gb, g2, b2, bo, H, err);
        }
        L.fold_ok = ok && L.ffi_fold.mfma_ok && (i == 0 || L.qkv_fold.mfma_ok);
    }
    ok = ok && e->status_.alloc(16, err);
    if (ok && hipStreamCreateWithFlags(&e->stream_, hipStreamNonBlocking) != hipSuccess) { err = "hipStreamCreate failed"; ok = false; }
    if (ok && hipEventCreateWithFlags(&e->busy_, hipEventDisableTiming) != hipSuccess) { err = "hipEventCreate failed"; ok = false; }
    if (!ok) { delete e; return nullptr; }
    return e;
}

Engine::~Engine() {
    (void)hipSetDevice(device_);
    (void)hipDeviceSynchronize();
    for (auto *L : layers_) delete L;
    for (auto ev : ev_pool_) (void)hipEventDestroy(ev);
    for (auto &p : pending_) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto &sl : slot_) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (busy_) (void)hipEventDestroy(busy_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

bool Engine::reserve(int n_tokens, int n_sentences, std::string &err) {
    HIP_OK(hipSetDevice(device_), err, false);
    if (n_tokens <= 0 || n_sentences <= 0) return true;
    return ensure_workspace((n_tokens + 255) / 256 * 256, n_sentences, err);
}

int Engine::check(std::string &err) {
    HIP_OK(hipSetDevice(device_), err, -1);
    HIP_OK(hipDeviceSynchronize(), err, -1);
    int st = 0;
    HIP_OK(hipMemcpy(&st, status_.p, sizeof(int), hipMemcpyDeviceToHost), err, -1);
    if (st) HIP_OK(hipMemset(status_.p, 0, sizeof(int)), err, -1);
    return st;
}

void Engine::set_option(const std::string &key, const std::string &value) {
#ifndef BERT_HIP_TEST_ROUTES
    if ((key == "gemm" || key == "attn") && value == "naive") {
        fprintf(stderr, "bert_hip_set_option: %s=naive is a test cross-check (libbert_test.so), not a route of libbert.so; ignored\n", key.c_str());
        return;
    }
#endif
    if (key == "gemm") {
        // the generic kernel reads GemmWeight::naive16, an image that is only built at load time (BERT_HIP_KERNELS=naive) or
        // for shapes the MFMA kernels cannot take: refuse the switch when a matrix lacks it
        bool have = true;
        for (auto *L : layers_)
            for (GemmWeightStore *w : {&L->qkv, &L->o, &L->ffi, &L->ffo}) have = have && w->w.naive16 != nullptr;
        if (value == "naive" && !have)
            fprintf(stderr, "bert_hip_set_option: gemm=naive needs BERT_HIP_KERNELS=naive at load time (the f16 row-major images were not built); ignored\n");
        else gemm_naive_ = value == "naive";
    } else if (key == "attn") attn_naive_ = value == "naive";
    else if (key == "qkv2") qkv2_ = value != "0";
    else if (key == "gemm256") gemm256_ = value != "0";
    else if (key == "ln_fold") ln_fold_ = value != "0";
    else if (key == "tail") tail_ = value != "0";
    else if (key == "latency") latency_ = value != "0";
    else if (key == "stage_kernel") stage_kernel_ = value != "0";
    else if (key == "window_slots") set_window_slots(atoi(value.c_str()));      // (process-wide: 16, or 8 — see kernels.h)
    else if (key == "latency_tokens") { const int v = atoi(value.c_str()); if (v >= 32) latency_tokens_ = v; }
    else if (key == "one_launch") one_launch_ = value == "0" ? 0 : value == "2" ? 2 : 1;
    else if (key == "f32") f32_exact_ = value != "f16";       // f32 files: "exact" (f32 arithmetic, default) | "f16" (f16 operands, fused kernels)
    else if (key == "chunk_tokens") { const int v = atoi(value.c_str()); if (v > 0) chunk_tokens_ = v; }
    else if (key == "profile_replay") {
        // "<kernel name>:<K>" (see timed()), "" switches back to an event pair per launch
        const size_t c = value.rfind(':');
        replay_name_ = c == std::string::npos ? value : value.substr(0, c);
        replay_k_ = c == std::string::npos ? 10 : std::max(1, atoi(value.c_str() + c + 1));
    }
}

bool Engine::ensure_workspace(int t_pad, int n_sentences, std::string &err) {
    const size_t H = hp_.n_embd, I = hp_.n_intermediate, tp = (size_t)t_pad;
    const size_t es = f32_file_ ? 4 : 2;                      // (f32 files: the f32 route's activations are f32)
    return x_.ensure(tp * H * es, err) && qkv_.ensure(tp * 3 * H * es, err) && ctx_.ensure(tp * H * es, err) &&
           y_.ensure(tp * H * es, err) && ff_.ensure(tp * I * es, err) && v32_.ensure((size_t)std::max(128, std::min(t_pad, (latency_tokens_ + 255) / 256 * 256)) * H * 4, err) &&
           d_out_.ensure((size_t)n_sentences * H * 4, err) &&
           windows_.ensure((size_t)n_sentences * sizeof(int2), err) &&
           // (LayerNorm folding, H = 768 route: 2 H / 256 partial (sum, sum of squares) pairs and one finalized float4 per row and LayerNorm)
           (!(H > 384 && H % 256 == 0) || (ln_stats1_.ensure(tp * (2 * H / 256) * 8, err) && ln_stats2_.ensure(tp * (2 * H / 256) * 8, err) &&
                                           ln_rows1_.ensure(tp * 16, err) && ln_rows2_.ensure(tp * 16, err)));
}

template <class F>
void Engine::timed(const char *name, double flops, hipStream_t s, F &&f) {
    if (!profiling_) { f(); return; }
    auto get = [&]() {
        hipEvent_t ev;
        if (!ev_pool_.empty()) { ev = ev_pool_.back(); ev_pool_.pop_back(); }
        else (void)hipEventCreate(&ev);
        return ev;
    };
    if (!replay_name_.empty()) {
        // replay form ("profile_replay" = "<kernel>:<K>"): the pass runs untimed; behind the FIRST launch of the named kernel
        // the same launch is repeated K times between ONE event pair — the pair's own cost (tens of microseconds around a
        // sub-millisecond kernel) is spread over K launches, so launches x average cannot exceed the step they belong to.
        // Kernels that work in place see their own output as input in the repeats: the pass's results are not to be used.
        f();
        if (replay_done_ || replay_name_ != name) return;
        replay_done_ = true;
        // (in-place kernels run on their own output from here on: this pass's embeddings are NOT results — bench.py restores
        // its output buffer; say so once for anybody else who turns the option on)
        static bool warned = false;
        if (!warned && !getenv("BERT_HIP_QUIET")) { warned = true; fprintf(stderr, "bert_hip: profile_replay is active: the embeddings of profiled passes are not valid results\n"); }
        Pending p{name, get(), get(), flops * replay_k_, replay_k_};
        (void)hipEventRecord(p.a, s);
        for (int k = 0; k < replay_k_; ++k) f();
        (void)hipEventRecord(p.b, s);
        pending_.push_back(p);
        return;
    }
    // an event pair attached to the dispatch itself (kernels.h BERT_LAUNCH): an upper bound of the kernel's time in the pass
    Pending p{name, get(), get(), flops, 1};
    LaunchTiming lt{p.a, p.b, 0};
    tl_launch_timing = &lt;
    f();
    tl_launch_timing = nullptr;
    // exactly one launch carries the pair; anything else (a launcher that returned early: stale timestamps of pooled events;
    // several launches: only the last one measured) is not a sample
    if (lt.launches == 1) pending_.push_back(p);
    else { ev_pool_.push_back(p.a); ev_pool_.push_back(p.b); }
}

void Engine::profile_enable(bool on) { profiling_ = on; }

std::string Engine::profile_report() {
    (void)hipSetDevice(device_);
    (void)hipDeviceSynchronize();
    for (auto &p : pending_) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            KernelStat &st = stats_[p.name];
            st.launches += p.launches; st.ms += ms; st.flops += p.flops;
        }
        ev_pool_.push_back(p.a); ev_pool_.push_back(p.b);
    }
    pending_.clear();
    std::string out;
    char line[256];
    for (auto &kv : stats_) {
        snprintf(line, sizeof(line), "%s %d %.6f %.6e\n", kv.first.c_str(), kv.second.launches, kv.second.ms,
                 kv.second.launches ? kv.second.flops / kv.second.launches : 0.0);
        out += line;
    }
    stats_.clear();
    for (auto &kv : families_) {
        snprintf(line, sizeof(line), "%s %d 0 0\n", kv.first.c_str(), kv.second);
        out += line;
    }
    families_.clear();
    return out;
}

void Engine::build_windows(const int32_t *cu, int B, std::vector<int2> &windows, int slot) {
    windows.clear();                                          // slot: 16 (or 8: kernels.h), the value the pass read once
    int first = 0, fill = 0;                                  // open window: sentences first .. b-1 occupy `fill` slots
    for (int b = 0; b < B; ++b) {
        const int n = cu[b + 1] - cu[b];
        if (b > first && fill + n > 128) {
            windows.push_back(make_int2(first, b - first));
            first = b; fill = 0;
        }
        fill = (fill + n + slot - 1) & ~(slot - 1);
    }
    if (B > first) windows.push_back(make_int2(first, B - first));
}

int Engine::eval_packed_device(const int32_t *d_tokens, const int32_t *d_cu, int B, int T, int max_len, float *d_out,
                               hipStream_t s, float *d_hidden, std::string &err, const int2 *d_windows, int n_windows, int slots_in) {
    if (B <= 0 || T <= 0) return 0;
    // the windows' place granularity, read ONCE per pass (the host path read it when it built its list): the window list, the
    // grid bound and the kernels' place rule must agree whatever another thread or context sets meanwhile
    const int slots = slots_in ? slots_in : window_slots();
    HIP_OK(hipSetDevice(device_), err, -1);
    const int H = hp_.n_embd, I = hp_.n_intermediate, nh = hp_.n_head, dh = H / nh;
    const int t_pad = (T + 255) / 256 * 256;                 // whole tiles of every kernel family (128- and 256-token tiles)
    if (!ensure_workspace(t_pad, B, err)) return -1;
    replay_done_ = false;
    // one forward pass at a time on the shared workspace: wait (on the caller's stream) for the previous pass
    HIP_OK(hipStreamWaitEvent(s, busy_, 0), err, -1);
    if (f32_file_ && f32_exact_) return forward_f32(d_tokens, d_cu, B, T, max_len, d_out, s, d_hidden, err);
    half_t *x = x_.as<half_t>(), *qkv = qkv_.as<half_t>(), *ctx = ctx_.as<half_t>(), *y = y_.as<half_t>(),
           *ff = ff_.as<half_t>();
    const double Td = (double)T;

    auto gemm = [&](const char *name, GemmWeightStore &W, const half_t *A, const float *bias, const half_t *resid,
                    half_t *C, int epi, const GemmLnFold *ln = nullptr) {
        const bool big = W.mfma_ok && gemm256_ && !gemm_naive_ && gemm256_supported(W.w, t_pad);
        const bool tiled = !big && W.mfma_ok && (!gemm_naive_ || !W.w.naive16);
        // (which kernel family served the mat-mul: reported as "family:<kernel>_<weights>" lines of the profile)
        if (profiling_ && replay_name_.empty())
            families_[std::string("family:") + (big ? "gemm256" : tiled ? "gemm_mfma" : "gemm_naive") + (big || tiled ? (W.w.type == GW_F16 ? "_f16" : "_q4") : "")] += 1;
        timed(name, 2.0 * Td * W.w.N * W.w.K, s, [&] {
            if (big) launch_gemm256(W.w, A, bias, resid, C, t_pad, epi, s, ln);
            else if (tiled) launch_gemm_mfma(W.w, A, bias, resid, C, t_pad, epi, s);
            else launch_gemm_naive(W.w, A, bias, resid, C, T, epi, s);
        });
    };
    auto tap = [&](int idx) {
        if (d_hidden) launch_f16_to_f32(x, d_hidden + (size_t)idx * T * H, (size_t)T * H, s);
    };

    timed("embed_ln", 0.0, s, [&] {
        launch_embed_ln(word_emb_.p, type_emb_.p, pos_emb_.p, table_type_, ln_e_w_.as<float>(), ln_e_b_.as<float>(),
                        d_tokens, d_cu, B, T, H, hp_.n_vocab, max_len, x, s);
    });
    tap(0);
    // attention FLOPs: 4 * sum_b N_b^2 * H; only T and max_len are known here -> upper bound T * max_len
    const double att_flops = 4.0 * Td * max_len * H;
    // sentence windows of the fused projection+attention kernel: the caller's (host path), or built here on the device from
    // cu_seqlens when packing can pay — sentences on average clearly shorter than max_len; for full-length batches the
    // uniform rule (max_len-sized places) gives the same windows without the extra launch
    const int *d_n_windows = nullptr;
    const bool fused_windows = qkv2_ && !gemm_naive_ && !attn_naive_ && layers_[0]->qkv.mfma_ok && qkv_attention2_supported(layers_[0]->qkv.w, nh, dh, max_len);
    // all layers in one launch (model_kernel.hip): a workgroup carries its window through every layer
    // (every layer's matrices are checked: a file may mix types or shapes from layer to layer, and the kernel takes all layers' pointers)
    bool one_launch_ok = fused_windows && one_launch_ && tail_ && !d_hidden && !(latency_ && T <= latency_tokens_);
    for (int il = 0; one_launch_ok && il < hp_.n_layer; ++il) {
        LayerWeights &L = *layers_[il];
        one_launch_ok = L.qkv.mfma_ok && L.o.mfma_ok && L.ffi.mfma_ok && L.ffo.mfma_ok && L.ffi.w.w16p && L.ffo.w.w16p &&
                        model_kernel_supported(L.qkv.w, L.o.w, L.ffi.w, L.ffo.w, hp_.n_layer, nh, dh, max_len);
    }
    const bool full_windows = (long long)B * 128 == T;
    if (!d_windows && fused_windows) {
        const int spw = qkv_attention2_sentences_per_window(max_len, slots), uniform = (B + spw - 1) / spw;
        // (forced one-launch: the kernel takes a window list or one sentence per window — its layer-tail phase needs a window's
        // tokens to be at most 128 whatever the sentences' lengths turn out to be)
        if (4ll * uniform * 128 > 5 * ((long long)T + (long long)(slots / 2) * B) || (one_launch_ok && one_launch_ == 2 && spw > 1 && !full_windows)) {
            int *count = status_.as<int>() + 1;
            timed("build_windows", 0.0, s, [&] { launch_build_windows(d_cu, B, windows_.as<int2>(), count, slots, s); });
            d_windows = windows_.as<int2>();
            d_n_windows = count;
            // upper bound from T and B alone (the extra workgroups return at once): "never more than the uniform rule" only
            // holds for batches that keep their max_len promise, and a broken promise must cost the offender its row, not
            // a neighbour its window
            n_windows = qkv_attention2_max_windows(B, T, slots);
        }
    }
    // When it pays: the layer-tail phase costs a window 128 rows' time however few tokens it holds, the layer-tail KERNEL runs
    // on the packed tokens — 0.32 + 0.68 fill against 0.94 (full windows: +6.7 %): from a fill of 0.91.  The window count is
    // known for the caller's list and for one sentence per window, not for a list built on the device.
    bool one_launch_pays = full_windows || one_launch_ == 2;
    if (!one_launch_pays && !d_n_windows) {
        const long long n_win = d_windows ? n_windows : B;
        one_launch_pays = (d_windows || qkv_attention2_sentences_per_window(max_len, slots) == 1) && 100ll * T >= 95ll * 128 * n_win;
    }
    // The latency route (skinny.hip): at most 128 tokens = one window of the fused kernels, which would keep one CU of 256 busy
    // per launch.  Same bits per sentence (the route must not show in the results), seven short launches per layer.
    const bool skinny = latency_ && tail_ && qkv2_ && !gemm_naive_ && !attn_naive_ && T <= latency_tokens_ && max_len <= 128 && (dh == 32 || dh == 64) &&
                        skinny_layer_supported(layers_[0]->qkv.w, layers_[0]->o.w, layers_[0]->ffi.w, layers_[0]->ffo.w) &&
                        qkv_attention2_supported(layers_[0]->qkv.w, nh, dh, max_len);
    if (skinny) {
        const int tb = (T + 31) / 32, Lz = hp_.n_layer;
        float *v32 = v32_.as<float>();
        for (int il = 0; il < Lz; ++il) {
            LayerWeights &L = *layers_[il];
            // (from the second layer on the QKV kernel LayerNorms the previous layer's output itself and writes x)
            LayerWeights *P = il ? layers_[il - 1] : nullptr;
            timed("skinny_qkv", 2.0 * Td * 3 * H * H, s, [&] {
                launch_skinny_gemm(0, L.qkv.w, x, P ? v32 : nullptr, P ? P->ln_out_w.as<float>() : nullptr, P ? P->ln_out_b.as<float>() : nullptr,
                                   x, L.qkv_b.as<float>(), nullptr, qkv, nullptr, tb, s);
            });
            timed("attention", att_flops, s, [&] { (void)launch_attention_mfma(qkv, d_cu, B, nh, dh, max_len, ctx, s); });
            timed("skinny_proj", 2.0 * Td * H * H, s, [&] {
                launch_skinny_gemm(1, L.o.w, ctx, nullptr, nullptr, nullptr, nullptr, L.o_b.as<float>(), x, nullptr, v32, tb, s);
            });
            timed("skinny_ffn_up", 2.0 * Td * H * I, s, [&] {
                launch_skinny_gemm(2, L.ffi.w, nullptr, v32, L.ln_att_w.as<float>(), L.ln_att_b.as<float>(), y, L.ffi_b.as<float>(), nullptr, ff, nullptr, tb, s);
            });
            timed("skinny_ffn_down", 2.0 * Td * H * I, s, [&] {
                launch_skinny_gemm(3, L.ffo.w, ff, nullptr, nullptr, nullptr, nullptr, L.ffo_b.as<float>(), y, nullptr, v32, tb, s);
            });
            if (il + 1 == Lz || d_hidden) {
                // (the last layer, or a hidden-state tap: somebody has to materialise x now; the next QKV kernel writes the same bits again)
                timed("skinny_layernorm", 0.0, s, [&] { launch_skinny_layernorm(v32, L.ln_out_w.as<float>(), L.ln_out_b.as<float>(), x, tb, H, s); });
            }
            tap(il + 1);
        }
    }
    // All layers in one launch, a workgroup per window (model_kernel.hip) — the two fused kernels' bodies as phases, no kernel
    // boundary to put the workgroups back in step.  Batches of FULL windows (every sentence exactly 128 tokens: T = 128 B) take the
    // specialised form (every window is one whole sentence, whatever list the caller built).
    const bool one_launch = !skinny && one_launch_ok && one_launch_pays;
    if (one_launch) {
        ModelLayerWeights mw[16];
        for (int il = 0; il < hp_.n_layer; ++il) {
            LayerWeights &L = *layers_[il];
            mw[il] = {&L.qkv.w, &L.o.w, &L.ffi.w, &L.ffo.w, L.qkv_b.as<float>(), L.o_b.as<float>(), L.ln_att_w.as<float>(), L.ln_att_b.as<float>(),
                      L.ffi_b.as<float>(), L.ffo_b.as<float>(), L.ln_out_w.as<float>(), L.ln_out_b.as<float>()};
        }
        timed("model_kernel", hp_.n_layer * (2.0 * Td * 3 * H * H + att_flops + 2.0 * Td * H * H + 4.0 * Td * H * I), s, [&] {
            launch_model_kernel(mw, hp_.n_layer, x, ctx, d_cu, B, T, d_windows, n_windows, d_n_windows, nh, d_out, max_len, status_.as<int>(), slots, s);
        });
    }
    // LayerNorm folding (kernels.h GemmLnFold): models whose layers run as gemm256 mat-muls on f16 images (H = 768) keep the
    // UN-normalised sums u1 (in y) and u2 (in x) and never launch a LayerNorm of their own but the last one; a hidden-state tap
    // wants the normalised states and takes the plain sequence
    bool fold = ln_fold_ && gemm256_ && !gemm_naive_ && !skinny && !one_launch && !d_hidden && H > 384 && H % 256 == 0 && ln_rows2_.p;
    for (int il = 0; fold && il < hp_.n_layer; ++il) {
        LayerWeights &L = *layers_[il];
        fold = L.fold_ok && gemm256_supported(L.o.w, t_pad) && gemm256_supported(L.ffo.w, t_pad) && gemm256_supported(L.ffi_fold.w, t_pad) &&
               gemm256_supported(L.qkv.w, t_pad) && !(qkv2_ && qkv_attention2_supported(L.qkv.w, nh, dh, max_len)) &&
               !(tail_ && layer_tail_supported(L.o.w, L.ffi.w, L.ffo.w));
    }
    for (int il = 0; fold && il < hp_.n_layer; ++il) {
        LayerWeights &L = *layers_[il];
        const int P = 2 * H / 256;
        float2 *st1 = ln_stats1_.as<float2>(), *st2 = ln_stats2_.as<float2>();
        float4 *rows1 = ln_rows1_.as<float4>(), *rows2 = ln_rows2_.as<float4>();
        GemmLnFold ln;
        if (il == 0) {
            // x = LayerNorm(embeddings), materialised by the embedding kernel: the plain projection (4-bit planes where the file has them)
            const bool planes = L.qkv_q4.w.qs && L.qkv_q4.mfma_ok && gemm256_supported(L.qkv_q4.w, t_pad);
            gemm("gemm_qkv", planes ? L.qkv_q4 : L.qkv, x, L.qkv_b.as<float>(), nullptr, qkv, EPI_BIAS);
        } else {
            // x holds u2 of the layer before: its output LayerNorm rides in the folded weights, the statistics k-step and the row scale
            ln = GemmLnFold(); ln.flags = GemmLnFold::IN; ln.rows_in = rows2; ln.waug = L.qkv_waug.as<half_t>();
            gemm("gemm_qkv", L.qkv_fold, x, nullptr, nullptr, qkv, EPI_BIAS, &ln);
        }
        timed("attention", att_flops, s, [&] {
            if (attn_naive_ || !launch_attention_mfma(qkv, d_cu, B, nh, dh, max_len, ctx, s))
                launch_attention_naive(qkv, d_cu, B, nh, dh, max_len, ctx, s);
        });
        // u1 = ctx Wo^T + bo + (x | LayerNorm(u2 of the layer before)) -> y, with its rows' partial statistics
        ln = GemmLnFold(); ln.flags = GemmLnFold::STATS | (il ? GemmLnFold::RES : 0); ln.stats = st1;
        ln.rows_res = rows2; ln.gb = L.o_gb.as<unsigned>();
        gemm("gemm_attn_out", L.o, ctx, L.o_b.as<float>(), x, y, EPI_BIAS_RESID, &ln);
        timed("ln_rows_finalize", 0.0, s, [&] { launch_ln_rows_finalize(st1, P, t_pad, H, rows1, s); });
        ln = GemmLnFold(); ln.flags = GemmLnFold::IN; ln.rows_in = rows1; ln.waug = L.ffi_waug.as<half_t>();
        gemm("gemm_ffn_up", L.ffi_fold, y, nullptr, nullptr, ff, EPI_BIAS_GELU, &ln);
        // u2 = ff W2^T + b2 + LayerNorm(u1) -> x
        ln = GemmLnFold(); ln.flags = GemmLnFold::STATS | GemmLnFold::RES; ln.stats = st2; ln.rows_res = rows1; ln.gb = L.ffo_gb.as<unsigned>();
        gemm("gemm_ffn_down", L.ffo, ff, L.ffo_b.as<float>(), y, x, EPI_BIAS_RESID, &ln);
        timed("ln_rows_finalize", 0.0, s, [&] { launch_ln_rows_finalize(st2, P, t_pad, H, rows2, s); });
        if (il + 1 == hp_.n_layer)     // (the pooling reads normalised rows: the one LayerNorm launch of the pass)
            timed("layernorm", 0.0, s, [&] { launch_layernorm(x, L.ln_out_w.as<float>(), L.ln_out_b.as<float>(), T, H, s); });
    }
    for (int il = 0; !fold && !skinny && !one_launch && il < hp_.n_layer; ++il) {
        LayerWeights &L = *layers_[il];
        if (qkv2_ && !gemm_naive_ && !attn_naive_ && L.qkv.mfma_ok && qkv_attention2_supported(L.qkv.w, nh, dh, max_len)) {
            // windows of 128 token slots holding whole sentences: Q|K|V never reach HBM whatever the sentence lengths
            timed("qkv_attention2", 2.0 * Td * L.qkv.w.N * L.qkv.w.K + att_flops, s, [&] {
                launch_qkv_attention2(L.qkv.w, x, L.qkv_b.as<float>(), d_cu, B, d_windows, n_windows, d_n_windows, max_len, nh, slots, ctx, s);
            });
        } else {
            // (q4 files: the 4-bit planes of the stacked matrix where its f16 image overflows an XCD's L2 and gemm256 takes the launch)
            const bool planes = L.qkv_q4.w.qs && L.qkv_q4.mfma_ok && gemm256_ && !gemm_naive_ && gemm256_supported(L.qkv_q4.w, t_pad);
            gemm("gemm_qkv", planes ? L.qkv_q4 : L.qkv, x, L.qkv_b.as<float>(), nullptr, qkv, EPI_BIAS);
            timed("attention", att_flops, s, [&] {
                if (attn_naive_ || !launch_attention_mfma(qkv, d_cu, B, nh, dh, max_len, ctx, s))
                    launch_attention_naive(qkv, d_cu, B, nh, dh, max_len, ctx, s);
            });
        }
        if (tail_ && !gemm_naive_ && L.o.mfma_ok && L.ffi.mfma_ok && L.ffo.mfma_ok && layer_tail_supported(L.o.w, L.ffi.w, L.ffo.w)) {
            // out-projection + LN + FFN + LN in one launch, a pair of specialist waves per 32 tokens: y and the intermediate
            // never leave the chip
            timed("layer_tail", 2.0 * Td * H * H + 4.0 * Td * H * I, s, [&] {
                launch_layer_tail(L.o.w, L.ffi.w, L.ffo.w, ctx, x, L.o_b.as<float>(), L.ln_att_w.as<float>(),
                                  L.ln_att_b.as<float>(), L.ffi_b.as<float>(), L.ffo_b.as<float>(),
                                  L.ln_out_w.as<float>(), L.ln_out_b.as<float>(), x, t_pad, s);
            });
        } else {
            gemm("gemm_attn_out", L.o, ctx, L.o_b.as<float>(), x, y, EPI_BIAS_RESID);
            timed("layernorm", 0.0, s, [&] { launch_layernorm(y, L.ln_att_w.as<float>(), L.ln_att_b.as<float>(), T, H, s); });
            gemm("gemm_ffn_up", L.ffi, y, L.ffi_b.as<float>(), nullptr, ff, EPI_BIAS_GELU);
            gemm("gemm_ffn_down", L.ffo, ff, L.ffo_b.as<float>(), y, x, EPI_BIAS_RESID);
            timed("layernorm", 0.0, s, [&] { launch_layernorm(x, L.ln_out_w.as<float>(), L.ln_out_b.as<float>(), T, H, s); });
        }
        tap(il + 1);
    }
    // (the one-launch kernel's workgroups pool their sentences themselves)
    if (!one_launch) timed("pool_normalize", 2.0 * Td * H, s, [&] { launch_pool_normalize(x, d_cu, B, H, max_len, status_.as<int>(), d_out, s); });
    (void)I;
    HIP_OK(hipGetLastError(), err, -1);
    HIP_OK(hipEventRecord(busy_, s), err, -1);
    return 0;
}

// f32 files at the reference's precision (f32_route.hip; reference bert.cpp:784-913 with GGML_TYPE_F32 tensors): the same
// sequence of operations as the tiled family, every one in f32.  Called from eval_packed_device behind its workspace sizing
// and its wait for the previous pass.
int Engine::forward_f32(const int32_t *d_tokens, const int32_t *d_cu, int B, int T, int max_len, float *d_out, hipStream_t s, float *d_hidden,
                        std::string &err) {
    const int H = hp_.n_embd, I = hp_.n_intermediate, nh = hp_.n_head, dh = H / nh;
    float *x = x_.as<float>(), *qkv = qkv_.as<float>(), *ctx = ctx_.as<float>(), *y = y_.as<float>(), *ff = ff_.as<float>();
    const double Td = (double)T;
    auto gemm = [&](const char *name, GemmWeightStore &W, const float *A, const float *bias, const float *resid, float *C, int epi) {
        if (profiling_ && replay_name_.empty()) families_["family:gemm_f32"] += 1;
        timed(name, 2.0 * Td * W.w.N * W.w.K, s, [&] { launch_f32_gemm(A, W.w.w32, bias, resid, C, T, W.w.N, W.w.K, epi, s); });
    };
    auto tap = [&](int idx) {
        if (d_hidden) (void)hipMemcpyAsync(d_hidden + (size_t)idx * T * H, x, (size_t)T * H * 4, hipMemcpyDeviceToDevice, s);
    };
    timed("embed_ln", 0.0, s, [&] {
        launch_f32_embed_ln(word_emb_.as<float>(), type_emb_.as<float>(), pos_emb_.as<float>(), ln_e_w_.as<float>(), ln_e_b_.as<float>(), d_tokens,
                            d_cu, B, T, H, hp_.n_vocab, x, s);
    });
    tap(0);
    for (int il = 0; il < hp_.n_layer; ++il) {
        LayerWeights &L = *layers_[il];
        gemm("gemm_qkv", L.qkv, x, L.qkv_b.as<float>(), nullptr, qkv, EPI_BIAS);
        timed("attention", 4.0 * Td * max_len * H, s, [&] { launch_f32_attention(qkv, d_cu, B, nh, dh, max_len, ctx, s); });
        gemm("gemm_attn_out", L.o, ctx, L.o_b.as<float>(), x, y, EPI_BIAS_RESID);
        timed("layernorm", 0.0, s, [&] { launch_f32_layernorm(y, L.ln_att_w.as<float>(), L.ln_att_b.as<float>(), T, H, s); });
        gemm("gemm_ffn_up", L.ffi, y, L.ffi_b.as<float>(), nullptr, ff, EPI_BIAS_GELU);
        gemm("gemm_ffn_down", L.ffo, ff, L.ffo_b.as<float>(), y, x, EPI_BIAS_RESID);
        timed("layernorm", 0.0, s, [&] { launch_f32_layernorm(x, L.ln_out_w.as<float>(), L.ln_out_b.as<float>(), T, H, s); });
        tap(il + 1);
    }
    timed("pool_normalize", 2.0 * Td * H, s, [&] { launch_f32_pool_normalize(x, d_cu, B, H, max_len, status_.as<int>(), d_out, s); });
    (void)I;
    HIP_OK(hipGetLastError(), err, -1);
    HIP_OK(hipEventRecord(busy_, s), err, -1);
    return 0;
}

static bool ensure_pinned(void **p, size_t *cap, size_t need, std::string &err) {
    if (need <= *cap) return true;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *cap = 0;
    need += need / 4;
    HIP_OK(hipHostMalloc(p, need, hipHostMallocDefault), err, false);
    *cap = need;
    return true;
}

// The rows of a call's LAST chunk leave the pinned block with nothing to hide the copy behind (earlier chunks are copied out under
// the next one's forward pass): a large block goes out on four threads (one core moves ~10 GB/s: 0.9 ms for the 9 MB of 6000
// MiniLM rows).  A thread that cannot be started is not an error: the caller copies its part.
static void copy_rows_out(float *dst, const float *src, size_t bytes) {
    constexpr size_t PART_MIN = (size_t)1 << 20;
    constexpr int MAX_PARTS = 4;
    const int parts = (int)std::min<size_t>(MAX_PARTS, bytes / PART_MIN);
    if (parts <= 1) { memcpy(dst, src, bytes); return; }
    const size_t each = (bytes / parts + 4095) & ~(size_t)4095;
    std::thread helpers[MAX_PARTS - 1];
    size_t inline_from = each;                         // [0, each) is the caller's; [inline_from, bytes) too when a start fails
    for (int k = 1; k < parts; ++k) {
        const size_t off = (size_t)k * each, n = std::min(each, bytes - std::min(bytes, off));
        if (n == 0) break;
        try {
            helpers[k - 1] = std::thread([=] { memcpy((char *)dst + off, (const char *)src + off, n); });
            inline_from = off + n;
        } catch (const std::system_error &) {
            break;
        }
    }
    memcpy(dst, src, std::min(each, bytes));
    if (inline_from < bytes) memcpy((char *)dst + inline_from, (const char *)src + inline_from, bytes - inline_from);
    for (auto &h : helpers)
        if (h.joinable()) h.join();
}

int Engine::eval_packed_host(const int32_t *tokens, const int32_t *cu, int B, float *embeddings, std::string &err,
                             float *d_embeddings) {
    if (B <= 0) return 0;
#ifdef BERT_HIP_HOST_TRACE
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_call = now();
    double t_mark = t_call;
    auto lap = [&](const char *what, size_t i) { const double t = now(); fprintf(stderr, "[host] chunk %zu %-10s %.3f ms\n", i, what, t - t_mark); t_mark = t; };
#define HOST_LAP(what, i) lap(what, i)
#else
#define HOST_LAP(what, i) do { } while (0)
#endif
    HIP_OK(hipSetDevice(device_), err, -1);
    const int H = hp_.n_embd;
    // chunks [b0, b1): at most chunk_tokens_ tokens, at least one sentence
    struct Chunk { int b0, b1, max_len; };
    std::vector<Chunk> chunks;
    size_t max_T = 0, max_nb = 0;
    for (int b0 = 0; b0 < B;) {
        int b1 = b0 + 1, max_len = cu[b0 + 1] - cu[b0];
        while (b1 < B && cu[b1 + 1] - cu[b0] <= chunk_tokens_) { max_len = std::max(max_len, cu[b1 + 1] - cu[b1]); ++b1; }
        chunks.push_back({b0, b1, max_len});
        max_T = std::max(max_T, (size_t)(cu[b1] - cu[b0]));
        max_nb = std::max(max_nb, (size_t)(b1 - b0));
        b0 = b1;
    }
    // every buffer is sized for the largest chunk BEFORE anything is queued: growing one later would free memory that
    // a queued chunk still uses
    const int n_slots = chunks.size() > 1 ? 2 : 1;
    auto pad16 = [](size_t n) { return (n + 15) & ~(size_t)15; };
    const size_t in_bytes = pad16(max_T * 4) + pad16((max_nb + 1) * 4) + pad16(max_nb * sizeof(int2));
    for (int i = 0; i < n_slots; ++i) {
        HostSlot &sl = slot_[i];
        const size_t in_cap = sl.h_in_cap;
        // (16 spare bytes: the staging kernel copies whole 16-byte units)
        if (!ensure_pinned((void **)&sl.h_in, &sl.h_in_cap, in_bytes + 16, err)) return -1;
        if (sl.h_in_cap != in_cap || !sl.d_in_host) HIP_OK(hipHostGetDevicePointer((void **)&sl.d_in_host, sl.h_in, 0), err, -1);
        const size_t out_cap = sl.h_out_cap;
        if (!ensure_pinned((void **)&sl.h_out, &sl.h_out_cap, max_nb * H * 4, err)) return -1;
        if (sl.h_out_cap != out_cap || !sl.d_out_host)
            HIP_OK(hipHostGetDevicePointer((void **)&sl.d_out_host, sl.h_out, 0), err, -1);
        if (!sl.d_in.ensure(in_bytes + 16, err) || (d_embeddings && !sl.d_out.ensure(max_nb * H * 4, err))) return -1;
        if (!sl.done) HIP_OK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming), err, -1);
    }
    if (!ensure_workspace((int)((max_T + 255) / 256 * 256), (int)max_nb, err)) return -1;
    HOST_LAP("prepared", (size_t)0);

    // The stream executes H2D, forward, D2H of chunk after chunk; the host runs one chunk ahead: it stages chunk i
    // into slot i & 1 and queues it, then unpacks chunk i-1 while chunk i computes.
    auto unpack = [&](size_t i) -> bool {
        HostSlot &sl = slot_[i & 1];
        if (hipEventSynchronize(sl.done) != hipSuccess) { err = "hipEventSynchronize failed"; return false; }
        HOST_LAP("wait", i);
        if (!d_embeddings) {
            const size_t bytes = (size_t)(chunks[i].b1 - chunks[i].b0) * H * 4;
            if (i + 1 == chunks.size()) copy_rows_out(embeddings + (size_t)chunks[i].b0 * H, sl.h_out, bytes);
            else memcpy(embeddings + (size_t)chunks[i].b0 * H, sl.h_out, bytes);
        }
        HOST_LAP("copy-out", i);
        return true;
    };
    auto fail = [&]() { (void)hipStreamSynchronize(stream_); return -1; };        // nothing may stay queued on the slots
    std::vector<int2> windows;
    const int slots = window_slots();                         // (once per call: the lists below and the kernels that place by them)
    for (size_t i = 0; i < chunks.size(); ++i) {
        HostSlot &sl = slot_[i & 1];
        const int b0 = chunks[i].b0, nb = chunks[i].b1 - b0, T = cu[chunks[i].b1] - cu[b0];
        const size_t off_cu = pad16((size_t)T * 4), off_w = off_cu + pad16((size_t)(nb + 1) * 4);
        int32_t *h_cu = (int32_t *)(sl.h_in + off_cu);
        memcpy(sl.h_in, tokens + cu[b0], (size_t)T * 4);
        for (int j = 0; j <= nb; ++j) h_cu[j] = cu[b0 + j] - cu[b0];
        int n_windows = 0;
        if (chunks[i].max_len <= 128) {
            build_windows(h_cu, nb, windows, slots);
            n_windows = (int)windows.size();
            memcpy(sl.h_in + off_w, windows.data(), windows.size() * sizeof(int2));
        }
        const size_t staged = off_w + (size_t)n_windows * sizeof(int2);
        HOST_LAP("staged", i);
        if (stage_kernel_ && staged <= ((size_t)256 << 10)) {
            // (a small block — measured up to the 130 KB of a 256 x 128 batch: a few workgroups read it across the host link, the copy
            // engine's start-up is ~20 us of a 0.8 ms call; full 1 MiB chunks stay with the copy engine, off the compute stream)
            launch_stage_copy(sl.d_in_host, sl.d_in.p, staged, stream_);
        } else if (hipMemcpyAsync(sl.d_in.p, sl.h_in, staged, hipMemcpyHostToDevice, stream_) != hipSuccess) {
            err = "hipMemcpyAsync (ids) failed";
            return fail();
        }
        const char *d_in = (const char *)sl.d_in.p;
        // host destination: the pooling kernel's rows go straight into the pinned block (no D2H copy behind the pass)
        float *out = d_embeddings ? sl.d_out.as<float>() : sl.d_out_host;
        if (eval_packed_device((const int32_t *)d_in, (const int32_t *)(d_in + off_cu), nb, T, chunks[i].max_len, out, stream_, nullptr, err,
                               n_windows ? (const int2 *)(d_in + off_w) : nullptr, n_windows, slots) != 0)
            return fail();
        if ((d_embeddings && hipMemcpyAsync(d_embeddings + (size_t)b0 * H, sl.d_out.p, (size_t)nb * H * 4, hipMemcpyDeviceToDevice, stream_) != hipSuccess) ||
            hipEventRecord(sl.done, stream_) != hipSuccess) {
            err = "hipMemcpyAsync (embeddings) failed";
            return fail();
        }
        HOST_LAP("queued", i);
        if (i >= 1 && !unpack(i - 1)) return fail();           // while chunk i computes; frees the slot chunk i+1 stages into
    }
    if (!unpack(chunks.size() - 1)) return fail();
#undef HOST_LAP
    return 0;
}

int Engine::eval_hidden(const int32_t *tokens, int N, float *hidden, float *embedding, std::string &err) {
    HIP_OK(hipSetDevice(device_), err, -1);
    const int H = hp_.n_embd, L = hp_.n_layer;
    int32_t cu[2] = {0, N};
    if (!d_tokens_.ensure((size_t)N * 4, err) || !d_cu_.ensure(8, err)) return -1;
    if (!d_hidden_.ensure((size_t)(L + 1) * N * H * 4, err)) return -1;
    const int t_pad = (N + 255) / 256 * 256;
    if (!ensure_workspace(t_pad, 1, err)) return -1;
    HIP_OK(hipMemcpy(d_tokens_.p, tokens, (size_t)N * 4, hipMemcpyHostToDevice), err, -1);
    HIP_OK(hipMemcpy(d_cu_.p, cu, 8, hipMemcpyHostToDevice), err, -1);
    if (eval_packed_device(d_tokens_.as<int32_t>(), d_cu_.as<int32_t>(), 1, N, N, d_out_.as<float>(), stream_,
                           d_hidden_.as<float>(), err) != 0)
        return -1;
    HIP_OK(hipStreamSynchronize(stream_), err, -1);
    if (hidden) HIP_OK(hipMemcpy(hidden, d_hidden_.p, (size_t)(L + 1) * N * H * 4, hipMemcpyDeviceToHost), err, -1);
    if (embedding) HIP_OK(hipMemcpy(embedding, d_out_.p, (size_t)H * 4, hipMemcpyDeviceToHost), err, -1);
    return 0;
}

}  // namespace bert_hip
