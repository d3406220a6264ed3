// test_api.hip — standalone kernel entry points of bert_hip.h for op-level parity tests.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bert_hip_test.h"
#include <stdexcept>

#include "engine.h"
#include "multi_device.h"

using namespace bert_hip;

#define CK(expr)                                                                       \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e__));                \
            return -1;                                                                 \
        }                                                                              \
    } while (0)

extern "C" {

int32_t bert_hip_test_gemm(int32_t M, int32_t N, int32_t K, const uint16_t *A, const void *W, int32_t wtype,
                           const float *bias, const uint16_t *resid, int32_t epilogue, int32_t impl, uint16_t *C) {
    std::string err;
    HostTensor t;
    t.type = wtype; t.n_dims = 2; t.ne0 = K; t.ne1 = N; t.data = (const uint8_t *)W;
    t.nbytes = wtype_row_bytes(wtype, K) * (size_t)N;
    GemmWeightStore ws;
    if (!ws.build({&t}, impl == 1, err)) { fprintf(stderr, "bert_hip_test_gemm: %s\n", err.c_str()); return -1; }
    if (impl != 1 && !ws.mfma_ok) { fprintf(stderr, "bert_hip_test_gemm: shape not supported by the MFMA path\n"); return -2; }
    const int M_pad = impl == 3 ? (M + 255) / 256 * 256 : (M + GEMM_BM - 1) / GEMM_BM * GEMM_BM;
    DevBuf dA, dB, dR, dC;
    if (!dA.alloc((size_t)M_pad * K * 2, err) || !dC.alloc((size_t)M_pad * N * 2, err) || !dB.upload(bias, (size_t)N * 4, err)) {
        fprintf(stderr, "bert_hip_test_gemm: %s\n", err.c_str());
        return -1;
    }
    CK(hipMemcpy(dA.p, A, (size_t)M * K * 2, hipMemcpyHostToDevice));
    if (resid) {
        if (!dR.alloc((size_t)M_pad * N * 2, err)) return -1;
        CK(hipMemcpy(dR.p, resid, (size_t)M * N * 2, hipMemcpyHostToDevice));
    }
    if (impl == 3) {
        if (!gemm256_supported(ws.w, M_pad)) return -2;
        launch_gemm256(ws.w, dA.as<half_t>(), dB.as<float>(), dR.as<half_t>(), dC.as<half_t>(), M_pad, epilogue, nullptr);
    } else if (impl == 0) launch_gemm_mfma(ws.w, dA.as<half_t>(), dB.as<float>(), dR.as<half_t>(), dC.as<half_t>(), M_pad, epilogue, nullptr);
    else launch_gemm_naive(ws.w, dA.as<half_t>(), dB.as<float>(), dR.as<half_t>(), dC.as<half_t>(), M, epilogue, nullptr);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(C, dC.p, (size_t)M * N * 2, hipMemcpyDeviceToHost));
    return 0;
}

int32_t bert_hip_test_gemm_lnfold(int32_t M, int32_t K1, int32_t H, int32_t N2, const uint16_t *A1, const uint16_t *W1, const float *b1,
                                  const uint16_t *r, const float *rg, const float *rb, const uint16_t *W2, const float *b2, const float *g,
                                  const float *be, int32_t epi2, uint16_t *u_out, uint16_t *out2, float *rows_out) {
    std::string err;
    auto tensor = [](const uint16_t *w, int n, int k) {
        HostTensor t;
        t.type = W_F16; t.n_dims = 2; t.ne0 = k; t.ne1 = n; t.data = (const uint8_t *)w; t.nbytes = (size_t)n * k * 2;
        return t;
    };
    const HostTensor t1 = tensor(W1, H, K1), t2 = tensor(W2, N2, H);
    GemmWeightStore w1, w2;
    DevBuf waug, gb;
    if (!w1.build({&t1}, false, err) || !w2.build_ln_fold({&t2}, g, be, b2, waug, err)) { fprintf(stderr, "bert_hip_test_gemm_lnfold: %s\n", err.c_str()); return -1; }
    const int M_pad = (M + 255) / 256 * 256, P = 2 * H / 256;
    if (!gemm256_supported(w1.w, M_pad) || !gemm256_supported(w2.w, M_pad) || H % 256) return -2;
    DevBuf dA, dB1, dR, dU, dOut, dStats, dRows, dRowsRes;
    if (!dA.alloc((size_t)M_pad * K1 * 2, err) || !dB1.upload(b1, (size_t)H * 4, err) || !dR.alloc((size_t)M_pad * H * 2, err) || !dU.alloc((size_t)M_pad * H * 2, err) ||
        !dOut.alloc((size_t)M_pad * N2 * 2, err) || !dStats.alloc((size_t)M_pad * P * 8, err) || !dRows.alloc((size_t)M_pad * 16, err) || !dRowsRes.alloc((size_t)M_pad * 16, err)) return -1;
    CK(hipMemcpy(dA.p, A1, (size_t)M * K1 * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dR.p, r, (size_t)M * H * 2, hipMemcpyHostToDevice));
    GemmLnFold ln;
    ln.flags = GemmLnFold::STATS; ln.stats = dStats.as<float2>();
    if (rg) {
        // the residual's own row statistics (what the mat-mul that produced r would have left behind), and the packed (gamma, beta + bias)
        std::vector<float> rows((size_t)M_pad * 4, 0.f);
        for (int t = 0; t < M; ++t) {
            double s1 = 0, s2 = 0;
            for (int f = 0; f < H; ++f) { _Float16 h; memcpy(&h, &r[(size_t)t * H + f], 2); s1 += (double)(float)h; s2 += (double)(float)h * (double)(float)h; }
            const double mean = s1 / H, var = std::max(s2 / H - mean * mean, 0.0) + 1e-5, sd = std::sqrt(var);
            rows[4 * (size_t)t] = (float)(1.0 / sd); rows[4 * (size_t)t + 1] = (float)(-mean / sd); rows[4 * (size_t)t + 2] = (float)-mean; rows[4 * (size_t)t + 3] = (float)sd;
        }
        CK(hipMemcpy(dRowsRes.p, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
        std::vector<uint32_t> v((size_t)H);
        for (int f = 0; f < H; ++f) {
            const _Float16 gg = (_Float16)rg[f], bb = (_Float16)(rb[f] + b1[f]);
            uint16_t gu, bu; memcpy(&gu, &gg, 2); memcpy(&bu, &bb, 2);
            v[(size_t)f] = (uint32_t)gu | ((uint32_t)bu << 16);
        }
        if (!gb.upload(v.data(), v.size() * 4, err)) return -1;
        ln.flags |= GemmLnFold::RES; ln.rows_res = dRowsRes.as<float4>(); ln.gb = gb.as<unsigned>();
    }
    launch_gemm256(w1.w, dA.as<half_t>(), dB1.as<float>(), dR.as<half_t>(), dU.as<half_t>(), M_pad, EPI_BIAS_RESID, nullptr, &ln);
    launch_ln_rows_finalize(dStats.as<float2>(), P, M_pad, H, dRows.as<float4>(), nullptr);
    GemmLnFold in;
    in.flags = GemmLnFold::IN; in.rows_in = dRows.as<float4>(); in.waug = waug.as<half_t>();
    launch_gemm256(w2.w, dU.as<half_t>(), nullptr, nullptr, dOut.as<half_t>(), M_pad, epi2, nullptr, &in);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(u_out, dU.p, (size_t)M * H * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out2, dOut.p, (size_t)M * N2 * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(rows_out, dRows.p, (size_t)M * 16, hipMemcpyDeviceToHost));
    return 0;
}

int32_t bert_hip_test_attention(int32_t n_sentences, const int32_t *cu_seqlens, int32_t n_head, int32_t d_head,
                                const uint16_t *qkv, int32_t impl, uint16_t *out) {
    std::string err;
    const int T = cu_seqlens[n_sentences], H = n_head * d_head;
    int max_len = 0;
    for (int b = 0; b < n_sentences; ++b) max_len = std::max(max_len, cu_seqlens[b + 1] - cu_seqlens[b]);
    DevBuf dq, dcu, dout;
    if (!dq.upload(qkv, (size_t)T * 3 * H * 2, err) || !dcu.upload(cu_seqlens, (size_t)(n_sentences + 1) * 4, err) ||
        !dout.alloc((size_t)T * H * 2, err)) {
        fprintf(stderr, "bert_hip_test_attention: %s\n", err.c_str());
        return -1;
    }
    if (impl == 0) {
        if (!launch_attention_mfma(dq.as<half_t>(), dcu.as<int32_t>(), n_sentences, n_head, d_head, max_len, dout.as<half_t>(), nullptr)) {
            fprintf(stderr, "bert_hip_test_attention: shape not supported by the MFMA path\n");
            return -2;
        }
    } else {
        launch_attention_naive(dq.as<half_t>(), dcu.as<int32_t>(), n_sentences, n_head, d_head, max_len, dout.as<half_t>(), nullptr);
    }
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, dout.p, (size_t)T * H * 2, hipMemcpyDeviceToHost));
    return 0;
}

int32_t bert_hip_test_qkv_attention(int32_t n_sentences, const int32_t *cu_seqlens, int32_t n_head, int32_t d_head,
                                    const uint16_t *x, const void *Wqkv, int32_t wtype, const float *bias, int32_t fused,
                                    uint16_t *out) {
    std::string err;
    const int T = cu_seqlens[n_sentences], H = n_head * d_head;
    const int T_pad = (T + GEMM_BM - 1) / GEMM_BM * GEMM_BM;
    int max_len = 0;
    for (int b = 0; b < n_sentences; ++b) max_len = std::max(max_len, cu_seqlens[b + 1] - cu_seqlens[b]);
    HostTensor t;
    t.type = wtype; t.n_dims = 2; t.ne0 = H; t.ne1 = 3 * H; t.data = (const uint8_t *)Wqkv;
    t.nbytes = wtype_row_bytes(wtype, H) * (size_t)3 * H;
    GemmWeightStore ws;
    if (!ws.build({&t}, false, err)) { fprintf(stderr, "bert_hip_test_qkv_attention: %s\n", err.c_str()); return -1; }
    if (!ws.mfma_ok) return -2;
    DevBuf dx, dqkv, dcu, db, dout;
    if (!dx.alloc((size_t)T_pad * H * 2, err) || !dqkv.alloc((size_t)T_pad * 3 * H * 2, err) ||
        !dcu.upload(cu_seqlens, (size_t)(n_sentences + 1) * 4, err) || !db.upload(bias, (size_t)3 * H * 4, err) ||
        !dout.alloc((size_t)T_pad * H * 2, err)) {
        fprintf(stderr, "bert_hip_test_qkv_attention: %s\n", err.c_str());
        return -1;
    }
    CK(hipMemcpy(dx.p, x, (size_t)T * H * 2, hipMemcpyHostToDevice));
    if (fused >= 2 && fused <= 4) {
        // second-generation kernel: 2 = next-fit windows of whole sentences, 3 = the uniform placement rule
        if (!qkv_attention2_supported(ws.w, n_head, d_head, max_len)) return -2;
        std::vector<int2> win;
        DevBuf dwin;
        DevBuf dcount;
        int n_win = 0;
        if (fused == 2) {
            Engine::build_windows(cu_seqlens, n_sentences, win, window_slots());
            if (!dwin.upload(win.data(), win.size() * sizeof(int2), err)) return -1;
            n_win = (int)win.size();
        } else if (fused == 4) {
            // the same windows built on the device; the grid is the launcher's upper bound
            if (!dwin.alloc((size_t)n_sentences * sizeof(int2), err) || !dcount.alloc(sizeof(int), err)) return -1;
            launch_build_windows(dcu.as<int32_t>(), n_sentences, dwin.as<int2>(), dcount.as<int>(), window_slots(), nullptr);
            n_win = qkv_attention2_max_windows(n_sentences, T, window_slots());
        }
        launch_qkv_attention2(ws.w, dx.as<half_t>(), db.as<float>(), dcu.as<int32_t>(), n_sentences,
                              fused == 3 ? nullptr : dwin.as<int2>(), n_win, fused == 4 ? dcount.as<int>() : nullptr, max_len, n_head,
                              window_slots(), dout.as<half_t>(), nullptr);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
    } else {
        launch_gemm_mfma(ws.w, dx.as<half_t>(), db.as<float>(), nullptr, dqkv.as<half_t>(), T_pad, EPI_BIAS, nullptr);
        if (!launch_attention_mfma(dqkv.as<half_t>(), dcu.as<int32_t>(), n_sentences, n_head, d_head, max_len, dout.as<half_t>(), nullptr))
            return -2;
    }
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, dout.p, (size_t)T * H * 2, hipMemcpyDeviceToHost));
    return 0;
}

int32_t bert_hip_test_layer_tail(int32_t M, int32_t H, int32_t I, const uint16_t *ctx, const uint16_t *x, const void *Wo,
                                 const void *W1, const void *W2, int32_t wtype, const float *bo, const float *g1,
                                 const float *be1, const float *b1, const float *b2, const float *g2, const float *be2,
                                 int32_t impl, uint16_t *out) {
    std::string err;
    HostTensor to, t1, t2;
    to.type = wtype; to.n_dims = 2; to.ne0 = H; to.ne1 = H; to.data = (const uint8_t *)Wo; to.nbytes = wtype_row_bytes(wtype, H) * (size_t)H;
    t1.type = wtype; t1.n_dims = 2; t1.ne0 = H; t1.ne1 = I; t1.data = (const uint8_t *)W1; t1.nbytes = wtype_row_bytes(wtype, H) * (size_t)I;
    t2.type = wtype; t2.n_dims = 2; t2.ne0 = I; t2.ne1 = H; t2.data = (const uint8_t *)W2; t2.nbytes = wtype_row_bytes(wtype, I) * (size_t)H;
    GemmWeightStore wo, w1, w2;
    if (!wo.build({&to}, false, err) || !w1.build({&t1}, false, err, true) || !w2.build({&t2}, false, err, true)) {
        fprintf(stderr, "bert_hip_test_layer_tail: %s\n", err.c_str());
        return -1;
    }
    if (!wo.mfma_ok || !w1.mfma_ok || !w2.mfma_ok) return -2;
    const int M_pad = (M + GEMM_BM - 1) / GEMM_BM * GEMM_BM;
    DevBuf dc, dx, dy, dout, dbo, dg1, dbe1, db1, db2, dg2, dbe2;
    if (!dc.alloc((size_t)M_pad * H * 2, err) || !dx.alloc((size_t)M_pad * H * 2, err) || !dy.alloc((size_t)M_pad * H * 2, err) ||
        !dout.alloc((size_t)M_pad * H * 2, err) || !dbo.upload(bo, (size_t)H * 4, err) || !dg1.upload(g1, (size_t)H * 4, err) ||
        !dbe1.upload(be1, (size_t)H * 4, err) || !db1.upload(b1, (size_t)I * 4, err) || !db2.upload(b2, (size_t)H * 4, err) ||
        !dg2.upload(g2, (size_t)H * 4, err) || !dbe2.upload(be2, (size_t)H * 4, err)) {
        fprintf(stderr, "bert_hip_test_layer_tail: %s\n", err.c_str());
        return -1;
    }
    CK(hipMemcpy(dc.p, ctx, (size_t)M * H * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx.p, x, (size_t)M * H * 2, hipMemcpyHostToDevice));
    if (impl == 1) {
        if (!layer_tail_supported(wo.w, w1.w, w2.w)) return -2;
        launch_layer_tail(wo.w, w1.w, w2.w, dc.as<half_t>(), dx.as<half_t>(), dbo.as<float>(), dg1.as<float>(), dbe1.as<float>(),
                          db1.as<float>(), db2.as<float>(), dg2.as<float>(), dbe2.as<float>(), dout.as<half_t>(), M_pad, nullptr);
    } else {
        // three GEMM kernels + two LayerNorm kernels
        DevBuf dff;
        if (!dff.alloc((size_t)M_pad * I * 2, err)) return -1;
        launch_gemm_mfma(wo.w, dc.as<half_t>(), dbo.as<float>(), dx.as<half_t>(), dy.as<half_t>(), M_pad, EPI_BIAS_RESID, nullptr);
        launch_layernorm(dy.as<half_t>(), dg1.as<float>(), dbe1.as<float>(), M_pad, H, nullptr);
        launch_gemm_mfma(w1.w, dy.as<half_t>(), db1.as<float>(), nullptr, dff.as<half_t>(), M_pad, EPI_BIAS_GELU, nullptr);
        launch_gemm_mfma(w2.w, dff.as<half_t>(), db2.as<float>(), dy.as<half_t>(), dout.as<half_t>(), M_pad, EPI_BIAS_RESID, nullptr);
        launch_layernorm(dout.as<half_t>(), dg2.as<float>(), dbe2.as<float>(), M_pad, H, nullptr);
        CK(hipDeviceSynchronize());
    }
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, dout.p, (size_t)M * H * 2, hipMemcpyDeviceToHost));
    return 0;
}


int32_t bert_hip_test_embed_ln(int32_t table_type, int32_t H, int32_t n_vocab, int32_t n_pos, const void *word, const void *type,
                               const void *pos, const float *gamma, const float *beta, const bert_vocab_id *tokens,
                               const int32_t *cu_seqlens, int32_t n_sentences, uint16_t *out) {
    std::string err;
    const int T = cu_seqlens[n_sentences];
    int max_len = 0;
    for (int b = 0; b < n_sentences; ++b) max_len = std::max(max_len, cu_seqlens[b + 1] - cu_seqlens[b]);
    const size_t rb = wtype_row_bytes(table_type, H);
    DevBuf dw, dt, dp, dg, db, dtok, dcu, dout;
    if (!dw.upload(word, rb * n_vocab, err) || !dt.upload(type, rb * 2, err) || !dp.upload(pos, rb * n_pos, err) ||
        !dg.upload(gamma, (size_t)H * 4, err) || !db.upload(beta, (size_t)H * 4, err) || !dtok.upload(tokens, (size_t)T * 4, err) ||
        !dcu.upload(cu_seqlens, (size_t)(n_sentences + 1) * 4, err) || !dout.alloc((size_t)T * H * 2, err)) {
        fprintf(stderr, "bert_hip_test_embed_ln: %s\n", err.c_str());
        return -1;
    }
    launch_embed_ln(dw.p, dt.p, dp.p, table_type, dg.as<float>(), db.as<float>(), dtok.as<int32_t>(), dcu.as<int32_t>(), n_sentences,
                    T, H, n_vocab, max_len, dout.as<half_t>(), nullptr);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, dout.p, (size_t)T * H * 2, hipMemcpyDeviceToHost));
    return 0;
}

int32_t bert_hip_test_pool_normalize(int32_t H, const uint16_t *x, const int32_t *cu_seqlens, int32_t n_sentences, int32_t max_len,
                                     float *out, int32_t *status) {
    std::string err;
    const int T = cu_seqlens[n_sentences];
    DevBuf dx, dcu, dout, dst;
    if (!dx.upload(x, (size_t)T * H * 2, err) || !dcu.upload(cu_seqlens, (size_t)(n_sentences + 1) * 4, err) ||
        !dout.alloc((size_t)n_sentences * H * 4, err) || !dst.alloc(16, err)) {
        fprintf(stderr, "bert_hip_test_pool_normalize: %s\n", err.c_str());
        return -1;
    }
    launch_pool_normalize(dx.as<half_t>(), dcu.as<int32_t>(), n_sentences, H, max_len, dst.as<int>(), dout.as<float>(), nullptr);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, dout.p, (size_t)n_sentences * H * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(status, dst.p, 4, hipMemcpyDeviceToHost));
    return 0;
}

int32_t bert_hip_test_model_digest(const char *fname, int32_t *legacy_q4, uint64_t *digest) {
    ModelFile mf;
    std::string err;
    if (!mf.load(fname, false, err)) { fprintf(stderr, "bert_hip_test_model_digest: %s\n", err.c_str()); return -1; }
    uint64_t h = 1469598103934665603ull;                       // FNV-1a over name, type and bytes (current layout) of every tensor
    auto mix = [&](const void *p, size_t n) { for (size_t i = 0; i < n; ++i) { h ^= ((const uint8_t *)p)[i]; h *= 1099511628211ull; } };
    for (const auto &kv : mf.tensors) {
        mix(kv.first.data(), kv.first.size());
        mix(&kv.second.type, 4);
        mix(kv.second.data, kv.second.nbytes);
    }
    *legacy_q4 = mf.legacy_q4 ? 1 : 0;
    *digest = h;
    return (int32_t)mf.tensors.size();
}

void bert_hip_test_shard_bounds(const int32_t *cu_seqlens, int32_t n_sentences, int32_t n_shards, int32_t *bounds) {
    std::vector<int> b;
    shard_bounds(cu_seqlens, n_sentences, n_shards, b);
    for (size_t i = 0; i < b.size(); ++i) bounds[i] = b[i];
}

int32_t bert_hip_test_build_windows(const int32_t *cu_seqlens, int32_t n_sentences, int32_t *windows) {
    std::vector<int2> w;
    Engine::build_windows(cu_seqlens, n_sentences, w, window_slots());
    for (size_t i = 0; i < w.size(); ++i) { windows[2 * i] = w[i].x; windows[2 * i + 1] = w[i].y; }
    return (int32_t)w.size();
}

int32_t bert_hip_test_max_windows(int32_t n_sentences, int32_t n_tokens) { return qkv_attention2_max_windows(n_sentences, n_tokens, window_slots()); }

int32_t bert_hip_test_set_window_slots(int32_t slots) { set_window_slots(slots); return window_slots(); }

int32_t bert_hip_test_build_windows_device(const int32_t *cu_seqlens, int32_t n_sentences, int32_t *windows) {
    std::string err;
    DevBuf dcu, dwin, dcount;
    if (!dcu.upload(cu_seqlens, (size_t)(n_sentences + 1) * sizeof(int32_t), err) ||
        !dwin.alloc((size_t)std::max(n_sentences, 1) * sizeof(int2), err) || !dcount.alloc(sizeof(int), err)) return -1;
    launch_build_windows(dcu.as<int32_t>(), n_sentences, dwin.as<int2>(), dcount.as<int>(), window_slots(), nullptr);
    int n = -1;
    if (hipMemcpy(&n, dcount.p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess || n < 0 || n > n_sentences) return -1;
    if (n && hipMemcpy(windows, dwin.p, (size_t)n * sizeof(int2), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}

int64_t bert_hip_test_shard_threads_created(void) { return (int64_t)ShardWorkers::threads_created(); }

int32_t bert_hip_test_dispatch(const bert_vocab_id *tokens, const int32_t *cu_seqlens, int32_t n_sentences, int32_t n_shards,
                               int32_t H, float *out) {
    std::vector<int> bounds;
    shard_bounds(cu_seqlens, n_sentences, n_shards, bounds);
    // one pool for the life of the process, like a context's (sized for the largest shard count seen so far)
    static std::unique_ptr<ShardWorkers> pool;
    if (!pool || pool->n_threads() < n_shards - 1) pool.reset(new ShardWorkers(n_shards - 1));
    return pool->run(bounds, [&](int shard, int b0, int b1) {
        if (H < 0) throw std::runtime_error("injected failure in shard " + std::to_string(shard));
        // stands in for Engine::eval_packed_host(tokens, cu + b0, b1 - b0, out + b0 * H): the global token array and a
        // window of the prefix sums, results into the caller's rows of this shard
        const int32_t *cu = cu_seqlens + b0;
        float *dst = out + (size_t)b0 * H;
        for (int b = 0; b < b1 - b0; ++b) {
            long long sum = 0;
            for (int t = cu[b]; t < cu[b + 1]; ++t) sum += tokens[t];
            for (int e = 0; e < H; ++e) dst[(size_t)b * H + e] = e == 0 ? (float)sum : e == 1 ? (float)(cu[b + 1] - cu[b]) : e == 2 ? (float)shard : (float)(b0 + b);
        }
        return 0;
    });
}

}  // extern "C"
