// misc_kernels.hip — the HBM-bound pieces of the forward pass (gfx950): embedding gather-sum +
// LayerNorm, stand-alone LayerNorm, mean-pool + L2 normalise.  One 64-lane wavefront per token row
// with __shfl_xor reductions; no LDS needed.
#include "kernels.h"

#include <algorithm>
#include "pool_normalize.h"

namespace bert_hip {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wave_sum(float v) { return wave_sum_f32(v); }      // (kernels.h: the __shfl_xor butterfly's pairs without the LDS)

// Element e of row r of an embedding table stored in the model-file layout,
// dequantised to f32 exactly as ggml_get_rows does: f16 -> f32, (q-8)*d, q*d+m.
__device__ __forceinline__ float table_elem(const void *tab, int type, int H, int r, int e) {
    if (type == 0) return ((const float *)tab)[(size_t)r * H + e];
    if (type == 1) return (float)((const half_t *)tab)[(size_t)r * H + e];
    const int bs = type == 2 ? 18 : 20;
    const unsigned char *blk = (const unsigned char *)tab + ((size_t)r * (H / 32) + e / 32) * bs;
    const int i = e & 31;
    const float d = (float)*(const half_t *)blk;
    const unsigned char byte = blk[(type == 2 ? 2 : 4) + (i & 15)];
    const int q = i < 16 ? (byte & 0x0F) : (byte >> 4);
    if (type == 2) return (float)(q - 8) * d;
    const float m = (float)*(const half_t *)(blk + 2);
    return (float)q * d + m;
}

// Two adjacent elements (e, e+1; e even) of row r of a table, dequantised to f32.
__device__ __forceinline__ void table_pair(const void *tab, int type, int H, int r, int e, float &a, float &b) {
    if (type == 0) {
        const float2 v = *(const float2 *)((const float *)tab + (size_t)r * H + e);
        a = v.x; b = v.y;
    } else if (type == 1) {
        const f16x2 v = *(const f16x2 *)((const half_t *)tab + (size_t)r * H + e);
        a = (float)v[0]; b = (float)v[1];
    } else {
        a = table_elem(tab, type, H, r, e);
        b = table_elem(tab, type, H, r, e + 1);
    }
}

// reference bert.cpp:796-814: inpL = word[ids]; inpL = type[0] + inpL; inpL = pos[0..N-1] + inpL;
// LayerNorm (ggml_norm eps 1e-5) then gamma * x + beta.  One wave per token, one pass: the row is
// held in registers as NJ element pairs per lane (H <= 128 * NJ, H even).
template <int NJ>
__global__ __launch_bounds__(256) void embed_ln_kernel(const void *word, const void *type, const void *pos,
                                                       int table_type, const float *gamma, const float *beta,
                                                       const int32_t *__restrict__ tokens,
                                                       const int32_t *__restrict__ cu_seqlens, int n_sentences,
                                                       int T, int H, int n_vocab, half_t *__restrict__ out) {
    // wave-uniform token index: the sentence search below then runs on scalar loads (constant cache)
    const int t = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    // sentence of token t: largest b with cu[b] <= t
    int lo = 0, hi = n_sentences;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cu_seqlens[mid] <= t) lo = mid; else hi = mid;
    }
    const int p = t - cu_seqlens[lo];
    int id = tokens[t];
    id = id < 0 ? 0 : (id >= n_vocab ? n_vocab - 1 : id);   // ids are validated on the host API; clamp for safety

    float v[NJ][2];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e = 2 * lane + 128 * j;
        v[j][0] = v[j][1] = 0.f;
        if (e < H) {
            float w0, w1, t0, t1, p0, p1;
            table_pair(word, table_type, H, id, e, w0, w1);
            table_pair(type, table_type, H, 0, e, t0, t1);
            table_pair(pos, table_type, H, p, e, p0, p1);
            v[j][0] = p0 + (t0 + w0);
            v[j][1] = p1 + (t1 + w1);
            sum += v[j][0] + v[j][1];
        }
    }
    const float mean = wave_sum(sum) / H;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e = 2 * lane + 128 * j;
        if (e < H) {
            v[j][0] -= mean; v[j][1] -= mean;
            sq += v[j][0] * v[j][0] + v[j][1] * v[j][1];
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / H + 1e-5f);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e = 2 * lane + 128 * j;
        if (e < H) {
            f16x2 o;
            o[0] = (_Float16)(gamma[e] * (v[j][0] * rstd) + beta[e]);
            o[1] = (_Float16)(gamma[e + 1] * (v[j][1] * rstd) + beta[e + 1]);
            *(f16x2 *)(out + (size_t)t * H + e) = o;
        }
    }
}

// Same operation for f32 / f16 tables with H % 8 == 0 and q4 tables, laid out for bandwidth: the grid is (4-token group, sentence), so
// a wave knows its sentence and position without searching cu_seqlens; a lane owns 16-byte runs of the row (8
// features: one 16-byte load per f16 table row, one 16-byte store), NC runs per lane (H <= 512 * NC).
typedef _Float16 f16x8m __attribute__((ext_vector_type(8)));
template <int TT>
__device__ __forceinline__ void table_run8(const void *tab, int H, int r, int e0, float (&v)[8]) {
    if (TT == 0) {
        const float4 a = *(const float4 *)((const float *)tab + (size_t)r * H + e0);
        const float4 b = *(const float4 *)((const float *)tab + (size_t)r * H + e0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else if (TT == 1) {
        const f16x8m h = *(const f16x8m *)((const half_t *)tab + (size_t)r * H + e0);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
    } else {
        // q4_0 / q4_1 rows as stored in the file (18 / 20-byte blocks of 32: d [, m], 16 bytes of nibbles; element j < 16 is the
        // low nibble of byte j, element j + 16 the high one): a run of 8 is one nibble of 8 consecutive bytes.  Same
        // arithmetic as table_elem (ggml_get_rows: (q - 8) d, q d + m in f32).
        constexpr int bs = TT == 2 ? 18 : 20, qo = TT == 2 ? 2 : 4;
        const unsigned char *blk = (const unsigned char *)tab + ((size_t)r * (H / 32) + (e0 >> 5)) * bs;
        const int off = e0 & 31;
        const float d = (float)*(const half_t *)blk;
        const float m = TT == 3 ? (float)*(const half_t *)(blk + 2) : 0.f;
        unsigned short q16[4];                                // (blocks are 2-byte aligned)
#pragma unroll
        for (int i = 0; i < 4; ++i) q16[i] = *(const unsigned short *)(blk + qo + (off & 15) + 2 * i);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int byte = (q16[i >> 1] >> (8 * (i & 1))) & 0xff, q = off < 16 ? (byte & 0x0F) : (byte >> 4);
            v[i] = TT == 2 ? (float)(q - 8) * d : (float)q * d + m;
        }
    }
}
// SEARCH: the grid is (4-token group of the batch) and a wave finds its sentence by bisection of cu_seqlens on scalar loads
// — for batches of short sentences, where the (group, sentence) grid would be mostly empty workgroups.
template <int TT, int NC, bool SEARCH>
__global__ __launch_bounds__(256) void embed_ln_rows_kernel(const void *word, const void *type, const void *pos,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const int32_t *__restrict__ tokens,
                                                            const int32_t *__restrict__ cu_seqlens, int n_sentences, int T,
                                                            int H, int n_vocab, half_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    int p = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // position in the sentence / token
    int t;
    if constexpr (SEARCH) {
        t = p;
        if (t >= T) return;
        int lo = 0, hi = n_sentences;                         // largest b with cu[b] <= t
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cu_seqlens[mid] <= t) lo = mid; else hi = mid;
        }
        p = t - cu_seqlens[lo];
    } else {
        const int b = blockIdx.y;
        const int tok0 = cu_seqlens[b], n = cu_seqlens[b + 1] - tok0;
        if (p >= n) return;
        t = tok0 + p;
    }
    int id = tokens[t];
    id = id < 0 ? 0 : (id >= n_vocab ? n_vocab - 1 : id);   // ids are validated on the host API; clamp for safety
    float v[NC][8];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int e0 = 8 * (lane + 64 * j);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[j][i] = 0.f;
        if (e0 < H) {
            float w[8], ty[8], ps[8];
            table_run8<TT>(word, H, id, e0, w);
            table_run8<TT>(type, H, 0, e0, ty);
            table_run8<TT>(pos, H, p, e0, ps);
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[j][i] = ps[i] + (ty[i] + w[i]); sum += v[j][i]; }
        }
    }
    const float mean = wave_sum(sum) / H;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j)
        if (8 * (lane + 64 * j) < H)
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[j][i] -= mean; sq += v[j][i] * v[j][i]; }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / H + 1e-5f);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int e0 = 8 * (lane + 64 * j);
        if (e0 < H) {
            const float4 g0 = *(const float4 *)(gamma + e0), g1 = *(const float4 *)(gamma + e0 + 4);
            const float4 b0 = *(const float4 *)(beta + e0), b1 = *(const float4 *)(beta + e0 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            f16x8m o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (_Float16)(gg[i] * (v[j][i] * rstd) + bb[i]);
            *(f16x8m *)(out + (size_t)t * H + e0) = o;
        }
    }
}

void launch_embed_ln(const void *word, const void *type, const void *pos, int table_type, const float *gamma,
                     const float *beta, const int32_t *tokens, const int32_t *cu_seqlens, int n_sentences, int T,
                     int H, int n_vocab, int max_len, half_t *out, hipStream_t stream) {
    if (T <= 0) return;
    if ((table_type <= 1 ? H % 8 == 0 : H % 32 == 0) && table_type <= 3 && H <= 1024 && max_len > 0) {
        // (group, sentence) grid while at least two thirds of its workgroups have tokens, else one group per 4 tokens
        const bool search = n_sentences > 65535 || 2ll * ((max_len + 3) / 4) * n_sentences > 3ll * ((T + 3) / 4);
        const dim3 g2 = search ? dim3((T + 3) / 4) : dim3((max_len + 3) / 4, n_sentences), b2(256);
#define EMBR(TT, NC) do { \
            if (search) BERT_LAUNCH((embed_ln_rows_kernel<TT, NC, true>), g2, b2, 0, stream, word, type, pos, gamma, beta, tokens, \
                                           cu_seqlens, n_sentences, T, H, n_vocab, out); \
            else BERT_LAUNCH((embed_ln_rows_kernel<TT, NC, false>), g2, b2, 0, stream, word, type, pos, gamma, beta, tokens, \
                                    cu_seqlens, n_sentences, T, H, n_vocab, out); } while (0)
        if (table_type == 0) { if (H <= 512) EMBR(0, 1); else EMBR(0, 2); }
        else if (table_type == 1) { if (H <= 512) EMBR(1, 1); else EMBR(1, 2); }
        else if (table_type == 2) { if (H <= 512) EMBR(2, 1); else EMBR(2, 2); }
        else { if (H <= 512) EMBR(3, 1); else EMBR(3, 2); }
#undef EMBR
        return;
    }
    const dim3 grid((T + 3) / 4), block(256);
    const int nj = (H + 127) / 128;
#define EMB(NJ) BERT_LAUNCH(embed_ln_kernel<NJ>, grid, block, 0, stream, word, type, pos, table_type, gamma, \
                                   beta, tokens, cu_seqlens, n_sentences, T, H, n_vocab, out)
    if (nj <= 1) EMB(1); else if (nj <= 3) EMB(3); else if (nj <= 6) EMB(6); else if (nj <= 8) EMB(8); else EMB(32);
#undef EMB
}

// reference bert.cpp:868-874 / :894-900 (ggml_norm + gamma/beta), in place on f16 rows.
// NJ = pairs per lane held in registers (H <= 128 * NJ).
template <int NJ>
__global__ __launch_bounds__(256) void layernorm_kernel(half_t *x, const float *gamma, const float *beta, int T, int H) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    half_t *row = x + (size_t)t * H;
    float v[NJ][2];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e = 2 * lane + 128 * j;
        if (e < H) {
            const f16x2 h2 = *(const f16x2 *)(row + e);
            v[j][0] = (float)h2[0]; v[j][1] = (float)h2[1];
        } else { v[j][0] = 0.f; v[j][1] = 0.f; }
        sum += v[j][0] + v[j][1];
    }
    const float mean = wave_sum(sum) / H;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e = 2 * lane + 128 * j;
        if (e < H) {
            v[j][0] -= mean; v[j][1] -= mean;
            sq += v[j][0] * v[j][0] + v[j][1] * v[j][1];
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / H + 1e-5f);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e = 2 * lane + 128 * j;
        if (e < H) {
            f16x2 o;
            o[0] = (_Float16)(gamma[e] * (v[j][0] * rstd) + beta[e]);
            o[1] = (_Float16)(gamma[e + 1] * (v[j][1] * rstd) + beta[e + 1]);
            *(f16x2 *)(row + e) = o;
        }
    }
}

// The same for H % 8 == 0: a lane owns 16-byte runs of the row (one 16-byte load and store per 8 features; the two-byte-pair
// form above moved 2.7 TB/s at H = 768), NC runs per lane (H <= 512 * NC), every run of the row requested before the first is
// touched.  Same arithmetic (two passes, eps 1e-5).  5.1 TB/s at H = 768 and 262 144 rows; two or four rows per wave (more
// loads in flight per lane) measured 0 / +2 % time: the kernel is not short of requests.
template <int NC>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(half_t *x, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, int T, int H) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    half_t *row = x + (size_t)t * H;
    f16x8m h[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int e0 = 8 * (lane + 64 * j);
#pragma unroll
        for (int i = 0; i < 8; ++i) h[j][i] = (_Float16)0.f;
        if (e0 < H) h[j] = *(const f16x8m *)(row + e0);
    }
    float v[NC][8];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[j][i] = (float)h[j][i]; sum += v[j][i]; }
    const float mean = wave_sum(sum) / H;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j)
        if (8 * (lane + 64 * j) < H)
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[j][i] -= mean; sq += v[j][i] * v[j][i]; }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / H + 1e-5f);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int e0 = 8 * (lane + 64 * j);
        if (e0 < H) {
            const float4 g0 = *(const float4 *)(gamma + e0), g1 = *(const float4 *)(gamma + e0 + 4);
            const float4 b0 = *(const float4 *)(beta + e0), b1 = *(const float4 *)(beta + e0 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            f16x8m o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (_Float16)(gg[i] * (v[j][i] * rstd) + bb[i]);
            *(f16x8m *)(row + e0) = o;
        }
    }
}

void launch_layernorm(half_t *x, const float *gamma, const float *beta, int T, int H, hipStream_t stream) {
    if (T <= 0) return;
    if (H % 8 == 0 && H <= 2048) {
        const dim3 grid((T + 3) / 4), block(256);
        if (H <= 512) BERT_LAUNCH(layernorm_rows_kernel<1>, grid, block, 0, stream, x, gamma, beta, T, H);
        else if (H <= 1024) BERT_LAUNCH(layernorm_rows_kernel<2>, grid, block, 0, stream, x, gamma, beta, T, H);
        else BERT_LAUNCH(layernorm_rows_kernel<4>, grid, block, 0, stream, x, gamma, beta, T, H);
        return;
    }
    const dim3 grid((T + 3) / 4), block(256);
    const int nj = (H + 127) / 128;      // H must be even (checked at load)
    if (nj <= 1) BERT_LAUNCH(layernorm_kernel<1>, grid, block, 0, stream, x, gamma, beta, T, H);
    else if (nj <= 3) BERT_LAUNCH(layernorm_kernel<3>, grid, block, 0, stream, x, gamma, beta, T, H);
    else if (nj <= 6) BERT_LAUNCH(layernorm_kernel<6>, grid, block, 0, stream, x, gamma, beta, T, H);
    else if (nj <= 8) BERT_LAUNCH(layernorm_kernel<8>, grid, block, 0, stream, x, gamma, beta, T, H);
    else BERT_LAUNCH(layernorm_kernel<32>, grid, block, 0, stream, x, gamma, beta, T, H);   // H <= 4096
}

// LayerNorm folded into the H = 768 mat-muls (kernels.h GemmLnFold): the residual mat-mul's epilogue left P partial (sum, sum of
// squares) pairs per row; the row's {1 / std, - mean / std, - mean, std} (eps 1e-5) for the mat-muls that read the un-normalised rows
__global__ __launch_bounds__(256) void ln_rows_finalize_kernel(const float2 *__restrict__ stats, int P, int T, float inv_h, float4 *__restrict__ rows) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    float s1 = 0.f, s2 = 0.f;
    for (int p = 0; p < P; ++p) { const float2 v = stats[(size_t)t * P + p]; s1 += v.x; s2 += v.y; }
    const float mean = s1 * inv_h, var = fmaxf(s2 * inv_h - mean * mean, 0.f) + 1e-5f;
    const float sd = sqrtf(var), rstd = 1.0f / sd;
    rows[t] = make_float4(rstd, -mean * rstd, -mean, sd);
}

void launch_ln_rows_finalize(const float2 *stats, int P, int T, int H, float4 *rows, hipStream_t stream) {
    if (T <= 0) return;
    BERT_LAUNCH(ln_rows_finalize_kernel, dim3((T + 255) / 256), dim3(256), 0, stream, stats, P, T, 1.0f / H, rows);
}

// reference bert.cpp:904-913: mean over all N tokens (mat-vec with a 1/N vector), then y / ||y||_2.
// One workgroup per sentence (pool_normalize.h: the body, shared with the epilogue of model_kernel.hip).
__global__ __launch_bounds__(256) void pool_normalize_kernel(const half_t *x, const int32_t *cu_seqlens, int H,
                                                             int max_len, int *status, float *out) {
    extern __shared__ float part[];          // [4][H] partial sums, then red[4]
    const int b = blockIdx.x;
    const int tok0 = cu_seqlens[b], n = cu_seqlens[b + 1] - tok0;
    pool_normalize_sentence(x, tok0, n, b, H, max_len, status, out, part, (int)threadIdx.x, true);
}

void launch_pool_normalize(const half_t *x, const int32_t *cu_seqlens, int n_sentences, int H, int max_len, int *status,
                           float *out, hipStream_t stream) {
    if (n_sentences <= 0) return;
    BERT_LAUNCH(pool_normalize_kernel, dim3(n_sentences), dim3(256), (4 * H + 4) * sizeof(float), stream, x,
                       cu_seqlens, H, max_len, status, out);
}

// A call's staged block (ids | cu_seqlens | windows) from MAPPED pinned host memory into device memory by a kernel: the copy
// engine needs about 20 us before the first kernel behind it can start, a few workgroups reading 16 bytes per lane across the host
// link need 5-8 for the 130 KB of a 256 x 128 batch (engine.hip eval_packed_host; small blocks only).
__global__ __launch_bounds__(256) void stage_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int n16) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

void launch_stage_copy(const void *mapped_src, void *dst, size_t bytes, hipStream_t stream) {
    const int n16 = (int)((bytes + 15) / 16);
    if (n16 <= 0) return;
    BERT_LAUNCH(stage_copy_kernel, dim3(std::min(64, (n16 + 255) / 256)), dim3(256), 0, stream, (const uint4 *)mapped_src, (uint4 *)dst, n16);
}

__global__ void f16_to_f32_kernel(const half_t *src, float *dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

void launch_f16_to_f32(const half_t *src, float *dst, size_t n, hipStream_t stream) {
    if (!n) return;
    BERT_LAUNCH(f16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, dst, n);
}

}  // namespace bert_hip
