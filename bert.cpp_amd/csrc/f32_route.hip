// f32_route.hip — the forward pass of f32 model files (ftype 0) in f32 arithmetic (gfx950).
//
// The reference evaluates an f32 file with ggml's f32 mat-mul: f32 weights x f32 activations, f32 accumulation (reference
// bert.cpp:825 `ggml_mul_mat` on GGML_TYPE_F32 tensors; the tensors get that type at bert.cpp:407-429 for ftype 0), f32
// LayerNorm, softmax and GELU.  Rounds 1-4 rounded such files to f16 at load and ran them through the f16 kernels — the one
// place where this engine computed in NARROWER arithmetic than the reference.  This file is the route that does not: f32
// activations in HBM, every weight mat-mul on v_mfma_f32_32x32x2_f32 (the matrix cores' f32 form: products and sums are
// f32 fma chains, 157 TFLOP/s dense), f32 softmax / tanh-GELU / LayerNorm.  It is a precision route, not a benchmark
// configuration (BASELINE.json's configs are f16 / q4 files): plain LDS-tiled kernels, one launch per operation, any
// H / d_head / length the file format allows.  `BERT_HIP_F32=f16` (or set_option "f32" = "f16") selects the old behaviour
// (f16 operands, the fused kernels) for f32 files.
//
// Replaces, for f32 files: bert.cpp:796-814 (embedding + LayerNorm), :822-839 / :859-865 / :878-891 (mat-muls + bias, GELU,
// residual), :843-856 (attention), :868-874 / :894-900 (LayerNorm), :904-913 (pooling).
#include "kernels.h"

#include <algorithm>

namespace bert_hip {

typedef float f32x16r __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float f32_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// LayerNorm of one f32 row held in global memory, by one wave: two passes for the statistics (mean, then the centred sum of
// squares: ggml_norm's order), eps 1e-5, gamma * x + beta.
__device__ __forceinline__ void f32_layernorm_row(float *row, const float *gamma, const float *beta, int H, int lane) {
    float sum = 0.f;
    for (int e = lane; e < H; e += 64) sum += row[e];
    const float mean = f32_wave_sum(sum) / (float)H;
    float sq = 0.f;
    for (int e = lane; e < H; e += 64) { const float d = row[e] - mean; sq += d * d; }
    const float rstd = 1.0f / sqrtf(f32_wave_sum(sq) / (float)H + 1e-5f);
    for (int e = lane; e < H; e += 64) row[e] = gamma[e] * ((row[e] - mean) * rstd) + beta[e];
}

// reference bert.cpp:796-814: inpL = word[ids]; inpL = type[0] + inpL; inpL = pos[0..N-1] + inpL; LayerNorm.  One wave per token.
__global__ __launch_bounds__(256) void f32_embed_ln_kernel(const float *word, const float *type, const float *pos, const float *gamma,
                                                           const float *beta, const int32_t *__restrict__ tokens,
                                                           const int32_t *__restrict__ cu_seqlens, int n_sentences, int T, int H, int n_vocab,
                                                           float *out) {
    const int t = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    int lo = 0, hi = n_sentences;                             // sentence of token t: largest b with cu[b] <= t
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cu_seqlens[mid] <= t) lo = mid; else hi = mid;
    }
    const int p = t - cu_seqlens[lo];
    int id = tokens[t];
    id = id < 0 ? 0 : (id >= n_vocab ? n_vocab - 1 : id);     // ids are validated on the host API; clamp for safety
    float *row = out + (size_t)t * H;
    for (int e = lane; e < H; e += 64) row[e] = pos[(size_t)p * H + e] + (type[e] + word[(size_t)id * H + e]);
    f32_layernorm_row(row, gamma, beta, H, lane);
}

__global__ __launch_bounds__(256) void f32_layernorm_kernel(float *x, const float *gamma, const float *beta, int T, int H) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    f32_layernorm_row(x + (size_t)t * H, gamma, beta, H, lane);
}

// C[t][n] = epi( sum_k A[t][k] * W[n][k] + bias[n] (+ resid[t][n]) ), everything f32; any M, N, K.
// Workgroup = 4 waves = 64 tokens x 64 features; a wave owns 32 x 32 (one accumulator block of v_mfma_f32_32x32x2_f32, computed
// "swapped" like every mat-mul of this library: A operand = weight rows, B operand = token rows, so a lane's accumulator
// column is one token and its registers are runs of 4 consecutive features: 16-byte stores).  Reduction tiles of 16 through
// LDS, rows padded to 17 floats (conflict-free fragment reads: a lane reads element [row l31][2 kk + hi]).
constexpr int F32_BM = 64, F32_BN = 64, F32_BK = 16, F32_LD = F32_BK + 1;
__device__ __forceinline__ float f32_gelu(float x) {
    // ggml's GELU (tanh form): 0.5 x (1 + tanh(sqrt(2/pi) x (1 + 0.044715 x^2)))
    return 0.5f * x * (1.0f + tanhf(0.79788456080286535588f * x * (1.0f + 0.044715f * x * x)));
}
template <int EPI>
__global__ __launch_bounds__(256) void f32_gemm_kernel(const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
                                                       const float *__restrict__ resid, float *__restrict__ C, int M, int N, int K) {
    __shared__ float As[F32_BM * F32_LD], Ws[F32_BN * F32_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.y * F32_BM, n0 = blockIdx.x * F32_BN;
    const int tb = wave & 1, fb = wave >> 1;
    f32x16r acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // staging: thread -> (row = tid / 4, 4 consecutive k = 4 (tid % 4)) of both tiles
    const int srow = tid >> 2, sk = (tid & 3) * 4;
    const bool k4 = (K & 3) == 0;
    for (int k0 = 0; k0 < K; k0 += F32_BK) {
        float av[4] = {0.f, 0.f, 0.f, 0.f}, wv[4] = {0.f, 0.f, 0.f, 0.f};
        const int am = m0 + srow, wn = n0 + srow, kk0 = k0 + sk;
        if (am < M) {
            if (k4 && kk0 + 3 < K) {
                const float4 v = *(const float4 *)(A + (size_t)am * K + kk0);
                av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (kk0 + i < K) av[i] = A[(size_t)am * K + kk0 + i];
            }
        }
        if (wn < N) {
            if (k4 && kk0 + 3 < K) {
                const float4 v = *(const float4 *)(W + (size_t)wn * K + kk0);
                wv[0] = v.x; wv[1] = v.y; wv[2] = v.z; wv[3] = v.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (kk0 + i < K) wv[i] = W[(size_t)wn * K + kk0 + i];
            }
        }
        __syncthreads();                                   // the previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) { As[srow * F32_LD + sk + i] = av[i]; Ws[srow * F32_LD + sk + i] = wv[i]; }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < F32_BK / 2; ++kk) {
            const float wf = Ws[(fb * 32 + l31) * F32_LD + 2 * kk + hi];
            const float af = As[(tb * 32 + l31) * F32_LD + 2 * kk + hi];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf, af, acc, 0, 0, 0);
        }
    }
    // accumulator register r of lane (l31, hi): token m0 + 32 tb + l31, feature n0 + 32 fb + (r & 3) + 8 (r >> 2) + 4 hi
    const int m = m0 + tb * 32 + l31;
    if (m >= M) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = n0 + fb * 32 + 8 * g + 4 * hi;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = 0.f;
            if (n + e < N) {
                v[e] = acc[4 * g + e] + bias[n + e];
                if (EPI == EPI_BIAS_GELU) v[e] = f32_gelu(v[e]);
                if (EPI == EPI_BIAS_RESID) v[e] += resid[(size_t)m * N + n + e];
            }
        }
        if ((N & 3) == 0 && n + 3 < N) *(float4 *)(C + (size_t)m * N + n) = float4{v[0], v[1], v[2], v[3]};
        else
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < N) C[(size_t)m * N + n + e] = v[e];
    }
}

void launch_f32_gemm(const float *A, const float *W, const float *bias, const float *resid, float *C, int M, int N, int K, int epilogue,
                     hipStream_t stream) {
    if (M <= 0 || N <= 0) return;
    const dim3 grid((N + F32_BN - 1) / F32_BN, (M + F32_BM - 1) / F32_BM), block(256);
    if (epilogue == EPI_BIAS) BERT_LAUNCH(f32_gemm_kernel<EPI_BIAS>, grid, block, 0, stream, A, W, bias, resid, C, M, N, K);
    else if (epilogue == EPI_BIAS_GELU) BERT_LAUNCH(f32_gemm_kernel<EPI_BIAS_GELU>, grid, block, 0, stream, A, W, bias, resid, C, M, N, K);
    else BERT_LAUNCH(f32_gemm_kernel<EPI_BIAS_RESID>, grid, block, 0, stream, A, W, bias, resid, C, M, N, K);
}

// reference bert.cpp:843-856 on f32 Q | K | V rows: one wave per (sentence, head, query); scores of the wave's query in LDS;
// softmax with the true maximum subtracted, exponentials and sums in f32; no mask (a sentence attends over its own tokens).
__global__ void f32_attention_kernel(const float *qkv, const int32_t *cu_seqlens, int n_head, int d, float *out) {
    extern __shared__ float sh[];          // [4 waves][max_len rounded up to 4] scores
    const int b = blockIdx.y, h = blockIdx.z;
    const int tok0 = cu_seqlens[b], n = cu_seqlens[b + 1] - tok0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + wave;
    if (q >= n) return;
    const int H = n_head * d, ld = 3 * H;
    float *s = sh + (size_t)wave * gridDim.x * 4;
    const float *qp = qkv + (size_t)(tok0 + q) * ld + h * d;
    const float scale = 1.0f / sqrtf((float)d);
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) {
        const float *kp = qkv + (size_t)(tok0 + j) * ld + H + h * d;
        float a = 0.f;
        for (int e = 0; e < d; ++e) a = fmaf(kp[e], qp[e], a);
        a *= scale;
        s[j] = a;
        mx = fmaxf(mx, a);
    }
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) { const float p = expf(s[j] - mx); s[j] = p; sum += p; }
    sum = f32_wave_sum(sum);
    const float inv = 1.0f / sum;
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < d; e += 64) {
        float a = 0.f;
        for (int j = 0; j < n; ++j) a = fmaf(qkv[(size_t)(tok0 + j) * ld + 2 * H + h * d + e], s[j], a);
        out[(size_t)(tok0 + q) * H + h * d + e] = a * inv;
    }
}

// reference bert.cpp:904-913: mean over all N tokens (a mat-vec with a 1/N vector: every term is x * (1/N)), y / ||y||_2 without
// epsilon.  One workgroup per sentence; a sentence outside [1, max_len] gets a NaN row and raises the status word (the
// device API's promise, as launch_pool_normalize).
__global__ __launch_bounds__(256) void f32_pool_normalize_kernel(const float *x, const int32_t *cu_seqlens, int H, int max_len, int *status,
                                                                 float *out) {
    extern __shared__ float part[];          // [H] pooled row, then red[4]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int tok0 = cu_seqlens[b], n = cu_seqlens[b + 1] - tok0;
    if (n <= 0 || n > max_len) {
        for (int e = tid; e < H; e += 256) out[(size_t)b * H + e] = __builtin_nanf("");
        if (tid == 0 && status) atomicOr(status, 1);
        return;
    }
    const float invn = 1.0f / (float)n;
    float sq = 0.f;
    for (int e = tid; e < H; e += 256) {
        float a = 0.f;
        for (int t = 0; t < n; ++t) a += x[(size_t)(tok0 + t) * H + e] * invn;
        part[e] = a;
        sq += a * a;
    }
    sq = f32_wave_sum(sq);
    float *red = part + H;
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    const float scale = 1.0f / sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    for (int e = tid; e < H; e += 256) out[(size_t)b * H + e] = part[e] * scale;
}

void launch_f32_embed_ln(const float *word, const float *type, const float *pos, const float *gamma, const float *beta, const int32_t *tokens,
                         const int32_t *cu_seqlens, int n_sentences, int T, int H, int n_vocab, float *out, hipStream_t stream) {
    if (T <= 0) return;
    BERT_LAUNCH(f32_embed_ln_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, word, type, pos, gamma, beta, tokens, cu_seqlens, n_sentences, T, H,
                n_vocab, out);
}

void launch_f32_layernorm(float *x, const float *gamma, const float *beta, int T, int H, hipStream_t stream) {
    if (T <= 0) return;
    BERT_LAUNCH(f32_layernorm_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, x, gamma, beta, T, H);
}

void launch_f32_attention(const float *qkv, const int32_t *cu_seqlens, int n_sentences, int n_head, int d_head, int max_len, float *out,
                          hipStream_t stream) {
    const int qblocks = (max_len + 3) / 4;
    const size_t lds = (size_t)4 * qblocks * 4 * sizeof(float);
    for (int b0 = 0; b0 < n_sentences; b0 += 65535) {          // (a grid dimension holds 65535 sentences)
        const dim3 grid(qblocks, std::min(65535, n_sentences - b0), n_head);
        BERT_LAUNCH(f32_attention_kernel, grid, dim3(256), lds, stream, qkv, cu_seqlens + b0, n_head, d_head, out);
    }
}

void launch_f32_pool_normalize(const float *x, const int32_t *cu_seqlens, int n_sentences, int H, int max_len, int *status, float *out,
                               hipStream_t stream) {
    if (n_sentences <= 0) return;
    BERT_LAUNCH(f32_pool_normalize_kernel, dim3(n_sentences), dim3(256), (H + 4) * sizeof(float), stream, x, cu_seqlens, H, max_len, status, out);
}

}  // namespace bert_hip
