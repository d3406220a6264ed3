// kernels.h — launch interface of the HIP kernels of the bert_eval hot path (gfx950 only).
//
// Reference ops each kernel replaces are listed in SURVEY.md §2.3; call sites bert.cpp:796-913.
// Activations are f16 row-major [T_pad][features]; T_pad is T rounded up to GEMM_BM tokens.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <cstdint>
#include <thread>

namespace bert_hip {

typedef _Float16 half_t;

// An f32 value the compiler must materialise: (_Float16)rounded_f32(a * b) is an f32 multiply followed by a conversion,
// never the fused v_fma_mixlo_f16 (ONE rounding, to f16).  Which of the two forms the compiler picks for (_Float16)(a * b)
// depends on the surrounding code, and kernels that must agree bit for bit (attention.hip, qkv_attention*.hip) would
// differ in one result of ~20 000.
__device__ __forceinline__ float rounded_f32(float v) {
    asm("" : "+v"(v));
    return v;
}

// ---- lane-crossing reductions without the LDS.  __shfl_xor is a ds_bpermute_b32: an LDS round trip (and an lgkmcnt wait) per step.
// The same PAIRS meet here — so sums and maxima keep their bits — through v_permlane32_swap (lane ^ 32: the two halves of the wave
// trade places), ds_swizzle (lane ^ 16 / 8 / 4: no address register, no LDS access) and DPP quad_perm on the add itself (lane ^ 2 / 1).
// tools/ubench/wave_sum.hip checks the six-step sum against the __shfl_xor butterfly bit for bit.
// xor32_pair: a = this lane's value, b = lane ^ 32's in the low half of the wave and the other way round in the high half — fine for
// commutative uses (a + b, max(a, b)).  By hand: this compiler's __builtin_amdgcn_permlane32_swap returns its first result twice; the
// instruction needs two registers (with one as both operands it copies the low half up and loses the high one); the wait states
// between a VALU write and a lane-crossing read are ours inside an asm.
__device__ __forceinline__ void xor32_pair(float &a, float &b) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xor32_sum(float v) { float a = v, b = v; xor32_pair(a, b); return a + b; }
__device__ __forceinline__ float xor32_max(float v) { float a = v, b = v; xor32_pair(a, b); return __builtin_fmaxf(a, b); }
// v + (lane ^ 32) + ... + (lane ^ 1), the pairs and the order of the __shfl_xor butterfly from 32 down to 1
__device__ __forceinline__ float wave_sum_f32(float v) {
    v = xor32_sum(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (16 << 10) | 0x1F));
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (8 << 10) | 0x1F));
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (4 << 10) | 0x1F));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    return v;
}

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

// Softmax numerators of EIGHT scores of one query (registers 8 st .. 8 st + 7 of an S^T tile = the B fragment of one P·V MFMA
// step) — ONE function for all attention bodies (attention.hip, qkv_attention2.hip and with it model_kernel.hip): equal bits
// across the routes.  Two forms:
//   BERT_HIP_EXP16 = 0 (default): p = exp2(fma(s, sc, -m)) in f32 (v_fma_f32, v_exp_f32), the row sum an f32 add, P rounded to
//     f16 for the V mat-mul (v_cvt_pk_f16_f32) — rounds 1-4's arithmetic.
//   BERT_HIP_EXP16 = 1: the reference's own precision — ggml's soft_max rounds (s - max) to fp16 and reads an fp16 table of exp
//     (reference bert.cpp:845 -> ggml_soft_max; oracle/bert_oracle.cpp:544): the argument one fma rounded ONCE to f16
//     (v_fma_mixlo / mixhi_f16 write the two halves of a register), v_exp_f16 on each half (the high one through SDWA with the
//     low half preserved), the pair IS the MFMA operand, the row sum f32 through v_dot2c_f32_f16 with (1, 1).  21 instructions
//     per 8 scores against 28 — and SLOWER on this chip (round 5, tools/ubench/valu_cost.hip, profiles/r5_valu_cost.txt):
//     v_fma_mix* issue at the transcendental rate (7.9 cycles per instance and wave beside MFMAs, 2 waves per SIMD, against 2.5
//     for v_fma_f32) and v_dot2c shares the matrix pipe (8.6 against 1.9 for v_add_f32): 40 cycles per score pair against 27.
//     Measured end to end: headline 327.0 k against 330.5 k sentences/s on one box, attention at 512 tokens 9.28 against 8.68 ms
//     per 12 launches.  Parity-green (395 tests) and kept as a build option; not the default.
// The trailing s_nop of the fp16 form: gfx940+ needs one wait state between an instruction that writes half a register (SDWA
// dst_sel) and a reader of that register, and the compiler's hazard pass does not look into an asm block.
#ifndef BERT_HIP_EXP16
#define BERT_HIP_EXP16 0
#endif
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x8_t __attribute__((ext_vector_type(8)));
// the three steps of softmax_p8 on four score PAIRS, separately callable so that a kernel can put MFMAs between them
// (attention.hip's software-pipelined chunk loop): arguments (8 VALU), exponentials (8 VALU + the wait state), row sum (4 dot2);
// softmax_pack: the B fragment of the P·V MFMA step (the exponentials themselves in the fp16 form).
// -DBERT_HIP_EXP16=0 (tuning builds): the f32 form of rounds 1-4 — fma, v_exp_f32, add, v_cvt_pk_f16_f32.
#if BERT_HIP_EXP16
typedef u32x4_t sm_arg_t;
typedef u32x4_t sm_exp_t;
__device__ __forceinline__ sm_arg_t softmax_args4(float s0, float s1, float s2, float s3, float s4, float s5, float s6, float s7, float sc, float m) {
    uint32_t a0, a1, a2, a3;            // (scalar outputs: asm outputs that are elements of a vector come out wrong)
    asm("v_fma_mixlo_f16 %0, %4, %12, -%13\n\t"
        "v_fma_mixlo_f16 %1, %6, %12, -%13\n\t"
        "v_fma_mixlo_f16 %2, %8, %12, -%13\n\t"
        "v_fma_mixlo_f16 %3, %10, %12, -%13\n\t"
        "v_fma_mixhi_f16 %0, %5, %12, -%13\n\t"
        "v_fma_mixhi_f16 %1, %7, %12, -%13\n\t"
        "v_fma_mixhi_f16 %2, %9, %12, -%13\n\t"
        "v_fma_mixhi_f16 %3, %11, %12, -%13"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
        : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(s6), "v"(s7), "s"(sc), "v"(m));
    return u32x4_t{a0, a1, a2, a3};
}
__device__ __forceinline__ sm_exp_t softmax_exp4(sm_arg_t a) {
    uint32_t p0, p1, p2, p3;
    asm("v_exp_f16_e32 %0, %4\n\t"
        "v_exp_f16_e32 %1, %5\n\t"
        "v_exp_f16_e32 %2, %6\n\t"
        "v_exp_f16_e32 %3, %7\n\t"
        "v_exp_f16_sdwa %0, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
        "v_exp_f16_sdwa %1, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
        "v_exp_f16_sdwa %2, %6 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
        "v_exp_f16_sdwa %3, %7 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
        "s_nop 0"
        : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
    return u32x4_t{p0, p1, p2, p3};
}
__device__ __forceinline__ void softmax_sum4(sm_exp_t p, float &psum) {
    const f16x2_t one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t pe = p[e];       // (a scalar copy: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever e is)
        psum = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, pe), one, psum, false);
    }
}
__device__ __forceinline__ f16x8_t softmax_pack(sm_exp_t p) { return __builtin_bit_cast(f16x8_t, p); }
#else
typedef f32x8_t sm_arg_t;
typedef f32x8_t sm_exp_t;
__device__ __forceinline__ sm_arg_t softmax_args4(float s0, float s1, float s2, float s3, float s4, float s5, float s6, float s7, float sc, float m) {
    const float s[8] = {s0, s1, s2, s3, s4, s5, s6, s7};
    f32x8_t a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = __builtin_fmaf(s[e], sc, -m);
    return a;
}
__device__ __forceinline__ sm_exp_t softmax_exp4(sm_arg_t a) {
    f32x8_t p;
#pragma unroll
    for (int e = 0; e < 8; ++e) p[e] = __builtin_amdgcn_exp2f(a[e]);
    return p;
}
__device__ __forceinline__ void softmax_sum4(sm_exp_t p, float &psum) {
#pragma unroll
    for (int e = 0; e < 8; ++e) psum += p[e];
}
__device__ __forceinline__ f16x8_t softmax_pack(sm_exp_t p) {
    f16x8_t pf;
#pragma unroll
    for (int e = 0; e < 8; ++e) pf[e] = (_Float16)p[e];
    return pf;
}
#endif
__device__ __forceinline__ f16x8_t softmax_p8(float s0, float s1, float s2, float s3, float s4, float s5, float s6, float s7, float sc,
                                              float m, float &psum) {
    const sm_exp_t p = softmax_exp4(softmax_args4(s0, s1, s2, s3, s4, s5, s6, s7, sc, m));
    softmax_sum4(p, psum);
    return softmax_pack(p);
}
// tanh-form GELU of two values, packed f16: x / (1 + 2^(x (c1 + c2 x^2)))
__device__ __forceinline__ f16x2_t gelu_pk16h(f16x2_t xh) {
    const float c1 = -2.0f * 0.79788456080286535588f * 1.44269504088896340736f;
    const f16x2_t C1 = {(_Float16)c1, (_Float16)c1}, C2 = {(_Float16)(c1 * 0.044715f), (_Float16)(c1 * 0.044715f)};
    const f16x2_t one = {(_Float16)1.0f, (_Float16)1.0f};
    const f16x2_t t = (xh * xh * C2 + C1) * xh;
    const f16x2_t e = {(_Float16)__builtin_exp2f16(t[0]), (_Float16)__builtin_exp2f16(t[1])};
    const f16x2_t d = e + one;
    const f16x2_t r = {(_Float16)__builtin_amdgcn_rcph(d[0]), (_Float16)__builtin_amdgcn_rcph(d[1])};
    return xh * r;
}
__device__ __forceinline__ f16x2_t gelu_pk16(float a0, float a1) { return gelu_pk16h(f16x2_t{(_Float16)a0, (_Float16)a1}); }

// LayerNorm statistics (sum, sum of squares over H values) -> (1 / std, -mean / std), eps 1e-5 (ggml's).  Every product and
// sum is spelled out: with -ffp-contract=fast the compiler picks which a * b + c it fuses by the surrounding code, and kernels
// that must agree bit for bit (layer_tail.hip and its feature-split mirror skinny.hip) share this function instead.
// 1 / sqrt = v_rsq_f32 + one Newton step (1 ulp).
__device__ __forceinline__ void layernorm_scale(float s1, float s2, float inv_h, float &rstd, float &nmr) {
    const float mean = s1 * inv_h, ex2 = s2 * inv_h;
    const float t = fmaxf(__builtin_fmaf(-mean, mean, ex2), 0.f) + 1e-5f;
    const float r = __builtin_amdgcn_rsqf(t);
    rstd = r * __builtin_fmaf(-0.5f * t, r * r, 1.5f);
    nmr = -mean * rstd;
}

constexpr int GEMM_BM = 128;   // token tile
constexpr int GEMM_BN = 128;   // feature tile
constexpr int GEMM_BK = 64;    // reduction tile (two 32-weight quant blocks)

enum Epilogue : int { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESID = 2 };
enum GemmWType : int { GW_F16 = 0, GW_Q4_0 = 1, GW_Q4_1 = 2 };

// A weight matrix W[N][K] (out-features x in-features) in its HBM layout.
//   GW_F16 : w16  [N_pad][K] f16 row-major (rows N..N_pad are zero)
//   GW_Q4_x: qs   one uint4 (16 B of nibbles) per 32-weight block, tile-contiguous:
//                 index = ((nt * (K/64) + kt) * 128 + row_in_tile) * 2 + block_in_ktile
//            sc   f16 scale per block (q4_0) or f16x2 {d, m} per block (q4_1), same order
//   naive16: optional f16 [N][K] row-major dequantised copy used by the generic fallback kernel
struct GemmWeight {
    int N = 0, K = 0, N_pad = 0;
    int type = GW_F16;
    const half_t *w16 = nullptr;
    const uint4 *qs = nullptr;
    const void *sc = nullptr;
    const half_t *naive16 = nullptr;
    // optional second f16 image whose k order inside every group of 16 is [0-3, 8-11, 4-7, 12-15]: a 16-byte half
    // of a group then holds the k's one lane of an MFMA accumulator column owns (layer_tail.hip)
    const half_t *w16p = nullptr;
    // f32 files (ftype 0) with the f32 route (f32_route.hip): the file's own f32 rows, [N][K] row-major
    const float *w32 = nullptr;
};

// C[t][n] = epi( sum_k A[t][k] * W[n][k] + bias[n] (+ resid[t][n]) ), t < M_pad (multiple of 128)
// lda = K, ldc = ldr = N.  MFMA path requires K % 64 == 0.
void launch_gemm_mfma(const GemmWeight &W, const half_t *A, const float *bias, const half_t *resid, half_t *C,
                      int M_pad, int epilogue, hipStream_t stream);
// Large-tile variant (gemm256.hip): 256 x 256 x 64 tiles, 8 waves; f16 weights, N % 256 == 0, M_pad % 256 == 0.
bool gemm256_supported(const GemmWeight &W, int M_pad);
// LayerNorm folded into the mat-muls around it (H = 768 route; reference bert.cpp:866-875, :892-901 have it as operations of their
// own): a residual mat-mul writes the UN-normalised sum u (f16) and per-row partial statistics of it (STATS); ln_rows_finalize turns
// those into {rstd, -mean rstd, -mean, std} per row; a mat-mul that consumes LayerNorm(u) reads u itself with gamma folded into its
// weights and ONE extra k-step that carries - mean s[n] + std c[n] (s = row sums of the folded weights, c = W beta + bias), its
// epilogue multiplying by rstd (IN); a residual mat-mul whose residual is LayerNorm(u) rebuilds it per element from u, the row
// statistics and packed (gamma, beta + bias) pairs (RES).  f16 images only.
struct GemmLnFold {
    enum : int { IN = 1, RES = 2, STATS = 4 };
    int flags = 0;
    const float4 *rows_in = nullptr;      // IN: statistics of A's rows
    const half_t *waug = nullptr;         // IN: [N][16] f16 weight side of the statistics k-step
    const float4 *rows_res = nullptr;     // RES: statistics of resid's rows
    const unsigned *gb = nullptr;         // RES: [N] f16 gamma | f16 (beta + bias) << 16
    float2 *stats = nullptr;              // STATS: [M_pad][2 N / 256] (sum, sum of squares)
};
void launch_gemm256(const GemmWeight &W, const half_t *A, const float *bias, const half_t *resid, half_t *C, int M_pad,
                    int epilogue, hipStream_t stream, const GemmLnFold *ln = nullptr);
// per-row partial statistics [T][P] -> {rstd, -mean rstd, -mean, std} (eps 1e-5, H features per row)
void launch_ln_rows_finalize(const float2 *stats, int P, int T, int H, float4 *rows, hipStream_t stream);
// Out-projection + LN + FFN + LN in one launch (layer_tail.hip): a pair of specialist waves per 32 tokens (up-projection +
// GELU / down-projection); H = 256 / 384; f16 weights (W1 / W2 need w16p) or q4 planes.
bool layer_tail_supported(const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2);
void launch_layer_tail(const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2, const half_t *ctx, const half_t *x,
                       const float *bo, const float *g1, const float *be1, const float *b1, const float *b2,
                       const float *g2, const float *be2, half_t *out, int M_pad, hipStream_t stream);
// All encoder layers of a batch in one launch (model_kernel.hip): a workgroup carries its window (whole sentences, at most 128
// tokens between them) through every layer — the window kernel's and the layer tail's bodies as phases, same bits.
struct ModelLayerWeights {
    const GemmWeight *Wqkv, *Wo, *W1, *W2;
    const float *bqkv, *bo, *g1, *be1, *b1, *b2, *g2, *be2;
};
bool model_kernel_supported(const GemmWeight &Wqkv, const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2, int n_layer,
                            int n_head, int d_head, int max_len);
// x: in = embeddings + LayerNorm, out = the last layer's output; ctx: workspace [T][H].  groups / n_groups / n_groups_dev: the
// window list as for launch_qkv_attention2 (nullptr: one sentence per window); n_tokens = 128 n_sentences selects the form
// specialised for full windows.  pooled != nullptr: the workgroups also pool and normalise their sentences (launch_pool_normalize's
// arguments and bits: [n_sentences][H] f32, max_len, status word).
void launch_model_kernel(const ModelLayerWeights *layers, int n_layer, half_t *x, half_t *ctx, const int32_t *cu_seqlens, int n_sentences,
                         int n_tokens, const int2 *groups, int n_groups, const int *n_groups_dev, int n_head, float *pooled, int max_len,
                         int *status, int slots, hipStream_t stream);
// The latency route (skinny.hip): the weight mat-muls of a layer split by output features AND token blocks over up to 192
// one-wave workgroups, for batches of at most 128 tokens; same bits per sentence as qkv_attention2 + layer_tail.
// mode: 0 QKV projection (-> f16), 1 out-projection (+ x + bo -> f32), 2 up-projection + GELU (-> f16, fragment order),
// 3 down-projection (+ y + b2 -> f32).  V != nullptr (modes 0, 2): the kernel LayerNorms the f32 rows V itself (gamma, beta)
// and writes them to ln_out; else A holds the f16 token rows.  n_token_blocks = ceil(T / 32) <= 4.
bool skinny_layer_supported(const GemmWeight &Wqkv, const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2);
void launch_skinny_gemm(int mode, const GemmWeight &W, const half_t *A, const float *V, const float *gamma, const float *beta,
                        half_t *ln_out, const float *bias, const half_t *resid, half_t *out16, float *out32, int n_token_blocks,
                        hipStream_t stream);
// the last layer's LayerNorm 2: f32 rows -> f16 rows
void launch_skinny_layernorm(const float *v, const float *gamma, const float *beta, half_t *out, int n_token_blocks, int H,
                             hipStream_t stream);
// Generic fallback (any K, N); needs W.naive16.
void launch_gemm_naive(const GemmWeight &W, const half_t *A, const float *bias, const half_t *resid, half_t *C,
                       int M, int epilogue, hipStream_t stream);

// word + token_type(0) + position gather-sum, LayerNorm(eps 1e-5), f16 out.  Tables in file layout.
void launch_embed_ln(const void *word, const void *type, const void *pos, int table_type, const float *gamma,
                     const float *beta, const int32_t *tokens, const int32_t *cu_seqlens, int n_sentences, int T,
                     int H, int n_vocab, int max_len, half_t *out, hipStream_t stream);

// In-place LayerNorm over H of f16 rows (eps 1e-5), gamma/beta f32.
void launch_layernorm(half_t *x, const float *gamma, const float *beta, int T, int H, hipStream_t stream);

// qkv [T_pad][3H] (Q | K | V), packed sentences; out ctx [T_pad][H].
// MFMA path supports d_head in {32, 64}; returns false if the shape needs the naive kernel.
bool launch_attention_mfma(const half_t *qkv, const int32_t *cu_seqlens, int n_sentences, int n_head, int d_head,
                           int max_len, half_t *out, hipStream_t stream);
void launch_attention_naive(const half_t *qkv, const int32_t *cu_seqlens, int n_sentences, int n_head, int d_head,
                            int max_len, half_t *out, hipStream_t stream);

// Q|K|V projection + attention in one kernel (qkv_attention2.hip): x [T_pad][H] -> ctx [T_pad][H].  A workgroup owns a window
// of 128 token slots holding one or several whole sentences (16-slot aligned); f16 or q4 weights, d_head 32, H = 128 / 256 / 384, every sentence <= 128 tokens.
// `groups` [n_groups] = {first sentence, count} per window (device memory), or nullptr: the uniform rule
// 128 / round_up(max_len, 16) sentences per window.
bool qkv_attention2_supported(const GemmWeight &Wqkv, int n_head, int d_head, int max_len);
int qkv_attention2_sentences_per_window(int max_len, int slots);
// n_groups_dev (optional): device word with the real number of windows when `groups` was built on the device and n_groups is
// only an upper bound (launch_build_windows / qkv_attention2_max_windows).
void launch_qkv_attention2(const GemmWeight &Wqkv, const half_t *x, const float *bias, const int32_t *cu_seqlens,
                           int n_sentences, const int2 *groups, int n_groups, const int *n_groups_dev, int max_len, int n_head,
                           int slots, half_t *out, hipStream_t stream);
// next-fit windows (the rule of Engine::build_windows) computed on the device from cu_seqlens; *n_windows receives their number
void launch_build_windows(const int32_t *cu_seqlens, int n_sentences, int2 *windows, int *n_windows, int slots, hipStream_t stream);
int qkv_attention2_max_windows(int n_sentences, int n_tokens, int slots);
// place granularity of the windows (16, or 8: qkv_attention2.hip): the process-wide default; every function above takes the
// value its caller read once per forward pass (`slots`)
int window_slots();
void set_window_slots(int slots);

// mean over the sentence's tokens, then L2 normalise; out f32 [n_sentences][H].  A sentence whose length is not in
// [1, max_len] (the caller of the device API promised max_len) gets a NaN row and sets *status (device word) to 1.
void launch_pool_normalize(const half_t *x, const int32_t *cu_seqlens, int n_sentences, int H, int max_len, int *status,
                           float *out, hipStream_t stream);

// The > 64 KiB dynamic-LDS opt-in (hipFuncSetAttribute) is per device: `seen` is the launcher's per-kernel record.  The
// devices of a context launch from threads of their own, and two contexts may share a device: the record is atomic, and
// the first launcher on a device finishes the opt-in (`mark_configured`) before anybody else launches past it.
constexpr int MAX_HIP_DEVICES = 64;
typedef std::atomic<int> DeviceFlags[MAX_HIP_DEVICES];        // 0 = not configured, 1 = being configured, 2 = done
inline int current_device_slot() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d & (MAX_HIP_DEVICES - 1);
}
// true for exactly one caller per device: it runs the opt-in and then calls mark_configured(); everybody else waits for that
inline bool first_launch_on_device(DeviceFlags &seen) {
    std::atomic<int> &f = seen[current_device_slot()];
    int expected = 0;
    if (f.load(std::memory_order_acquire) == 2) return false;
    if (f.compare_exchange_strong(expected, 1, std::memory_order_acq_rel)) return true;
    while (f.load(std::memory_order_acquire) != 2) std::this_thread::yield();
    return false;
}
inline void mark_configured(DeviceFlags &seen) { seen[current_device_slot()].store(2, std::memory_order_release); }
template <class F>
inline void configure_once(DeviceFlags &seen, F &&opt_in) {
    if (first_launch_on_device(seen)) { opt_in(); mark_configured(seen); }
}

// Every kernel of the path is launched through BERT_LAUNCH.  While the engine profiles (Engine::timed) the launching thread
// points tl_launch_timing at an event pair and the launch goes through hipExtLaunchKernelGGL, which attaches the events to the
// dispatch itself (no hipEventRecord barrier packets of their own in the stream).  Measured (round 4): a launch timed alone
// still reads long — model_kernel 825-866 us against 780 us in rocprofv3's trace of the same steps — whatever the events'
// fence flags; for kernels of a millisecond and more the two agree within 1 %.  These times feed the per-kernel BREAKDOWN; the
// roofline's kernel time comes from replay groups (engine.hip timed(), bench.py kernel_roofline).
// `launches` counts the launches a timed body issued: the event pair is only meaningful for exactly one (a body that returns
// without launching leaves stale timestamps in pooled events, a body with two launches measures the last one).
struct LaunchTiming { hipEvent_t start, stop; int launches; };
inline thread_local LaunchTiming *tl_launch_timing = nullptr;
#define BERT_LAUNCH(kernel, grid, block, lds, stream, ...)                                                                     \
    do {                                                                                                                      \
        if (::bert_hip::tl_launch_timing) {                                                                                   \
            ++::bert_hip::tl_launch_timing->launches;                                                                         \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ::bert_hip::tl_launch_timing->start,                      \
                                  ::bert_hip::tl_launch_timing->stop, 0, __VA_ARGS__);                                        \
        } else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                             \
    } while (0)

// The f32 route (f32_route.hip): the forward pass of f32 model files in f32 arithmetic — f32 activations, mat-muls on
// v_mfma_f32_32x32x2_f32 (any M, N, K; C = epi(A W^T + bias (+ resid))), f32 softmax / GELU / LayerNorm / pooling.
void launch_f32_embed_ln(const float *word, const float *type, const float *pos, const float *gamma, const float *beta, const int32_t *tokens,
                         const int32_t *cu_seqlens, int n_sentences, int T, int H, int n_vocab, float *out, hipStream_t stream);
void launch_f32_gemm(const float *A, const float *W, const float *bias, const float *resid, float *C, int M, int N, int K, int epilogue,
                     hipStream_t stream);
void launch_f32_attention(const float *qkv, const int32_t *cu_seqlens, int n_sentences, int n_head, int d_head, int max_len, float *out,
                          hipStream_t stream);
void launch_f32_layernorm(float *x, const float *gamma, const float *beta, int T, int H, hipStream_t stream);
void launch_f32_pool_normalize(const float *x, const int32_t *cu_seqlens, int n_sentences, int H, int max_len, int *status, float *out,
                               hipStream_t stream);

// bytes (rounded up to 16) from mapped pinned host memory to device memory by a kernel (small staged blocks of the host API)
void launch_stage_copy(const void *mapped_src, void *dst, size_t bytes, hipStream_t stream);

// f16 [rows][cols] -> f32 (hidden-state tap)
void launch_f16_to_f32(const half_t *src, float *dst, size_t n, hipStream_t stream);

}  // namespace bert_hip
