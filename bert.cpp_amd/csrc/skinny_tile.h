// skinny_tile.h — one wave's 32 tokens x 32 features of a weight mat-mul whose weight rows sit in LDS: the body shared by the
// latency route's kernels (skinny.hip: a launch per mat-mul, one computing wave per workgroup) and the one-launch form of the
// route (sentence_kernel.hip: a workgroup per feature tile, a wave per token block).  Same source, same MFMA sequence, same
// epilogue arithmetic: the routes give equal bits (tests/test_gpu_parity.py).
#pragma once
#include "tile_stream.h"

namespace bert_hip {

enum SkinnyMode : int { SK_QKV = 0, SK_PROJ = 1, SK_UP = 2, SK_DOWN = 3 };

struct SkinnyArgs {
    const half_t *W;         // [N_pad][K] f16 (QKV, PROJ: GemmWeight::w16; UP, DOWN: w16p)
    const half_t *A;         // [T_pad][K] f16 activations: QKV without LayerNorm: x; PROJ: ctx; DOWN: the GELU'ed intermediate,
                             // stored in fragment order (see the UP epilogue)
    const float *V;          // LayerNorm-fused forms (UP always, QKV from the second layer on): pre-LayerNorm values [T_pad][K] f32
    const float *gamma, *beta;
    half_t *ln_out;          // the LayerNorm'ed rows [T_pad][K] f16 (written by the workgroups of feature tile 0: the residual later)
    const float *bias;       // [N]
    const half_t *resid;     // PROJ: x [T_pad][N]; DOWN: y [T_pad][N]
    half_t *out16;           // QKV: [T_pad][N]; UP: [T_pad][N] in fragment order
    float *out32;            // PROJ, DOWN: [T_pad][N] pre-LayerNorm values
    int N, K;
};

// weights -> LDS: the tile's 32 rows x K halfs, one contiguous block of 1 KiB pieces; 16-byte unit u = row * cpr + c holds
// chunk (c & ~15) | ((c ^ row) & 15) of the row (XOR-swizzled on the source side: conflict-free fragment reads).  Every wave of
// the workgroup requests its share of the pieces.
__device__ __forceinline__ void skinny_request_weights(const half_t *wtile, int K, char *lds, int wave, int n_waves, int lane) {
    const int cpr = K >> 3, n_pieces = K >> 4;                // 16-byte chunks per row; 32 rows * K * 2 B / 1 KiB
    for (int pc = wave; pc < n_pieces; pc += n_waves) {
        const int u = pc * 64 + lane, row = u / cpr, c = u - row * cpr;
        const int src = (c & ~15) | ((c ^ row) & 15);
        __builtin_amdgcn_global_load_lds(AS_GLOBAL((const char *)wtile + (size_t)row * K * 2 + src * 16), AS_LDS(lds + pc * 1024), 16, 0, 0);
    }
}

// k-step q's weight fragment of lane (l31 = feature row, hi = k half)
__device__ __forceinline__ f16x8 skinny_weight_frag(const char *wlds, int K, int l31, int hi, int q) {
    const int c = 2 * q + hi;
    return *(const f16x8 *)(wlds + (size_t)l31 * (K >> 3) * 16 + (((c & ~15) | ((c ^ l31) & 15)) << 4));
}

// bias and residual of the tile: requested early, used behind the wait for the weight block (a load behind the MFMAs, or a
// wait for these values in front of the fragment requests, is a round trip of its own)
template <int MODE>
struct SkinnyEdge {
    f32x4 bias4[4];
    f16x4 resid4[4];
    __device__ __forceinline__ void request(const SkinnyArgs &p, int n0, int tok, int hi) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = n0 + 8 * g + 4 * hi;
            bias4[g] = *(const f32x4 *)(p.bias + f);
            if constexpr (MODE == SK_PROJ || MODE == SK_DOWN) resid4[g] = *(const f16x4 *)(p.resid + (size_t)tok * p.N + f);
        }
    }
    // the accumulators' initial value: register r = feature n0 + 8 (r >> 2) + 4 hi + (r & 3) of token `tok`
    __device__ __forceinline__ void form_acc(f32x16 &acc, int n0, int hi) const {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = n0 + 8 * g + 4 * hi;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if constexpr (MODE == SK_PROJ) {                  // x + bo (layer_tail.hip: accp)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (float)resid4[g][e] + bias4[g][e];
            } else if constexpr (MODE == SK_UP) {             // b1 (layer_tail.hip: accU)
                v = bias4[g];
            } else if constexpr (MODE == SK_DOWN) {           // b2, + y for the features layer_tail's D wave owns (acc2)
                v = bias4[g];
                if ((f & 127) >= 64) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (float)resid4[g][e] + v[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * g + e] = v[e];
        }
    }
    __device__ __forceinline__ void epilogue(const SkinnyArgs &p, const f32x16 &acc, int n0, int tok, int hi) const {
        const int N = p.N;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = n0 + 8 * g + 4 * hi;
            if constexpr (MODE == SK_QKV) {                   // acc + bias, one rounding (gemm.hip / qkv_attention2.hip)
                f16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (_Float16)(acc[4 * g + e] + bias4[g][e]);
                *(f16x4 *)(p.out16 + (size_t)tok * N + f) = o;
            } else if constexpr (MODE == SK_UP) {             // packed-f16 GELU of adjacent pairs (layer_tail.hip: gelu_pair)
                const f16x2_t g0 = gelu_pk16(acc[4 * g], acc[4 * g + 1]), g1 = gelu_pk16(acc[4 * g + 2], acc[4 * g + 3]);
                const f16x4 o = {g0[0], g0[1], g1[0], g1[1]};
                // stored in FRAGMENT order: inside every group of 16 features the runs sit at [0-3, 8-11, 4-7, 12-15] (w16p's
                // order), so that the down-projection's token fragment is one 16-byte load: run 8 (g & 1) + 4 hi of group
                // g >> 1 goes to position 8 hi + 4 (g & 1)
                *(f16x4 *)(p.out16 + (size_t)tok * N + n0 + 16 * (g >> 1) + 8 * hi + 4 * (g & 1)) = o;
            } else {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[4 * g + e];
                if constexpr (MODE == SK_DOWN) {              // U's features: the residual comes last (layer_tail.hip, LayerNorm 2)
                    if ((f & 127) < 64) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)resid4[g][e];
                    }
                }
                *(f32x4 *)(p.out32 + (size_t)tok * N + f) = v;
            }
        }
    }
};

// The whole of a wave's tile with the token operand from memory (16-byte fragments straight from HBM / L2: x in the first
// layer, ctx, the intermediate in fragment order): batches of 8 k-steps, four batches in flight.  landed(): the caller's
// "every wave's pieces of the weight block are in LDS" (s_waitcnt vmcnt(0) + workgroup barrier), called behind the first requests.
template <int MODE, class Landed>
__device__ __forceinline__ void skinny_wave(const SkinnyArgs &p, const char *wlds, int n0, int tok, int lane, Landed landed) {
    const int l31 = lane & 31, hi = lane >> 5, K = p.K;
    SkinnyEdge<MODE> edge;
    edge.request(p, n0, tok, hi);
    const half_t *arow = p.A + (size_t)tok * K + 8 * hi;
    const int nb = K >> 7;
    f16x8 b[4][8];
    auto load_b = [&](auto slot_tag, int batch) __attribute__((always_inline)) {
        constexpr int sl = decltype(slot_tag)::value;
#pragma unroll
        for (int u = 0; u < 8; ++u) b[sl][u] = *(const f16x8 *)(arow + 16 * (batch * 8 + u));
    };
    static_for<4>([&](auto j_tag) __attribute__((always_inline)) { if (decltype(j_tag)::value < nb) load_b(j_tag, decltype(j_tag)::value); });
    landed();
    f32x16 acc;
    edge.form_acc(acc, n0, hi);
    for (int i0 = 0; i0 < nb; i0 += 4) {
        static_for<4>([&](auto j_tag) __attribute__((always_inline)) {
            constexpr int j = decltype(j_tag)::value;
            const int batch = i0 + j;
            if (batch < nb) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(skinny_weight_frag(wlds, K, l31, hi, batch * 8 + u), b[j][u], acc, 0, 0, 0);
                if (batch + 4 < nb) load_b(j_tag, batch + 4);
            }
        });
    }
    edge.epilogue(p, acc, n0, tok, hi);
}

}  // namespace bert_hip
