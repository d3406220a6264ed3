// skinny.hip — the latency route: one encoder layer for a batch of at most 128 tokens (a single sentence, the regime of
// the reference's real callers: bert_encode per request, reference bert.cpp:943-950, examples/server.cpp:98-114).
//
// The fused kernels give a workgroup 128 tokens x ALL features: one sentence keeps ONE of 256 CUs busy for the whole layer
// (qkv_attention2 31 us + layer_tail 71 us per layer, 0.63 ms per sentence).  Here every weight mat-mul of the layer is
// split by OUTPUT FEATURES over many workgroups — 32 features per workgroup, a wave per block of 32 tokens — so the
// 2.6 MB of a layer's weights stream through 12 .. 48 CUs at once:
//     QKV projection (36 workgroups at H = 384) -> attention (attention.hip, a workgroup per head) -> out-projection (12)
//     -> LayerNorm 1 -> up-projection + GELU (48) -> down-projection (12) -> LayerNorm 2
// Operands go HBM / L2 -> registers directly (a wave needs one 16-byte weight fragment and one token fragment per MFMA:
// nothing is shared between waves that LDS could serve twice), eight k-steps in flight.
//
// SAME BITS AS THE BATCH ROUTE.  A sentence's embedding must not depend on what it is batched with (SURVEY.md section 8b:
// the reference evaluates sentences one by one), so every kernel here performs, per output element, exactly the arithmetic
// of its fused counterpart: the same v_mfma_f32_32x32x16_f16 sequence over k ascending with the same fragment contents (the
// k order inside a group of 16 included: GemmWeight::w16p for the feed-forward), the same initial accumulator values
// (x + bo; b1; y + b2 resp. b2 with y added at the end for the features layer_tail's U wave owns), the same packed-f16
// GELU on the same element pairs, and LayerNorm statistics summed in the order layer_tail's lanes and wave pairs sum them.
// tests/test_gpu_parity.py::test_latency_route_gives_the_batch_route_s_bits holds the two routes against each other.
#include "tile_stream.h"

namespace bert_hip {

namespace {

enum SkinnyMode : int { SK_QKV = 0, SK_PROJ = 1, SK_UP = 2, SK_DOWN = 3 };

// (the weight matrix W [N_pad][K] f16 — QKV, PROJ: GemmWeight::w16; UP, DOWN: w16p — K, N and the workgroup's wave count are
// kernel arguments of their own, in front: the first sixteen dwords of the arguments are preloaded into SGPRs at dispatch
// (-mllvm -amdgpu-kernarg-preload-count), so the weight block is requested without waiting for a load of the arguments)
struct SkinnyArgs {
    const half_t *A;         // [T_pad][K] f16 activations: QKV without LayerNorm: x; PROJ: ctx; DOWN: the GELU'ed intermediate,
                             // stored in fragment order (see the UP epilogue)
    const float *V;          // LayerNorm-fused forms (UP always, QKV from the second layer on): pre-LayerNorm values [T_pad][K] f32
    const float *gamma, *beta;
    half_t *ln_out;          // the LayerNorm'ed rows [T_pad][K] f16 (written by the workgroups of feature tile 0: the residual later)
    const float *bias;       // [N]
    const half_t *resid;     // PROJ: x [T_pad][N]; DOWN: y [T_pad][N]
    half_t *out16;           // QKV: [T_pad][N]; UP: [T_pad][N] in fragment order
    float *out32;            // PROJ, DOWN: [T_pad][N] pre-LayerNorm values
};

}  // namespace

// grid = (N / 32 feature tiles, token blocks), block = 128 or 256: ONE wave of the workgroup owns 32 tokens x 32 features (up to
// 192 workgroups at once for a 128-token sentence), the others only request their share of the LDS-DMA pieces.  The tile's weight rows (32 x K halfs = 24 .. 96 KiB, one contiguous block)
// come in by LDS-DMA in one round trip — fully coalesced 1 KiB pieces, the 16-byte chunk of a row XOR-swizzled on the
// source side so that the fragment reads are conflict-free.  The token operand:
//   LN == 0: 16-byte fragments straight from HBM / L2 (x in the first layer, ctx, the intermediate in fragment order);
//   LN != 0: the workgroup LayerNorms its 32 rows itself (redundantly per feature tile: 1 us of arithmetic; sharing the
//   rows among four tiles of a workgroup was measured: slower) from the f32 values the kernel before left, which arrive by
//   LDS-DMA behind the weight block — for the up-projection the normalised runs in their registers ARE its fragments
//   (layer_tail.hip's trick: w16p); for the QKV projection of the next layer (plain k order) the two lane halves swap one
//   run per k-step.  The workgroups of feature tile 0 write the normalised rows for their later use as residual.
template <int MODE, int LN, int NT>
__global__ __launch_bounds__(512) void skinny_gemm_kernel(const half_t *__restrict__ W, int K, int N, int n_waves, SkinnyArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // wave 0 owns the tile; waves 1 .. only help to request the weight block (one wave issues a 96 KiB block in 2.8 us,
    // four in 0.7) and leave at the barrier.  (The down-projection's k range handed from wave to wave through LDS, every
    // wave with its quarter of the token fragments requested at once, keeps the bits and costs 1.2 us per launch more.)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int n0 = blockIdx.x * 32, tok = blockIdx.y * 32 + l31;
    const int cpr = K >> 3;                                   // 16-byte chunks per weight row (a multiple of 16)

    // ---- weights -> LDS: 16-byte unit u = row * cpr + c holds chunk (c & ~15) | ((c ^ row) & 15) of the row
    {
        const char *wbase = (const char *)(W + (size_t)n0 * K);
        const int n_pieces = K >> 4;                          // 32 rows * K * 2 B / 1 KiB
        for (int pc = wave; pc < n_pieces; pc += n_waves) {   // (every wave of the workgroup requests its share)
            const int u = pc * 64 + lane, row = u / cpr, c = u - row * cpr;
            const int src = (c & ~15) | ((c ^ row) & 15);
            __builtin_amdgcn_global_load_lds(AS_GLOBAL(wbase + (size_t)row * K * 2 + src * 16), AS_LDS(smem + pc * 1024), 16, 0, 0);
        }
    }
    // ---- LN != 0: the 32 pre-LayerNorm rows -> LDS behind the weights, run (n, g) of all lanes = one 1 KiB piece (lane
    // (l31, hi): features 32 n + 8 g + 4 hi .. + 3 of its token).  In registers the row's 64 x H floats leave the compiler
    // no room to keep the parameter reads in flight (it waited for every pair in turn: 4 us per launch).
    char *const xl = smem + (size_t)32 * K * 2;
    if constexpr (LN != 0) {
        const char *vrow = (const char *)(p.V + (size_t)tok * K + 4 * hi);
        for (int pc = wave; pc < 16 * NT; pc += n_waves)
            __builtin_amdgcn_global_load_lds(AS_GLOBAL(vrow + (pc >> 2) * 128 + (pc & 3) * 32), AS_LDS(xl + pc * 1024), 16, 0, 0);
    }

    if (wave != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        return;
    }
    // ---- requested now, used behind the wait for the weight block (a load behind the MFMAs, or a wait for these values in
    // front of the fragment requests, is a round trip of its own): bias and residual of the tile
    f32x4 bias4[4];
    [[maybe_unused]] f16x4 resid4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int f = n0 + 8 * g + 4 * hi;
        bias4[g] = *(const f32x4 *)(p.bias + f);
        if constexpr (MODE == SK_PROJ || MODE == SK_DOWN) resid4[g] = *(const f16x4 *)(p.resid + (size_t)tok * N + f);
    }
    // the accumulators' initial value: register r = feature n0 + 8 (r >> 2) + 4 hi + (r & 3) of token `tok`
    f32x16 acc;
    auto form_acc = [&]() __attribute__((always_inline)) {
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
    };

    const char *wl = smem + (size_t)l31 * cpr * 16;           // this lane's weight row in LDS
    auto weight_frag = [&](int q) __attribute__((always_inline)) {
        const int c = 2 * q + hi;
        return *(const f16x8 *)(wl + (((c & ~15) | ((c ^ l31) & 15)) << 4));
    };

    if constexpr (LN != 0) {
        // ---- LayerNorm in registers, then k ascending over the whole row (K = H = 128 NT)
        f16x4 y[4 * NT][4];
        layernorm_runs_of<LN == 1, NT>(
            [&](int n, int g) __attribute__((always_inline)) { return *(const f32x4 *)(xl + (n * 4 + g) * 1024 + lane * 16); },
            [&]() __attribute__((always_inline)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the weight block and the rows have landed
                __builtin_amdgcn_s_barrier();
            },
            p.gamma, p.beta, hi, y);
        form_acc();
        if (blockIdx.x == 0) {
            half_t *orow = p.ln_out + (size_t)tok * K;
#pragma unroll
            for (int n = 0; n < 4 * NT; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) *(f16x4 *)(orow + 32 * n + 8 * g + 4 * hi) = y[n][g];
        }
#pragma unroll
        for (int q = 0; q < 8 * NT; ++q) {
            constexpr int dummy = 0; (void)dummy;
            const int n = q >> 1, s = q & 1;
            f16x8 b;
            if constexpr (MODE == SK_UP) {                    // fragment order = the runs' own order
#pragma unroll
                for (int e = 0; e < 4; ++e) { b[e] = y[n][2 * s][e]; b[4 + e] = y[n][2 * s + 1][e]; }
            } else {                                          // plain k order: k = 16 q + 8 hi .. + 7 = run 2 s + hi of BOTH lane halves
                const f16x4 send = hi ? y[n][2 * s] : y[n][2 * s + 1];
                const unsigned s0 = __builtin_bit_cast(unsigned, f16x2_t{send[0], send[1]}), s1 = __builtin_bit_cast(unsigned, f16x2_t{send[2], send[3]});
                const f16x2_t r0 = __builtin_bit_cast(f16x2_t, (unsigned)__shfl_xor((int)s0, 32)), r1 = __builtin_bit_cast(f16x2_t, (unsigned)__shfl_xor((int)s1, 32));
                const f16x4 recv = {r0[0], r0[1], r1[0], r1[1]};
                const f16x4 lo = hi ? recv : y[n][2 * s], up = hi ? y[n][2 * s + 1] : recv;
#pragma unroll
                for (int e = 0; e < 4; ++e) { b[e] = lo[e]; b[4 + e] = up[e]; }
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(weight_frag(q), b, acc, 0, 0, 0);
        }
    } else {
        // ---- token fragments from memory: batches of 8 k-steps, four batches in flight
        const half_t *arow = p.A + (size_t)tok * K + 8 * hi;
        const int nb = K >> 7;
        f16x8 b[4][8];
        auto load_b = [&](auto slot_tag, int batch) __attribute__((always_inline)) {
            constexpr int sl = decltype(slot_tag)::value;
#pragma unroll
            for (int u = 0; u < 8; ++u) b[sl][u] = *(const f16x8 *)(arow + 16 * (batch * 8 + u));
        };
        static_for<4>([&](auto j_tag) __attribute__((always_inline)) { if (decltype(j_tag)::value < nb) load_b(j_tag, decltype(j_tag)::value); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the weight block has landed (and the first fragments)
        __builtin_amdgcn_s_barrier();
        form_acc();
        for (int i0 = 0; i0 < nb; i0 += 4) {
            static_for<4>([&](auto j_tag) __attribute__((always_inline)) {
                constexpr int j = decltype(j_tag)::value;
                const int batch = i0 + j;
                if (batch < nb) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(weight_frag(batch * 8 + u), b[j][u], acc, 0, 0, 0);
                    if (batch + 4 < nb) load_b(j_tag, batch + 4);
                }
            });
        }
    }

    // ---- epilogue
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int f = n0 + 8 * g + 4 * hi;
        if constexpr (MODE == SK_QKV) {                       // acc + bias, one rounding (gemm.hip / qkv_attention2.hip)
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (_Float16)(acc[4 * g + e] + bias4[g][e]);
            *(f16x4 *)(p.out16 + (size_t)tok * N + f) = o;
        } else if constexpr (MODE == SK_UP) {                 // packed-f16 GELU of adjacent pairs (layer_tail.hip: gelu_pair)
            const f16x2_t g0 = gelu_pk16(acc[4 * g], acc[4 * g + 1]), g1 = gelu_pk16(acc[4 * g + 2], acc[4 * g + 3]);
            const f16x4 o = {g0[0], g0[1], g1[0], g1[1]};
            // stored in FRAGMENT order: inside every group of 16 features the runs sit at [0-3, 8-11, 4-7, 12-15] (w16p's order),
            // so that the down-projection's token fragment is one 16-byte load: run 8 (g & 1) + 4 hi of group g >> 1 goes
            // to position 8 hi + 4 (g & 1)
            *(f16x4 *)(p.out16 + (size_t)tok * N + n0 + 16 * (g >> 1) + 8 * hi + 4 * (g & 1)) = o;
        } else {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[4 * g + e];
            if constexpr (MODE == SK_DOWN) {                  // U's features: the residual comes last (layer_tail.hip, LayerNorm 2)
                if ((f & 127) < 64) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)resid4[g][e];
                }
            }
            *(f32x4 *)(p.out32 + (size_t)tok * N + f) = v;
        }
    }
}

// the last layer's LayerNorm 2 (every other LayerNorm of the route is fused into the projection behind it)
template <int NT>
__global__ __launch_bounds__(64) void skinny_layernorm_kernel(const float *__restrict__ v, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, half_t *__restrict__ out) {
    constexpr int H = 128 * NT;
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5, tok = blockIdx.x * 32 + l31;
    f16x4 y[4 * NT][4];
    layernorm_runs<false, NT>(v + (size_t)tok * H, gamma, beta, hi, y);
    half_t *orow = out + (size_t)tok * H;
#pragma unroll
    for (int n = 0; n < 4 * NT; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) *(f16x4 *)(orow + 32 * n + 8 * g + 4 * hi) = y[n][g];
}

bool skinny_layer_supported(const GemmWeight &Wqkv, const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2) {
    // exactly the models the fused batch route serves (the bit-for-bit mirror is of THOSE kernels)
    const int H = W1.K, I = W1.N;
    return layer_tail_supported(Wo, W1, W2) && Wqkv.type == GW_F16 && Wqkv.w16 && Wqkv.K == H && Wqkv.N == 3 * H && (H == 256 || H == 384) &&
           I % 128 == 0 && (size_t)32 * I * 2 <= 160 * 1024;
}

void launch_skinny_gemm(int mode, const GemmWeight &W, const half_t *A, const float *V, const float *gamma, const float *beta,
                        half_t *ln_out, const float *bias, const half_t *resid, half_t *out16, float *out32, int n_token_blocks,
                        hipStream_t stream) {
    SkinnyArgs a;
    const half_t *w = (mode == SK_UP || mode == SK_DOWN) ? W.w16p : W.w16;
    a.A = A; a.V = V; a.gamma = gamma; a.beta = beta; a.ln_out = ln_out;
    a.bias = bias; a.resid = resid; a.out16 = out16; a.out32 = out32;
    // eight waves per workgroup: wave 0 computes, ALL of them request their share of the LDS-DMA pieces (24 .. 96 of 1 KiB) —
    // two, four, eight request waves: 232, 229, 219 us per 128-token call (profiles/r4_experiments.txt): more requests in flight
    const dim3 grid(W.N / 32, n_token_blocks), block(512);     
    const size_t lds = (size_t)32 * W.K * 2 + (V ? (size_t)32 * W.K * 4 : 0);     // the tile's weight rows (+ the pre-LayerNorm rows)
    static DeviceFlags configured[8];
    auto go = [&](auto kernel, int m) {
        if (lds > 64 * 1024) configure_once(configured[m], [&] { (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        BERT_LAUNCH(kernel, grid, block, lds, stream, w, W.K, W.N, (int)(block.x / 64), a);
    };
    const bool nt2 = W.K == 256;                             // (only the LayerNorm-fused forms depend on NT: K = H there)
    switch (mode) {
        case SK_QKV:
            if (!V) go(skinny_gemm_kernel<SK_QKV, 0, 3>, 0);
            else if (nt2) go(skinny_gemm_kernel<SK_QKV, 2, 2>, 1);
            else go(skinny_gemm_kernel<SK_QKV, 2, 3>, 2);
            break;
        case SK_PROJ: go(skinny_gemm_kernel<SK_PROJ, 0, 3>, 3); break;
        case SK_UP:
            if (nt2) go(skinny_gemm_kernel<SK_UP, 1, 2>, 4); else go(skinny_gemm_kernel<SK_UP, 1, 3>, 5);
            break;
        default: go(skinny_gemm_kernel<SK_DOWN, 0, 3>, 6); break;
    }
}

void launch_skinny_layernorm(const float *v, const float *gamma, const float *beta, half_t *out, int n_token_blocks, int H,
                             hipStream_t stream) {
    const dim3 grid(n_token_blocks), block(64);
    if (H == 256) BERT_LAUNCH(skinny_layernorm_kernel<2>, grid, block, 0, stream, v, gamma, beta, out);
    else BERT_LAUNCH(skinny_layernorm_kernel<3>, grid, block, 0, stream, v, gamma, beta, out);
}

}  // namespace bert_hip
