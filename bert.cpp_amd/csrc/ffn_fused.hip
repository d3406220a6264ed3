// ffn_fused.hip — the whole feed-forward block of an encoder layer in ONE kernel (gfx950):
//     x_out = LayerNorm( gelu(y W1^T + b1) W2^T + b2 + y ) * gamma + beta
//
// Replaces ggml_mul_mat(ff_i_w) + bias + ggml_gelu + ggml_mul_mat(ff_o_w) + bias + residual add +
// ggml_norm + gamma/beta (reference bert.cpp:878-901) — 63 % of the forward pass's FLOPs.  The
// [tokens, n_intermediate] activation never exists in HBM: it is produced and consumed in LDS in
// 128-feature chunks, and the LayerNorm is done on the accumulators.
//
// One workgroup = 128 tokens, 8 waves (2 token halves x 4 feature quarters of a 128-wide tile).
// For every 128-wide chunk c of the intermediate dimension:
//   U phase  accU[128 tok x 128]  = y_tile * W1[c]^T         (K = H, streamed in 64-deep tiles)
//            hc = f16(gelu(accU + b1[c]))  -> LDS (swizzled [128][128])
//   D phase  acc2[128 tok x H]   += hc * W2[:, c]^T          (K = 128, H/128 feature thirds)
// All operands of both phases arrive as one uniform stream of 32-KiB ring slots (U: y k-tile + W1
// k-tile, D: W2 tile) written by global_load_lds_dwordx4 two tiles ahead of their use and retired
// with a counted s_waitcnt vmcnt(N) + one s_barrier per tile, so the matrix pipe never drains
// between the 12 x (H/64 + 2 H/128) tiles of a workgroup.  MFMA operand roles are as in gemm.hip
// (weights = A operand, activations = B operand: a lane's accumulator column is one token), which
// makes the GELU epilogue, the residual add and the LayerNorm statistics per-lane-column work.
#include "tile_stream.h"

#include <cstdlib>
#include <type_traits>

namespace bert_hip {

struct FfnArgs {
    const half_t *y;       // [T_pad][H]   LayerNorm'ed attention output (input and residual)
    const half_t *w1;      // [I_pad][H]   f16            (WT == GW_F16)
    const half_t *w2;      // [H_pad][I]   f16
    const uint4 *q1, *q2;  // q4 nibble planes of W1 / W2 (WT != GW_F16), tile-contiguous (kernels.h)
    const void *s1, *s2;   // q4 scale planes
    const float *b1, *b2, *gamma, *beta;
    half_t *out;           // [T_pad][H]
    // optional leading phase (PROJ): y = LayerNorm(ctx Wo^T + bo + x) * g1 + beta1 is computed by this
    // kernel first (written to `ybuf`, which then plays the role of `y`)
    const half_t *ctx, *x;   // [T_pad][H] attention context, layer input (residual)
    const half_t *wo;        // [H_pad][H] f16
    const uint4 *qo;         // q4 planes of Wo
    const void *so;
    const float *bo, *g1, *beta1;
    half_t *ybuf;            // [T_pad][H]
    int I;
};

template <int NT, int WT, bool PROJ>
__global__ __launch_bounds__(512, 2) void ffn_fused_kernel(FfnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int H = 128 * NT, KU = H / 64, TPC = KU + 2 * NT;
    const int I = a.I, NC = I / 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // provably wave-uniform -> SGPR
    const int wt = wave >> 2, wq = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.x * 128;

    char *ring = smem;
    char *hc = smem + FF_HC;
    float *cb1 = (float *)(smem + FF_CONST);
    float *cb2 = cb1 + I, *cg = cb2 + H, *cbeta = cg + H;
    float *red = cbeta + H;                                   // [4 quarters][128 tokens]

    const half_t *ybase = (PROJ ? a.ybuf : a.y) + (size_t)m0 * H;

    // lane's byte offsets inside a [128 rows x 64 halfs] source tile with row stride H (y, W1) or I (W2);
    // the 16-byte chunk is pre-swizzled here and un-swizzled by the fragment reads (off64)
    unsigned loffH[2], loffI[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((r >> 1) & 7);
        loffH[i] = (unsigned)(r * H + ch * 8) * 2u;
    }
    // tile (chunk c, position p) -> ring slot `slot`; U tiles carry a y k-tile and a W1 k-tile, D tiles a W2
    // tile.  f16 weights travel by LDS-DMA like the activations; q4 weights are fetched into `pend` (two
    // ordinary loads per thread) and expanded into their slot by commit() one interval later.
    QRegs pend = {{0, 0, 0, 0}, 0};
    auto issue = [&](int c, int p, int slot) {
        char *dst = ring + slot * FF_SLOT;
        if (p < KU) {
            // keep the (loop-invariant) y addresses out of long-lived VGPR pairs: recompute per use
            const half_t *yb = ybase;
            asm volatile("" : "+s"(yb));
            dma_tile8(yb + p * 64, loffH, dst, wave);
            if (WT == GW_F16) dma_tile8(a.w1 + (size_t)c * 128 * H + p * 64, loffH, dst + 16384, wave);
            else pend = q4_fetch<WT>(a.q1, a.s1, (size_t)c * KU + p, tid);
        } else {
            const int k2 = (p - KU) / NT, n3 = (p - KU) - k2 * NT;      // k-half major: all of k2 = 0 first
            if (WT == GW_F16) dma_tile8(a.w2 + (size_t)n3 * 128 * I + c * 128 + k2 * 64, loffI, dst, wave);
            else pend = q4_fetch<WT>(a.q2, a.s2, (size_t)n3 * (I / 64) + 2 * c + k2, tid);
        }
    };
    // expand the pending q4 tile (position p) into its slot
    auto commit = [&](int p, int slot) {
        if (WT != GW_F16) q4_expand_to_lds<WT>(pend, ring + slot * FF_SLOT + (p < KU ? 16384 : 0), tid);
    };

    // ---- prologue: constants into LDS (the first FFN tiles are requested further down)
    for (int i = tid; i < I; i += 512) cb1[i] = a.b1[i];
    for (int i = tid; i < H; i += 512) { cb2[i] = a.b2[i]; cg[i] = a.gamma[i]; cbeta[i] = a.beta[i]; }

    // per-lane LDS byte offsets of the MFMA fragments (swizzles are XORs, so one VGPR per k-step)
    // U phase: the wave's 32 MFMA rows are 16 features of each 64-feature half of the chunk
    // (tile rows wq*16.. and 64 + wq*16..), so accumulator registers 0-7 belong to k-half 0 and 8-15 to
    // k-half 1 of the D phase: the GELU of half 1 can run under the MFMAs of the first D tiles.
    int aW[4], aWU[4], aY[4], aH[2][4];                // (aWU, aH, loffI are filled in after the PROJ phase)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        aW[kk] = off64(wq * 32 + l31, kk * 2 + hi);
        aY[kk] = off64(wt * 64 + l31, kk * 2 + hi);
    }

    f32x16 acc2[NT][2], accU[2];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[n][j][r] = 0.f;

    [[maybe_unused]] int tl = 1;
    TL_STAMP(0);
    if (PROJ) {
        // ---- leading phase: attention output projection + residual + LayerNorm for the same 128 tokens
        // (reference bert.cpp:859-875), on the same ring: tiles (n3, k) = ctx k-tile + Wo tile.  Its result
        // goes to global memory (the FFN streams it back k-tile by k-tile and needs it as residual).
        float *cbo = red + 512, *cg1 = cbo + H, *cbeta1 = cg1 + H;
        for (int i = tid; i < H; i += 512) { cbo[i] = a.bo[i]; cg1[i] = a.g1[i]; cbeta1[i] = a.beta1[i]; }
        const half_t *cbase = a.ctx + (size_t)m0 * H;
        auto issue_p = [&](int t, int slot) {
            const int n3 = t / KU, k = t - n3 * KU;
            char *dst = ring + slot * FF_SLOT;
            dma_tile8(cbase + k * 64, loffH, dst, wave);
            if (WT == GW_F16) dma_tile8(a.wo + (size_t)n3 * 128 * H + k * 64, loffH, dst + 16384, wave);
            else pend = q4_fetch<WT>(a.qo, a.so, (size_t)n3 * KU + k, tid);
        };
        constexpr int PT = NT * KU;
        issue_p(0, 0);
        commit(0, 0);
        issue_p(1, 1);
        int slot = 0;
#pragma unroll
        for (int n3 = 0; n3 < NT; ++n3) {
            for (int k = 0; k < KU; ++k) {
                const int t = n3 * KU + k;
                if (t + 1 < PT) wait_vm_barrier<4>(); else wait_vm_barrier<0>();
                TL_STAMP(tl++);
                if (t + 1 < PT) commit(0, slot == 2 ? 0 : slot + 1);
                const char *sl = ring + slot * FF_SLOT;
                f16x8 wf[4], a0[4], a1[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    wf[kk] = *(const f16x8 *)(sl + 16384 + aW[kk]);
                    a0[kk] = *(const f16x8 *)(sl + aY[kk]);
                    a1[kk] = *(const f16x8 *)(sl + aY[kk] + 32 * 128);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (t + 2 < PT) issue_p(t + 2, slot == 0 ? 2 : slot - 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    acc2[n3][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk], a0[kk], acc2[n3][0], 0, 0, 0);
                    acc2[n3][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk], a1[kk], acc2[n3][1], 0, 0, 0);
                }
                slot = slot == 2 ? 0 : slot + 1;
            }
        }
        ln_epilogue<NT>(acc2, cbo, cg1, cbeta1, red, a.x + (size_t)m0 * H, a.ybuf + (size_t)m0 * H, ring, tid, wt, wq, l31, hi);
        // the y rows are re-read below by LDS-DMA: stores complete (write-through to L2), then everybody
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[n][j][r] = 0.f;
    }
    // per-lane values of the FFN phases only (computed here so they are not live across the PROJ phase)
    {
        const int rowU = (l31 < 16 ? 0 : 48) + wq * 16 + l31;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            aWU[kk] = off64(rowU, kk * 2 + hi);
            aH[0][kk] = off_hc(wt * 64 + l31, kk * 2 + hi);
            aH[1][kk] = off_hc(wt * 64 + l31, 8 + kk * 2 + hi);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (wave * 2 + i) * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ ((r >> 1) & 7);
            loffI[i] = (unsigned)(r * I + ch * 8) * 2u;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accU[j][r] = 0.f;
    }
    TL_STAMP(tl++);
    // first two FFN tiles in flight
    issue(0, 0, 0);
    commit(0, 0);
    issue(0, 1, 1);                                  // its q4 part is committed in interval 0

    // One chunk = TPC tiles; everything about a tile except the chunk index is a compile-time
    // constant of its position p, so the steady state is branch-free.  LAST = final chunk (its last
    // two tiles prefetch nothing and its last wait drains the DMA queue).
    // bias + GELU + f16 of one k-half of the chunk's activation (accumulator registers 8*half .. 8*half+7 of
    // both token fragments) into hc[token][feature]; the registers are re-zeroed for the next chunk
    auto gelu_half = [&](int c, int half) {
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int g = half * 2 + gg;
            const int fl = half * 64 + wq * 16 + gg * 8 + 4 * hi;          // feature within the chunk
            const f32x4 bv = *(const f32x4 *)(cb1 + c * 128 + fl);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int tok = wt * 64 + j * 32 + l31;
                f16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (_Float16)gelu_fast(accU[j][4 * g + e] + bv[e]);
                    accU[j][4 * g + e] = 0.f;
                }
                *(f16x4 *)(hc + off_hc(tok, fl >> 3) + (fl & 4) * 2) = o;
            }
        }
    };
    int sbase = 0;                                   // ring slot of the chunk's first tile
    auto chunk = [&](auto last_tag, int c) {
        constexpr bool LAST = decltype(last_tag)::value;
#pragma unroll
        for (int p = 0; p < TPC; ++p) {
            int slot = sbase + (p % 3);
            slot = slot >= 3 ? slot - 3 : slot;
            // tile landed for every wave?  (the next tile may stay in flight: 4 pieces if U, 2 if D)
            if (LAST && p == TPC - 1) wait_vm_barrier<0>();
            else if (((p + 1) % TPC) < KU) wait_vm_barrier<4>();
            else wait_vm_barrier<2>();
            if (c < 2 || LAST) TL_STAMP(tl++);
            // q4: the weights of tile +1 were fetched one interval ago; expand them into their (free) slot now
            if (WT != GW_F16 && !(LAST && p + 1 >= TPC)) {
                int s1 = slot + 1;
                s1 = s1 >= 3 ? s1 - 3 : s1;
                commit((p + 1) % TPC, s1);
            }
            const int so = slot * FF_SLOT;
            f16x8 wf[4], a0[4], a1[4];
            if (p < KU) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    wf[kk] = *(const f16x8 *)(ring + so + 16384 + aWU[kk]);
                    a0[kk] = *(const f16x8 *)(ring + so + aY[kk]);
                    a1[kk] = *(const f16x8 *)(ring + so + aY[kk] + 32 * 128);
                }
            } else {
                const int k2 = (p - KU) / NT;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    wf[kk] = *(const f16x8 *)(ring + so + aW[kk]);
                    a0[kk] = *(const f16x8 *)(hc + aH[k2][kk]);
                    a1[kk] = *(const f16x8 *)(hc + aH[k2][kk] + 32 * 256);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // prefetch tile +2 into the slot that was read one interval ago (free since the barrier above)
            if (!(LAST && p + 2 >= TPC)) {
                int s2 = slot + 2;
                s2 = s2 >= 3 ? s2 - 3 : s2;
                if (p + 2 < TPC) issue(c, p + 2, s2); else issue(c + 1, p + 2 - TPC, s2);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (p < KU) {
                // ---- U: accU += W1 tile (features) x y tile (tokens); fragments were all fetched above so
                // the 8 MFMAs issue back to back
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    accU[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk], a0[kk], accU[0], 0, 0, 0);
                    accU[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk], a1[kk], accU[1], 0, 0, 0);
                }
                if (p == KU - 1) gelu_half(c, 0);             // k-half 0 of hc is needed by the next tile
            } else {
                // ---- D: acc2[n3] += W2 tile (features) x hc (tokens), k-half k2 of the chunk
                const int n3 = (p - KU) % NT;
                // k-half 1 of hc is first read NT tiles from now: its GELU overlaps this tile's MFMAs
                if (p == KU) gelu_half(c, 1);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    acc2[n3][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk], a0[kk], acc2[n3][0], 0, 0, 0);
                    acc2[n3][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk], a1[kk], acc2[n3][1], 0, 0, 0);
                }
                if (p == KU) {
                    // pin the interleave: one MFMA, then a slice of the GELU's VALU work in its shadow
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);   // VALU (incl. transcendental)
                    }
                }
            }
        }
        sbase += TPC % 3;
        sbase = sbase >= 3 ? sbase - 3 : sbase;
    };
    for (int c = 0; c < NC - 1; ++c) chunk(std::false_type{}, c);
    chunk(std::true_type{}, NC - 1);

    // ---- final epilogue: + b2 + residual (the block input y), LayerNorm, gamma/beta, full-row stores
    // (opaque copies of the lane ids: keeps the compiler from carrying the PROJ phase's epilogue addresses
    // through the whole FFN loop just to reuse them here)
    int tid_e = tid, l31_e = l31, hi_e = hi;
    asm volatile("" : "+v"(tid_e), "+v"(l31_e), "+v"(hi_e));
    TL_STAMP(tl++);
    ln_epilogue<NT>(acc2, cb2, cg, cbeta, red, ybase, a.out + (size_t)m0 * H, ring, tid_e, wt, wq, l31_e, hi_e);
#ifdef BERT_HIP_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TL_STAMP(tl++);
#endif
}

bool ffn_fused_supported(const GemmWeight &W1, const GemmWeight &W2) {
    const int H = W1.K, I = W1.N;
    return W1.type == W2.type && W2.N == H && W2.K == I && H % 128 == 0 && H <= 384 && I % 128 == 0 && I <= 6144;
}

static void launch_ffn_impl(FfnArgs &a, const GemmWeight &W1, bool proj, int M_pad, hipStream_t stream) {
    a.I = W1.N;
    const int H = W1.K;
    const size_t lds = FF_CONST + (size_t)(a.I + 3 * H + 512 + (proj ? 3 * H : 0)) * sizeof(float);
    const int grid = M_pad / 128;
    static DeviceFlags configured[2][3][4];
    auto go = [&](auto kernel, int nt) {
        configure_once(configured[proj][W1.type][nt], [&] { (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), lds, stream, a);
        TL_DUMP(grid >= 256, (proj ? (H / 128) * (H / 64) + 1 : 0) + 3 * (H / 64 + 2 * (H / 128)) + 4);
    };
#define FFN_NT(WTV, PJ)                                                                \
    switch (H / 128) {                                                                  \
        case 1: go(ffn_fused_kernel<1, WTV, PJ>, 1); break;                             \
        case 2: go(ffn_fused_kernel<2, WTV, PJ>, 2); break;                             \
        default: go(ffn_fused_kernel<3, WTV, PJ>, 3); break;                            \
    }
    if (proj) {
        if (W1.type == GW_F16) { FFN_NT(GW_F16, true) } else if (W1.type == GW_Q4_0) { FFN_NT(GW_Q4_0, true) } else { FFN_NT(GW_Q4_1, true) }
    } else {
        if (W1.type == GW_F16) { FFN_NT(GW_F16, false) } else if (W1.type == GW_Q4_0) { FFN_NT(GW_Q4_0, false) } else { FFN_NT(GW_Q4_1, false) }
    }
#undef FFN_NT
}

void launch_ffn_fused(const GemmWeight &W1, const GemmWeight &W2, const half_t *y, const float *b1, const float *b2,
                      const float *gamma, const float *beta, half_t *out, int M_pad, hipStream_t stream) {
    FfnArgs a = {};
    a.y = y; a.w1 = W1.w16; a.w2 = W2.w16; a.q1 = W1.qs; a.q2 = W2.qs; a.s1 = W1.sc; a.s2 = W2.sc; a.b1 = b1; a.b2 = b2; a.gamma = gamma; a.beta = beta; a.out = out;
    launch_ffn_impl(a, W1, false, M_pad, stream);
}

// out-projection + residual + LayerNorm, then the whole FFN block, in ONE launch (same 128-token panels):
//   y   = LayerNorm(ctx Wo^T + bo + x) * g1 + beta1        -> ybuf
//   out = LayerNorm(gelu(y W1^T + b1) W2^T + b2 + y) * g2 + beta2
// out may alias x (every workgroup reads and later writes only its own 128 rows); it must not alias ctx / ybuf.
bool proj_ffn_fused_supported(const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2) {
    return ffn_fused_supported(W1, W2) && Wo.type == W1.type && Wo.N == W1.K && Wo.K == W1.K;
}

void launch_proj_ffn_fused(const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2, const half_t *ctx,
                           const half_t *x, const float *bo, const float *g1, const float *beta1, half_t *ybuf,
                           const float *b1, const float *b2, const float *g2, const float *beta2, half_t *out, int M_pad,
                           hipStream_t stream) {
    FfnArgs a = {};
    a.y = ybuf; a.w1 = W1.w16; a.w2 = W2.w16; a.q1 = W1.qs; a.q2 = W2.qs; a.s1 = W1.sc; a.s2 = W2.sc; a.b1 = b1; a.b2 = b2; a.gamma = g2; a.beta = beta2; a.out = out;
    a.ctx = ctx; a.x = x; a.wo = Wo.w16; a.qo = Wo.qs; a.so = Wo.sc; a.bo = bo; a.g1 = g1; a.beta1 = beta1; a.ybuf = ybuf;
    launch_ffn_impl(a, W1, true, M_pad, stream);
}

}  // namespace bert_hip
