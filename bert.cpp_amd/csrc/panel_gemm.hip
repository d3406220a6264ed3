// panel_gemm.hip — "row panel" weight mat-muls on the tile-stream machinery of ffn_fused.hip (gfx950).
//
// One workgroup (8 waves) owns 128 tokens and walks ALL feature tiles of the output, so operand tiles
// form one long stream through a 3-slot LDS ring (global_load_lds two tiles ahead, counted vmcnt, one
// barrier per tile) and the matrix pipe does not drain between feature tiles.  Two kernels:
//   proj_ln_kernel<NT>   out = LayerNorm(A W^T + b + resid) * gamma + beta, N = 128*NT <= 384: the
//                        attention output projection + residual + LayerNorm of reference
//                        bert.cpp:859-875 (also usable for :885-901) — the row statistics are
//                        computed on the accumulators, the stand-alone LayerNorm kernel disappears.
//   panel_store_kernel   C = A W^T + b for any N (multiple of 8): the fused Q|K|V projection of
//                        reference bert.cpp:822-839; each finished 128x128 tile is transposed through
//                        LDS and written as full 256-byte rows while the next tiles stream in.
// f16 weights stream by LDS-DMA, q4_0 / q4_1 weights are expanded in registers on their way into the ring;
// shapes outside the limits of panel_gemm_supported() use gemm.hip.
#include "tile_stream.h"

namespace bert_hip {

struct PanelArgs {
    const half_t *A;        // [T_pad][K]
    const half_t *W;        // [N_pad][K] f16            (WT == GW_F16)
    const uint4 *qs;        // q4 nibble plane, tile-contiguous (kernels.h)   (WT != GW_F16)
    const void *sc;         // q4 scale plane
    const float *bias;      // [N]
    const half_t *resid;    // [T_pad][N]       (proj_ln)
    const float *gamma, *beta;
    half_t *out;            // [T_pad][N]
    int N, K;
};

// acc[j] += Wtile(features wq*32..+32) x Atile(tokens wt*64 + j*32..+32) for the tile in `slot`
__device__ __forceinline__ void mma_slot(const char *slot, const int (&aW)[4], const int (&aY)[4], f32x16 (&acc)[2]) {
    f16x8 wf[4], a0[4], a1[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        wf[kk] = *(const f16x8 *)(slot + 16384 + aW[kk]);
        a0[kk] = *(const f16x8 *)(slot + aY[kk]);
        a1[kk] = *(const f16x8 *)(slot + aY[kk] + 32 * 128);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk], a0[kk], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk], a1[kk], acc[1], 0, 0, 0);
    }
}

#define PANEL_COMMON_SETUP                                                                         \
    extern __shared__ __attribute__((aligned(16))) char smem[];                                    \
    const int tid = threadIdx.x, lane = tid & 63;                                                  \
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                     \
    const int wt = wave >> 2, wq = wave & 3, l31 = lane & 31, hi = lane >> 5;                      \
    const int m0 = blockIdx.x * 128, K = a.K, KT = K / 64;                                         \
    char *ring = smem;                                                                             \
    const half_t *Abase = a.A + (size_t)m0 * K;                                                    \
    unsigned loffK[2];                                                                             \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                \
        const int r = (wave * 2 + i) * 8 + (lane >> 3);                                            \
        const int ch = (lane & 7) ^ ((r >> 1) & 7);                                                \
        loffK[i] = (unsigned)(r * K + ch * 8) * 2u;                                                \
    }                                                                                              \
    int aW[4], aY[4];                                                                              \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                             \
        aW[kk] = off64(wq * 32 + l31, kk * 2 + hi);                                                \
        aY[kk] = off64(wt * 64 + l31, kk * 2 + hi);                                                \
    }                                                                                              \
    /* tile t = (feature tile nt = t / KT, k-tile k = t % KT) -> ring slot; q4 weights go through   */ \
    /* `pend` and are expanded into the slot by commit() one interval later (see ffn_fused.hip)     */ \
    QRegs pend = {{0, 0, 0, 0}, 0};                                                                \
    auto issue = [&](int nt, int k, int slot) {                                                    \
        char *dst = ring + slot * FF_SLOT;                                                         \
        dma_tile8(Abase + k * 64, loffK, dst, wave);                                               \
        if (WT == GW_F16) dma_tile8(a.W + (size_t)nt * 128 * K + k * 64, loffK, dst + 16384, wave); \
        else pend = q4_fetch<WT>(a.qs, a.sc, (size_t)nt * KT + k, tid);                            \
    };                                                                                             \
    auto commit = [&](int slot) {                                                                  \
        if (WT != GW_F16) q4_expand_to_lds<WT>(pend, ring + slot * FF_SLOT + 16384, tid);          \
    };

template <int NT, int WT>
__global__ __launch_bounds__(512, 2) void proj_ln_kernel(PanelArgs a) {
    PANEL_COMMON_SETUP
    constexpr int H = 128 * NT;
    float *cb = (float *)(smem + FF_RING), *cg = cb + H, *cbeta = cg + H, *red = cbeta + H;
    for (int i = tid; i < H; i += 512) { cb[i] = a.bias[i]; cg[i] = a.gamma[i]; cbeta[i] = a.beta[i]; }
    const int ntiles = NT * KT;
    issue(0, 0, 0);
    commit(0);
    if (ntiles > 1) issue(KT > 1 ? 0 : 1, KT > 1 ? 1 : 0, 1);    // its q4 part is committed in interval 0

    f32x16 acc[NT][2];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][j][r] = 0.f;

    int t = 0, slot = 0, nt2 = KT > 2 ? 0 : (KT > 1 ? 1 : 2), k2 = 2 % KT;   // (nt2, k2) = coordinates of tile t + 2
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        for (int k = 0; k < KT; ++k, ++t) {
            if (t + 1 < ntiles) wait_vm_barrier<4>(); else wait_vm_barrier<0>();
            if (t + 1 < ntiles) commit(slot == 2 ? 0 : slot + 1);
            if (t + 2 < ntiles) {
                int s2 = slot + 2; s2 = s2 >= 3 ? s2 - 3 : s2;
                issue(nt2, k2, s2);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_slot(ring + slot * FF_SLOT, aW, aY, acc[n]);
            slot = slot == 2 ? 0 : slot + 1;
            if (++k2 == KT) { k2 = 0; ++nt2; }
        }
    }
    ln_epilogue<NT>(acc, cb, cg, cbeta, red, a.resid + (size_t)m0 * H, a.out + (size_t)m0 * H, ring, tid, wt, wq, l31, hi);
}

template <int WT>
__global__ __launch_bounds__(512, 2) void panel_store_kernel(PanelArgs a) {
    PANEL_COMMON_SETUP
    const int N = a.N, NTN = (N + 127) / 128;
    half_t *Cs = (half_t *)(smem + FF_RING);                  // [128][128] f16 staging, 16-B chunk ^ (tok & 15)
    float *cb = (float *)(smem + FF_RING + 32768);            // bias[NTN*128]
    for (int i = tid; i < NTN * 128; i += 512) cb[i] = i < N ? a.bias[i] : 0.f;
    const int ntiles = NTN * KT;
    issue(0, 0, 0);
    commit(0);
    if (ntiles > 1) issue(KT > 1 ? 0 : 1, KT > 1 ? 1 : 0, 1);    // its q4 part is committed in interval 0

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // staged tile -> global memory as full 256-byte rows (16 lanes per row)
    auto flush = [&](int nt) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int idx = s * 512 + tid, tok = idx >> 4, chunk = idx & 15;
            const int f0 = nt * 128 + chunk * 8;
            if (f0 < N)
                *(uint4 *)(a.out + ((size_t)m0 + tok) * N + f0) = *(const uint4 *)((const char *)Cs + off_hc(tok, chunk));
        }
    };

    int t = 0, slot = 0, nt2 = KT > 2 ? 0 : (KT > 1 ? 1 : 2), k2 = 2 % KT;
    for (int nt = 0; nt < NTN; ++nt) {
        for (int k = 0; k < KT; ++k, ++t) {
            if (t + 1 < ntiles) wait_vm_barrier<4>(); else wait_vm_barrier<0>();
            if (t + 1 < ntiles) commit(slot == 2 ? 0 : slot + 1);
            // the previous feature tile was staged before this barrier: write it out first, so that the
            // stores are OLDER than the DMA pieces issued below (keeps the counted waits tight)
            if (k == 0 && nt > 0) flush(nt - 1);
            if (t + 2 < ntiles) {
                int s2 = slot + 2; s2 = s2 >= 3 ? s2 - 3 : s2;
                issue(nt2, k2, s2);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_slot(ring + slot * FF_SLOT, aW, aY, acc);
            slot = slot == 2 ? 0 : slot + 1;
            if (++k2 == KT) { k2 = 0; ++nt2; }
        }
        // ---- tile epilogue: + bias, f16, into the staging tile (flushed after the next barrier; it is
        // rewritten only KT barriers later)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int fl = wq * 32 + 8 * g + 4 * hi;
            const f32x4 bv = *(const f32x4 *)(cb + nt * 128 + fl);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int tok = wt * 64 + j * 32 + l31;
                f16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (_Float16)(acc[j][4 * g + e] + bv[e]); acc[j][4 * g + e] = 0.f; }
                *(f16x4 *)((char *)Cs + off_hc(tok, fl >> 3) + (fl & 4) * 2) = o;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    flush(NTN - 1);
}

bool panel_gemm_supported(const GemmWeight &W, bool with_ln) {
    if (W.K % 64 != 0 || W.K < 64) return false;
    if (with_ln) return W.N % 128 == 0 && W.N <= 384;
    return W.N % 8 == 0 && W.N_pad <= 8192;
}

void launch_proj_ln(const GemmWeight &W, const half_t *A, const float *bias, const half_t *resid, const float *gamma,
                    const float *beta, half_t *out, int M_pad, hipStream_t stream) {
    PanelArgs a;
    a.A = A; a.W = W.w16; a.qs = W.qs; a.sc = W.sc; a.bias = bias; a.resid = resid; a.gamma = gamma; a.beta = beta; a.out = out;
    a.N = W.N; a.K = W.K;
    const int NT = W.N / 128;
    const size_t lds = FF_RING + (size_t)(3 * W.N + 512) * sizeof(float);
    const dim3 grid(M_pad / 128), block(512);
    static DeviceFlags configured[3][4];
    auto go = [&](auto kernel) {
        configure_once(configured[W.type][NT], [&] { (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, a);
    };
#define PROJ_NT(WTV)                                                     \
    switch (NT) {                                                         \
        case 1: go(proj_ln_kernel<1, WTV>); break;                        \
        case 2: go(proj_ln_kernel<2, WTV>); break;                        \
        default: go(proj_ln_kernel<3, WTV>); break;                       \
    }
    if (W.type == GW_F16) { PROJ_NT(GW_F16) } else if (W.type == GW_Q4_0) { PROJ_NT(GW_Q4_0) } else { PROJ_NT(GW_Q4_1) }
#undef PROJ_NT
}

void launch_panel_store(const GemmWeight &W, const half_t *A, const float *bias, half_t *out, int M_pad, hipStream_t stream) {
    PanelArgs a;
    a.A = A; a.W = W.w16; a.qs = W.qs; a.sc = W.sc; a.bias = bias; a.resid = nullptr; a.gamma = nullptr; a.beta = nullptr; a.out = out;
    a.N = W.N; a.K = W.K;
    const size_t lds = FF_RING + 32768 + (size_t)W.N_pad * sizeof(float);
    static DeviceFlags configured[3];
    auto go = [&](auto kernel) {
        configure_once(configured[W.type], [&] { (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        hipLaunchKernelGGL(kernel, dim3(M_pad / 128), dim3(512), lds, stream, a);
    };
    if (W.type == GW_F16) go(panel_store_kernel<GW_F16>);
    else if (W.type == GW_Q4_0) go(panel_store_kernel<GW_Q4_0>);
    else go(panel_store_kernel<GW_Q4_1>);
}

}  // namespace bert_hip
