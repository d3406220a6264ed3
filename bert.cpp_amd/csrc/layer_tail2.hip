// layer_tail2.hip — layer_tail.hip's operation with TWO waves per SIMD (gfx950, f16 weights, H = 256 / 384):
//     y     = LayerNorm(ctx Wo^T + bo + x) * g1 + be1                     (reference bert.cpp:859-875)
//     x_out = LayerNorm(gelu(y W1^T + b1) W2^T + b2 + y) * g2 + be2       (reference bert.cpp:878-901)
//
// Why: layer_tail.hip gives a wave 32 tokens x ALL features (192 accumulator registers at H = 384), which needs the whole
// 512-register budget: one wave per SIMD, and that wave issues its MFMAs, fragment reads, DMA pieces, GELU and barrier
// waits in order — the non-MFMA stream is as long as the MFMA stream and nothing hides it.  Here a
// token block of 32 belongs to a PAIR of waves on one SIMD, 256 registers each, and one wave's fragment reads, DMA issue
// and GELU run under the other's MFMAs.
//
// How the pair splits the work (workgroup = 128 tokens = 4 token blocks x 2 waves; wave = (block tb, half fh)):
//   * out-projection and down-projection by OUTPUT FEATURES: of every [128 rows x 64 k] weight tile wave fh multiplies row
//     blocks 2 fh and 2 fh + 1 (8 MFMAs per tile), so it owns features n3 * 128 + fh * 64 .. + 64 (n3 = 0 .. NT-1): 2 NT
//     accumulator blocks = 96 registers at H = 384;
//   * up-projection by K: the LayerNorm'ed y a wave produces (its own features) IS its share of the reduction dimension,
//     so y never leaves its registers (4 NT B fragments): of every [64 rows x 128 k] tile wave fh multiplies both row
//     blocks with k-steps 4 fh .. 4 fh + 3 (8 MFMAs); the pair then swaps one 32 x 32 f32 partial block each way
//     through LDS (4 KiB), each wave finishes "its" half of the chunk (bias, GELU in packed f16) and the two GELU'ed halves
//     (two B fragments each) are swapped the same way;
//   * LayerNorm: each wave sums its own features, one float pair per token crosses LDS;
//   * the residual inputs are the accumulators' initial values (x + bo before the out-projection, y + b2 before the
//     feed-forward), so neither x nor y is staged in LDS.
// Everything that crosses between the waves of a pair is written before one of the per-tile workgroup barriers and read
// behind it, and every piece of that exchange rides between MFMAs (what a wave does between its last MFMA and a barrier keeps
// seven waves waiting).  The weight-tile stream (3-slot ring, two tiles ahead, one barrier per tile) is layer_tail.hip's.
// Status (round 2): selected with BERT_HIP_TAIL=2, level with layer_tail.hip (+-2 % per launch); without the pair exchange it
// runs 25-30 % below it (L2_ABLATE) — the exchange path is what is left to shorten.
#include "tile_stream.h"

namespace bert_hip {

namespace {

constexpr int L2_TILE = 16384;
// L2_ABLATE (tuning builds only, results are wrong): bit 0 no weight DMA, bit 1 no s_barrier, bit 2 no fragment reads, bit 3 no
// MFMAs, bit 4 no partial / GELU exchange, bit 5 no GELU arithmetic (the exchange stays),
// bit 6 no send_partial, bit 7 no absorb, bit 8 no publish, bit 9 no fetch_chunk — what a component costs is the time its removal saves (tools/variant.sh)
#ifndef L2_ABLATE
#define L2_ABLATE 0
#endif
// L2_DMA_LATE: 1 = a tile interval requests tile +2 behind its fragment reads instead of in front of them
// L2_ABSORB_EARLY: 1 = the partner's partial sums are requested in front of the tile's fragment reads (one LDS round trip
// for both) and added behind its second k-step, 0 = requested behind the third k-step
#ifndef L2_ABSORB_EARLY
#define L2_ABSORB_EARLY 1
#endif
#ifndef L2_DMA_LATE
#define L2_DMA_LATE 1
#endif

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Tail2Args {
    const half_t *ctx, *x;            // [T_pad][H]
    const half_t *wo;                 // [H_pad][H] f16
    const half_t *w1p;                // [I_pad][H] f16, k order permuted inside groups of 16 (GemmWeight::w16p)
    const half_t *w2p;                // [H_pad][I] f16, same
    const float *bo, *g1, *be1, *b1, *b2, *g2, *be2;
    half_t *out;                      // [T_pad][H]
    int I;
};

// closing barrier of a tile interval: this wave's pieces of the NEXT tile have landed (all but the newest VM: those of the
// tile after it), its LDS reads and writes are complete
template <int VM>
__device__ __forceinline__ void l2_barrier() {
#if L2_ABLATE & 2
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(VM) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" : : "n"(VM) : "memory");
#endif
}

}  // namespace

template <int NT>
__global__ __launch_bounds__(512, 1) void layer_tail2_kernel(Tail2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int H = 128 * NT, KU = 2 * NT, NB = 2 * NT, NQ = 8 * NT, NY = 4 * NT;
    const int I = a.I, NC = I / 64;                           // chunks of 64 intermediate features
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tb = wave & 3, fh = wave >> 2;                  // token block, half (waves tb and tb + 4 share a SIMD)
    const int l31 = lane & 31, hi = lane >> 5;
    const int tok_w = blockIdx.x * 128 + tb * 32;             // first token of this wave

    char *ring = smem;                                        // 3 x 16 KiB weight tiles
    char *X = ring + 3 * L2_TILE;                             // 8 x 4 KiB: up-projection partial blocks for the partner
    char *G = X + 8 * 4096;                                   // 8 x 2 KiB: GELU'ed half chunks
    float *ST = (float *)(G + 8 * 2048);                      // 8 x 32 x {sum, sum of squares}
    float *cbo = ST + 8 * 64;
    float *cg1 = cbo + H, *cbe1 = cg1 + H, *cb2 = cbe1 + H, *cg2 = cb2 + H, *cbe2 = cg2 + H, *cb1 = cbe2 + H;

    for (int i = tid; i < H; i += 512) {
        cbo[i] = a.bo[i]; cg1[i] = a.g1[i]; cbe1[i] = a.be1[i]; cb2[i] = a.b2[i]; cg2[i] = a.g2[i]; cbe2[i] = a.be2[i];
    }
    for (int i = tid; i < I; i += 512) cb1[i] = a.b1[i];

    // ---- accumulators of the out-projection start from x + bo: block b = n3 * 2 + obp holds features
    // n3*128 + fh*64 + obp*32 + 8 (r >> 2) + 4 hi + (r & 3) of token l31 in register r
    f32x16 acc2[NB];
    {
        const half_t *xr = a.x + (size_t)(tok_w + l31) * H + fh * 64 + 4 * hi;
        const float *br = a.bo + fh * 64 + 4 * hi;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int f = (b >> 1) * 128 + (b & 1) * 32 + 8 * gq;
                const f16x4 xv = *(const f16x4 *)(xr + f);
                const f32x4 bv = *(const f32x4 *)(br + f);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc2[b][4 * gq + e] = (float)xv[e] + bv[e];
            }
    }
    // attention context of this wave's tokens as B fragments, straight into registers (k order as stored)
    f16x8 bf[NQ];
    {
        const half_t *cw = a.ctx + (size_t)(tok_w + l31) * H + 8 * hi;
#pragma unroll
        for (int q = 0; q < NQ; ++q) bf[q] = *(const f16x8 *)(cw + 16 * q);
    }

    // ---- weight-tile stream: 16 pieces of 1 KiB per tile, pieces 2 wave and 2 wave + 1 are this wave's.
    // [128 rows x 64 k] tiles (128-byte rows): piece p covers rows p*8 .. +8, 16-byte chunk c of a row at chunk c ^ ((row >> 1) & 7);
    // [64 rows x 128 k] tiles (256-byte rows): piece p covers rows p*4 .. +4, chunk c at c ^ (row & 15)
    unsigned off64H[2], off64I[2], offU[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = 2 * wave + i;
        const int r64 = p * 8 + (lane >> 3), c64 = (lane & 7) ^ ((r64 >> 1) & 7);
        off64H[i] = (unsigned)(r64 * H * 2 + c64 * 16);
        off64I[i] = (unsigned)(r64 * I * 2 + c64 * 16);
        const int ru = p * 4 + (lane >> 4), cu = (lane & 15) ^ (ru & 15);
        offU[i] = (unsigned)(ru * H * 2 + cu * 16);
    }
    // (the offsets pass through an opaque copy: left alone the compiler keeps a zero-extended 64-bit pair per piece and tile
    // kind alive through the whole kernel — twenty registers that the fragments need)
    auto dma2 = [&](const half_t *base, const unsigned (&off)[2], int slot) __attribute__((always_inline)) {
        char *dst = ring + slot * L2_TILE + wave * 2048;
        if (L2_ABLATE & 1) return;
        unsigned o0 = off[0], o1 = off[1];
        asm volatile("" : "+v"(o0), "+v"(o1));
        __builtin_amdgcn_global_load_lds(AS_GLOBAL((const char *)base + o0), AS_LDS(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(AS_GLOBAL((const char *)base + o1), AS_LDS(dst + 1024), 16, 0, 0);
    };
    const half_t *const wo = a.wo, *const w1p = a.w1p, *const w2p = a.w2p;
    auto dma_proj = [&](int n3, int kt, int slot) __attribute__((always_inline)) { dma2(wo + (size_t)n3 * 128 * H + kt * 64, off64H, slot); };
    auto dma_down = [&](int c, int n3, int slot) __attribute__((always_inline)) { dma2(w2p + (size_t)n3 * 128 * I + c * 64, off64I, slot); };
    auto dma_up = [&](int c, int j, int slot) __attribute__((always_inline)) { dma2(w1p + (size_t)c * 64 * H + j * 128, offU, slot); };

    // ---- fragment addresses (byte offsets into LDS): the k-step kk of a tile is the address of k-step 0 with kk XORed into bits 5..
    // (everything that depends on the half fh is folded into these addresses, so that no register array is indexed or
    // selected by it: "block 0" of the up-projection is the row block this wave finishes, "k-steps 0, 1" of the down-projection
    // are the half chunk this wave GELU'ed)
    const unsigned aA = (unsigned)(off64(l31, hi) + fh * 8192);                                        // out-projection: + obp * 4096, ^ (kk << 5)
    const unsigned aD = aA ^ (unsigned)(fh << 6);                                                      // down-projection: logical k-step L = kk ^ 2 fh
    const unsigned aU = ((unsigned)(l31 * 256 + ((hi ^ (l31 & 15)) << 4)) ^ (unsigned)(fh << 7)) + (unsigned)(fh * 8192);   // own row block, ^ (kk' << 5)
    const int uo = fh ? -8192 : 8192;                                                                  // the other row block
    auto frag = [&](unsigned off) __attribute__((always_inline)) {
        if (L2_ABLATE & 4) { f16x8 z = (f16x8)(_Float16)0.f; asm volatile("" : "+v"(z) : "v"(off)); return z; }
        return *(const f16x8 *)(ring + off);
    };

    int slot = 0;
    // one tile interval: request tile +2 into the slot freed by the last barrier, multiply this tile, close
    auto rows_tile = [&](unsigned base, auto n3_tag, const f16x8 &b0, const f16x8 &b1, const f16x8 &b2, const f16x8 &b3, auto &&prefetch,
                         auto &&filler, auto &&early) __attribute__((always_inline)) {
        // [128 x 64] tile: acc2[n3*2 + obp] += W(row block 2 fh + obp, k-step kk) x b[kk]
        constexpr int n3 = decltype(n3_tag)::value;
        const unsigned so = (unsigned)slot * L2_TILE;
        const f16x8 bq[4] = {b0, b1, b2, b3};
        if (!L2_DMA_LATE) prefetch(slot == 0 ? 2 : slot - 1);
        early();                                              // LDS reads of the pair exchange: in front of the fragment reads, one shared round trip
        unsigned ab = base + so;                              // (opaque: the four k-step addresses are XORs made here, not kept)
        asm volatile("" : "+v"(ab));
        f16x8 F[4][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            F[kk][0] = frag(ab ^ (unsigned)(kk << 5));
            F[kk][1] = frag((ab ^ (unsigned)(kk << 5)) + 4096);
        }
        __builtin_amdgcn_sched_barrier(0);                    // every fragment of the tile is requested before the first MFMA
        if (L2_DMA_LATE) prefetch(slot == 0 ? 2 : slot - 1);  // the requests for tile +2 behind the reads: they have two intervals to land
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (L2_ABLATE & 8) { asm volatile("" : "+v"(acc2[n3 * 2][0]), "+v"(acc2[n3 * 2 + 1][0]) : "v"(F[kk][0]), "v"(F[kk][1]), "v"(bq[kk])); continue; }
            acc2[n3 * 2 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[kk][0], bq[kk], acc2[n3 * 2 + 0], 0, 0, 0);
            acc2[n3 * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[kk][1], bq[kk], acc2[n3 * 2 + 1], 0, 0, 0);
            filler(kk);
        }
    };
    auto no_filler = [](int) __attribute__((always_inline)) {};
    auto no_early = []() __attribute__((always_inline)) {};
    [[maybe_unused]] int tl = 1;                              // (BERT_HIP_TIMELINE builds: one stamp per tile interval, `make timeline`)
    TL_STAMP(0);
    auto close = [&](auto vm_tag) __attribute__((always_inline)) {
        l2_barrier<decltype(vm_tag)::value>();
        TL_STAMP(tl++);
        slot = slot == 2 ? 0 : slot + 1;
    };
    using VM2 = std::integral_constant<int, 2>;
    using VM0 = std::integral_constant<int, 0>;

    // ================================ out-projection ================================
    dma_proj(0, 0, 0);
    dma_proj(KU > 1 ? 0 : 1, KU > 1 ? 1 : 0, 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // parameters, x, ctx fragments, tiles 0 and 1
    static_for<NT * KU>([&](auto t_tag) __attribute__((always_inline)) {
        constexpr int t = decltype(t_tag)::value, n3 = t / KU, kt = t % KU, t2 = t + 2;
        rows_tile(aA, std::integral_constant<int, n3>{}, bf[4 * kt], bf[4 * kt + 1], bf[4 * kt + 2], bf[4 * kt + 3], [&](int s2) __attribute__((always_inline)) {
            if constexpr (t2 < NT * KU) dma_proj(t2 / KU, t2 % KU, s2);
            else dma_up(0, t2 - NT * KU, s2);                 // the first up-projection tiles of chunk 0
        }, no_filler, no_early);
        close(VM2{});
    });

    // ---- LayerNorm over a token's row: own features summed here, the partner's through ST
    auto layernorm_stats = [&](float &mean, float &rstd) __attribute__((always_inline)) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s1 += acc2[b][r]; s2 = __builtin_fmaf(acc2[b][r], acc2[b][r], s2); }
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (hi == 0) *(f32x2 *)(ST + wave * 64 + l31 * 2) = f32x2{s1, s2};
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const f32x2 o = *(const f32x2 *)(ST + (wave ^ 4) * 64 + l31 * 2);
        s1 += o[0]; s2 += o[1];
        mean = s1 * (1.0f / H);
        const float var = fmaxf(s2 * (1.0f / H) - mean * mean, 0.f);
        rstd = 1.0f / sqrtf(var + 1e-5f);
    };
    // feature of register r of own block b (without the 4 hi part: that is the lane's)
    auto feat0 = [&](int b, int gq) { return (b >> 1) * 128 + fh * 64 + (b & 1) * 32 + 8 * gq + 4 * hi; };

    // ================================ LayerNorm 1 -> y fragments (registers) ================================
    f16x8 yf[NY];                                             // yf[2 b + s]: registers 8 s .. 8 s + 7 of block b = k-step 2 b + s of the wave's K range
    {
        float mean, rstd;
        layernorm_stats(mean, rstd);
        const float nmr = -mean * rstd;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int f0 = feat0(b, gq);
                const f32x4 gv = *(const f32x4 *)(cg1 + f0), bv = *(const f32x4 *)(cbe1 + f0), b2v = *(const f32x4 *)(cb2 + f0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * gq + e;
                    // g * ((v - mean) * rstd) + b  =  v * (g * rstd) + (g * (-mean * rstd) + b)
                    const _Float16 y = (_Float16)__builtin_fmaf(acc2[b][r], gv[e] * rstd, __builtin_fmaf(gv[e], nmr, bv[e]));
                    yf[2 * b + (gq >> 1)][4 * (gq & 1) + e] = y;
                    acc2[b][r] = (float)y + b2v[e];           // the feed-forward's residual and bias: its accumulators' initial value
                }
            }
    }

    // ================================ feed-forward ================================
    f32x16 accU[2];
    f16x8 g[4];                                               // the GELU'ed chunk the down-projection is multiplying: k-steps 0..3
    // (16-byte unit q of lane l at q * 1 KiB + l * 16: consecutive lanes on consecutive banks; lane-major units — l * 64 + q * 16 —
    // are a 16-way bank conflict on every access, which cost a third of the kernel)
    char *const Xw = X + wave * 4096 + lane * 16, *const Xp = X + (wave ^ 4) * 4096 + lane * 16;
    char *const Gw = G + wave * 2048 + lane * 16, *const Gp = G + (wave ^ 4) * 2048 + lane * 16;

    // `c_up` = the chunk the tile belongs to: its first k-step starts the accumulators from the chunk's bias (the row block
    // this wave finishes) and from 0 (the partner's) instead of adding to cleared registers
    auto up_tile = [&](auto j_tag, int c_up, auto &&prefetch, auto &&filler) __attribute__((always_inline)) {
        // [64 x 128] tile j: accU[0] (the row block this wave finishes) and accU[1] (the partner's) += W1(rows, k-step 4 fh + kk) x yf[4 j + kk]
        constexpr int j = decltype(j_tag)::value;
        const unsigned so = (unsigned)slot * L2_TILE;
        if (!L2_DMA_LATE) prefetch(slot == 0 ? 2 : slot - 1);
        unsigned ab = aU + so;
        int uoo = uo;
        asm volatile("" : "+v"(ab), "+v"(uoo));
        f16x8 F[4][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            F[kk][0] = frag(ab ^ (unsigned)(kk << 5));
            F[kk][1] = frag((ab ^ (unsigned)(kk << 5)) + uoo);
        }
        f32x16 binit;
        if constexpr (j == 0) {
            const float *bias = cb1 + c_up * 64 + fh * 32 + 4 * hi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *(const f32x4 *)(bias + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) binit[4 * q + e] = bv[e];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (L2_DMA_LATE) prefetch(slot == 0 ? 2 : slot - 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (L2_ABLATE & 8) { asm volatile("" : "+v"(accU[0][0]), "+v"(accU[1][0]) : "v"(F[kk][0]), "v"(F[kk][1]), "v"(yf[4 * j + kk])); continue; }
            if (j == 0 && kk == 0) {
                accU[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[kk][0], yf[4 * j + kk], binit, 0, 0, 0);
                accU[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[kk][1], yf[4 * j + kk], (f32x16)0.f, 0, 0, 0);
            } else {
                accU[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[kk][0], yf[4 * j + kk], accU[0], 0, 0, 0);
                accU[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[kk][1], yf[4 * j + kk], accU[1], 0, 0, 0);
            }
            filler(kk);
        }
    };
    // the partial sums of the chunk half the partner finishes go to X (before a barrier)
    auto send_partial = [&]() __attribute__((always_inline)) {
        if (L2_ABLATE & (16 | 64)) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = accU[1][4 * q + e];
            *(f32x4 *)(Xw + q * 1024) = v;
        }
    };
    // own half of chunk c, in pieces so that no tile interval carries more than a few of them (a wave that runs the whole GELU
    // of a half chunk — 32 quarter-rate transcendentals — behind one tile keeps the other seven waiting at that tile's barrier:
    // measured, 40 % of the kernel):
    //   absorb()       own + partner's partial sums (the bias came in as the accumulators' initial value) -> 16 pre-activations,
    //                  rounded to packed f16 (the GELU's input precision)
    //   gelu_steps(i)  the GELU of the pairs of filler interval i, in place
    //   publish()      the two B fragments to G (before a barrier)
    f16x2_t pre[8];                                           // pairs 2 q, 2 q + 1 = registers 4 q .. 4 q + 3 of the half
#pragma unroll
    for (int k = 0; k < 8; ++k) pre[k] = f16x2_t{(_Float16)0.f, (_Float16)0.f};
    f32x4 xin[4];                                             // the partner's partial sums on their way in
    auto absorb_load = [&]() __attribute__((always_inline)) {
        if (L2_ABLATE & (16 | 128)) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) xin[q] = *(const f32x4 *)(Xp + q * 1024);      // four reads in flight together
    };
    auto absorb_math = [&]() __attribute__((always_inline)) {
        if (L2_ABLATE & (16 | 128)) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pre[2 * q] = f16x2_t{(_Float16)(accU[0][4 * q] + xin[q][0]), (_Float16)(accU[0][4 * q + 1] + xin[q][1])};
            pre[2 * q + 1] = f16x2_t{(_Float16)(accU[0][4 * q + 2] + xin[q][2]), (_Float16)(accU[0][4 * q + 3] + xin[q][3])};
        }
    };
    auto absorb = [&]() __attribute__((always_inline)) { absorb_load(); absorb_math(); };
    // the 8 GELU pairs of a half chunk ride in NG tile intervals: DOWN tiles 2 .. NT-1 of the previous chunk (NT - 2 of them),
    // then UP tiles 0 .. NT-2 of the next one; interval gi carries pairs gstart(gi) .. gstart(gi + 1) - 1
    constexpr int NG = 2 * NT - 3;
    auto gstart = [](int gi) constexpr { return (8 * gi + NG - 1) / NG; };
    static_assert(NG >= 1, "");
    auto gelu_filler = [&](auto gi_tag) __attribute__((always_inline)) {
        return [&](int kk) __attribute__((always_inline)) {
            constexpr int gi = decltype(gi_tag)::value, p0 = (8 * gi + NG - 1) / NG, p1 = (8 * (gi + 1) + NG - 1) / NG, np = p1 - p0;
            if (L2_ABLATE & (16 | 32)) return;
#pragma unroll
            for (int k = 0; k < np; ++k)
                if (kk == (np <= 4 ? k : k / 2)) pre[p0 + k] = gelu_pk16h(pre[p0 + k]);
        };
    };
    auto gelu_range = [&](int first, int last) __attribute__((always_inline)) {     // pairs first .. last - 1 at once (first / last chunk)
        if (L2_ABLATE & 16) return;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k >= first && k < last && !(L2_ABLATE & 32)) pre[k] = gelu_pk16h(pre[k]);
    };
    auto publish = [&]() __attribute__((always_inline)) {
        if (L2_ABLATE & (16 | 256)) return;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            f16x8 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[2 * k] = pre[4 * s2 + k][0]; o[2 * k + 1] = pre[4 * s2 + k][1]; }
            *(f16x8 *)(Gw + s2 * 1024) = o;
        }
    };
    // both halves of the GELU'ed chunk from G (behind a barrier): g[0..1] = the half this wave made, g[2..3] = the partner's
    // (the down-projection's fragment addresses visit the weight k-steps in that order)
    auto fetch_chunk = [&]() __attribute__((always_inline)) {
        if (L2_ABLATE & 512) return;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // (the half this wave made is still in `pre`: no need to read it back)
#pragma unroll
            for (int k = 0; k < 4; ++k) { g[s][2 * k] = pre[4 * s + k][0]; g[s][2 * k + 1] = pre[4 * s + k][1]; }
            g[2 + s] = *(const f16x8 *)(Gp + s * 1024);
        }
    };

    // stream order: UP(0) | UP(1) | DOWN(0) | UP(2) | DOWN(1) | ... | UP(NC-1) | DOWN(NC-2) | DOWN(NC-1), NT tiles each.
    // Life of chunk c: partial sums in UP(c) -> sent behind the first MFMAs of DOWN(c-1)'s first tile -> taken in behind its
    // second tile -> GELU pairs behind the following tiles up to UP(c+1)'s last but one -> published behind the first
    // MFMAs of UP(c+1)'s last tile -> fetched by both waves of the pair behind its barrier -> multiplied in DOWN(c).
    // (Every piece rides between MFMAs: whatever a wave does between its last MFMA and a barrier keeps seven waves waiting.)
    // ---- UP(0)  (its first two tiles were requested by the last out-projection intervals)
    static_for<NT>([&](auto j_tag) __attribute__((always_inline)) {
        constexpr int j = decltype(j_tag)::value, j2 = j + 2;
        up_tile(j_tag, 0, [&](int s2) __attribute__((always_inline)) {
            if constexpr (j2 < NT) dma_up(0, j2, s2); else dma_up(1, j2 - NT, s2);
        }, no_filler);
        if constexpr (j == NT - 1) send_partial();
        close(VM2{});
    });
    absorb();
    gelu_range(0, gstart(NT - 2));                            // (what DOWN(-1)'s tiles would have carried)

    // ---- step c: UP(c+1), then DOWN(c)
    auto step = [&](int c, auto last_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;     // c == NC - 2: no UP(c+2) to request
        static_for<NT>([&](auto j_tag) __attribute__((always_inline)) {
            constexpr int j = decltype(j_tag)::value, j2 = j + 2;
            auto pf = [&](int s2) __attribute__((always_inline)) {
                if constexpr (j2 < NT) dma_up(c + 1, j2, s2); else dma_down(c, j2 - NT, s2);
            };
            if constexpr (j < NT - 1) up_tile(j_tag, c + 1, pf, gelu_filler(std::integral_constant<int, (NT - 2) + j>{}));   // chunk c's last pairs
            else up_tile(j_tag, c + 1, pf, [&](int kk) __attribute__((always_inline)) { if (kk == 0) publish(); });                           // chunk c goes to G
            close(VM2{});
        });
        fetch_chunk();                                        // chunk c: both halves were published before the last barrier
        static_for<NT>([&](auto d_tag) __attribute__((always_inline)) {
            constexpr int d = decltype(d_tag)::value, d2 = d + 2;
            auto pf = [&](int s2) __attribute__((always_inline)) {
                if constexpr (d2 < NT) dma_down(c, d2, s2);
                else if constexpr (!LAST) dma_up(c + 2, d2 - NT, s2);
                else dma_down(c + 1, d2 - NT, s2);            // chunk c+1 is the last one: only its DOWN tiles are left
            };
            // chunk c+1: its partial sums cross to the partner behind tile 0 (they have been final since the last barrier),
            // are taken in behind tile 1, the first GELU pairs follow
            if constexpr (d == 0) rows_tile(aD, d_tag, g[0], g[1], g[2], g[3], pf, [&](int kk) __attribute__((always_inline)) { if (kk == 0) send_partial(); }, no_early);
            else if constexpr (d == 1) rows_tile(aD, d_tag, g[0], g[1], g[2], g[3], pf, [&](int kk) __attribute__((always_inline)) {
                if (L2_ABSORB_EARLY ? kk == 1 : kk == 3) absorb_math();
                if (!L2_ABSORB_EARLY && kk == 2) absorb_load();
            }, [&]() __attribute__((always_inline)) { if (L2_ABSORB_EARLY) absorb_load(); });
            else rows_tile(aD, d_tag, g[0], g[1], g[2], g[3], pf, gelu_filler(std::integral_constant<int, d - 2>{}), no_early);
            close(VM2{});
        });
    };
    for (int c = 0; c + 2 < NC; ++c) step(c, std::false_type{});
    step(NC - 2, std::true_type{});
    // ---- DOWN(NC-1): the rest of its GELU at once
    gelu_range(gstart(NT - 2), 8);
    publish();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    fetch_chunk();
    static_for<NT>([&](auto d_tag) __attribute__((always_inline)) {
        constexpr int d = decltype(d_tag)::value, d2 = d + 2;
        rows_tile(aD, d_tag, g[0], g[1], g[2], g[3], [&](int s2) __attribute__((always_inline)) {
            if constexpr (d2 < NT) dma_down(NC - 1, d2, s2);
        }, no_filler, no_early);
        if constexpr (d2 < NT) close(VM2{}); else close(VM0{});
    });

    // ================================ LayerNorm 2 -> rows through LDS -> HBM ================================
    {
        float mean, rstd;
        layernorm_stats(mean, rstd);
        const float nmr = -mean * rstd;
        // a wave passes its 32 tokens x 64 features of every n3 through a private 4 KiB area of the (idle) ring: every global
        // store is 16 bytes of a full 128-byte row segment
        char *stg = ring + wave * 4096;
        half_t *ow = a.out + (size_t)tok_w * H + fh * 64;
#pragma unroll
        for (int n3 = 0; n3 < NT; ++n3) {
#pragma unroll
            for (int obp = 0; obp < 2; ++obp)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int b = n3 * 2 + obp, f0 = feat0(b, gq);
                    const f32x4 gv = *(const f32x4 *)(cg2 + f0), bv = *(const f32x4 *)(cbe2 + f0);
                    f16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o[e] = (_Float16)__builtin_fmaf(acc2[b][4 * gq + e], gv[e] * rstd, __builtin_fmaf(gv[e], nmr, bv[e]));
                    *(f16x4 *)(stg + l31 * 128 + (((obp * 4 + gq) ^ (l31 & 7)) << 4) + hi * 8) = o;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                const uint4 v = *(const uint4 *)(stg + row * 128 + ((ch ^ (row & 7)) << 4));
                *(uint4 *)(ow + (size_t)row * H + n3 * 128 + ch * 8) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

bool layer_tail2_supported(const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2) {
    const int H = W1.K, I = W1.N;
    if (Wo.type != GW_F16 || W1.type != GW_F16 || W2.type != GW_F16 || !W1.w16p || !W2.w16p) return false;
    if (Wo.N != H || Wo.K != H || W2.N != H || W2.K != I) return false;
    if (H % 128 != 0 || H < 256 || H > 384 || I % 64 != 0 || I < 192) return false;
    const size_t lds = (size_t)3 * L2_TILE + 8 * 4096 + 8 * 2048 + 8 * 64 * 4 + (size_t)(6 * H + I) * sizeof(float);
    return lds <= 160 * 1024;
}

void launch_layer_tail2(const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2, const half_t *ctx, const half_t *x,
                        const float *bo, const float *g1, const float *be1, const float *b1, const float *b2,
                        const float *g2, const float *be2, half_t *out, int M_pad, hipStream_t stream) {
    Tail2Args a;
    a.ctx = ctx; a.x = x; a.wo = Wo.w16; a.w1p = W1.w16p; a.w2p = W2.w16p;
    a.bo = bo; a.g1 = g1; a.be1 = be1; a.b1 = b1; a.b2 = b2; a.g2 = g2; a.be2 = be2; a.out = out;
    a.I = W1.N;
    const int H = W1.K;
    const size_t lds = (size_t)3 * L2_TILE + 8 * 4096 + 8 * 2048 + 8 * 64 * 4 + (size_t)(6 * H + a.I) * sizeof(float);
    static bool configured[4][MAX_HIP_DEVICES] = {};
    auto go = [&](auto kernel, int nt) {
        if (first_launch_on_device(configured[nt]))
            (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kernel, dim3(M_pad / 128), dim3(512), lds, stream, a);
        TL_DUMP(M_pad >= 128 * 256, 200);
    };
    if (H == 256) go(layer_tail2_kernel<2>, 2); else go(layer_tail2_kernel<3>, 3);
}

}  // namespace bert_hip
