// layer_tail.hip — everything of an encoder layer after the attention, in ONE kernel in which every wave OWNS its
// tokens (gfx950, f16 weights):
//     y     = LayerNorm(ctx Wo^T + bo + x) * g1 + be1                     (reference bert.cpp:859-875)
//     x_out = LayerNorm(gelu(y W1^T + b1) W2^T + b2 + y) * g2 + be2       (reference bert.cpp:878-901)
//
// Workgroup = 128 tokens = 4 waves x 32 tokens, one wave per SIMD with the whole 512-register budget.  A wave
// computes ALL output features of its 32 tokens, so
//   * the MFMA B operand (tokens) never goes through LDS between the GEMMs: the LayerNorm'ed y and the GELU'ed
//     intermediate chunk are produced in the accumulator layout (lane = token, 4-feature runs) and, converted to
//     f16, ARE the next GEMM's B fragments — the weights are stored with the matching order inside every group
//     of 16 k (GemmWeight::w16p), so no shuffle is needed;
//   * both LayerNorms are wave-local (row statistics = the lane's registers + one cross-half shuffle), y is
//     never written to HBM, and there is no barrier between a GEMM and its epilogue;
//   * the GELU of chunk c+1 is issued as VALU filler between the MFMAs of chunk c's down-projection (one set of up-
//     projection accumulators, two GELU'ed chunks in flight), so the matrix pipe waits for the first chunk's GELU only.
// The intermediate dimension is processed in chunks of 64 features (up-projection tile = [64 rows x 128 k], down-
// projection tile = [128 rows x 64 k]): 32 up-projection accumulator registers next to the 192 output accumulators fit
// the 256 AccVGPRs, and the tile loop runs without a single spill — a scratch reload inside the loop would queue behind
// the weight tiles in flight (vmcnt is in-order) and serialise the whole pipeline.
// Only the weight tiles (16 KiB) are shared: they stream through a 3-slot LDS ring by LDS-DMA,
// two tiles ahead, one barrier per tile.  Each wave hides its own LDS latency (tile_stream.h, hand-issued reads):
// the fragments of a tile are read in two halves, and the second half's MFMAs run after the next barrier, under
// the reads of the next tile.
//
// LDS: 4 x 24 KiB wave-private staging (x rows -> y fragments -> output rows), 48 KiB ring, biases / LayerNorm
// parameters.  Registers (H = 384): 192 output accumulators + 32 up-projection accumulators + 2 x 16 (GELU'ed
// chunks) + 64 weight fragments + 32 y fragments.
#include "tile_stream.h"

namespace bert_hip {

namespace {

constexpr int LT_TILE = 16384;

// LT_ABLATE (tuning builds only, results are wrong): bit 0 no weight DMA, bit 1 no GELU arithmetic, bit 2 no tile barrier,
// bit 3 no fragment reads, bit 4 no MFMAs, bit 5 no GELU filler in the last interval of a chunk, bit 6 no GELU filler at all, bit 7 no LDS-read waits — what a component costs is the time its removal saves (tools/variant.sh)
#ifndef LT_ABLATE
#define LT_ABLATE 0
#endif
// LT_GROLE: -1 = the two GELU'ed-chunk buffers alternate roles with the chunk parity; 0 / 1 = the down-projection always
// reads g[LT_GROLE], the filler always writes the other one, which is copied over once per chunk
#ifndef LT_GROLE
#define LT_GROLE 1
#endif
// LT_GELU16: the GELU of the intermediate is evaluated in packed f16 (two elements per instruction where the ISA has a
// packed form; the reference reads it from an f16 table, ggml_gelu_f16), 0 = f32 arithmetic per element
#ifndef LT_GELU16
#define LT_GELU16 1
#endif
// LT_BIASINIT: the up-projection accumulators start from the bias of their chunk (written where they used to be
// zeroed) instead of the bias being added in front of the GELU
#ifndef LT_DMA_SPREAD
#define LT_DMA_SPREAD 1
#endif
#ifndef LT_BIASINIT
#define LT_BIASINIT 1
#endif
struct TailArgs {
    const half_t *ctx, *x;            // [T_pad][H]
    const half_t *wo;                 // [H_pad][H] f16
    const half_t *w1p;                // [I_pad][H] f16, k order permuted inside groups of 16 (GemmWeight::w16p)
    const half_t *w2p;                // [H_pad][I] f16, same
    const float *bo, *g1, *be1, *b1, *b2, *g2, *be2;
    half_t *out;                      // [T_pad][H]
    int I;
};

// tile kinds of the stream
constexpr int K_NONE = 0, K_PROJ = 1, K_UP = 2, K_DOWN = 3;
template <int KIND, int I0, int I1>
struct TileDesc {
    static constexpr int kind = KIND, i0 = I0, i1 = I1;
};
using NoTile = TileDesc<K_NONE, 0, 0>;

// (LT_ABLATE bit 7: no wait for LDS reads anywhere — what their exposed latency costs: 1 %)
#if LT_ABLATE & 128
#define LT_LGKM(n) "15"
#else
#define LT_LGKM(n) #n
#endif
__device__ __forceinline__ void wait_lgkm8(f16x8 (&f)[8]) {
    asm volatile("s_waitcnt lgkmcnt(" LT_LGKM(8) ")"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : : "memory");
}
__device__ __forceinline__ void wait_lgkm12(f16x8 (&f)[8], f16x8 (&y)[4]) {
    asm volatile("s_waitcnt lgkmcnt(" LT_LGKM(12) ")"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                   "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]) : : "memory");
}
// GELU filler placement: the 16 element pairs of gelu(c+1) sit behind the MFMAs of the NT intervals of DOWN(c); the
// first 8 MFMAs of interval 0 still finish UP(c+1), which leaves 8 + (NT-1)*16 usable slots.  Pair q sits at usable
// slot floor(q * total / 16).  lt_pair_of: the pair at MFMA slot `slot16` (0..15) of interval d, or -1;
// lt_pair_rank: how many pairs sit in slots grp*8 .. grp*8+k-1 of that interval.
template <int NT>
constexpr int lt_pair_of(int d, int slot16) {
    static_assert(NT >= 2, "16 pairs need at least 16 usable slots");
    const int first = d == 0 ? 8 : 0, before = d == 0 ? 0 : 8 + (d - 1) * 16, total = 8 + (NT - 1) * 16;
    if (slot16 < first) return -1;
    const int u = before + slot16 - first;
    const int q = (u * 16 + total - 1) / total;               // smallest q with q * total / 16 >= u
    return (q < 16 && (q * total) / 16 == u) ? q : -1;
}
template <int NT>
constexpr int lt_pair_rank(int d, int grp, int k) {
    int n = 0;
    for (int kk = 0; kk < k; ++kk) n += lt_pair_of<NT>(d, grp * 8 + kk) >= 0;
    return n;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ f32x2 lds_read_b64_u(unsigned addr) {
    f32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// hands N hand-read bias pairs to their users; WAIT: after retiring everything but the newest 8 LDS reads
// (every operand names a different register pair: a repeated operand would be COPIED before the asm, i.e. before its
// read has landed)
#define LT_FENCE(TEXT, ...) asm volatile(TEXT : __VA_ARGS__ : : "memory")
template <int N, bool WAIT>
__device__ __forceinline__ void bias_fence_n(f32x2 (&b)[6]) {
    static_assert(N >= 0 && N <= 6, "");
    if constexpr (WAIT) {
        if constexpr (N == 1) LT_FENCE("s_waitcnt lgkmcnt(8)", "+v"(b[0]));
        if constexpr (N == 2) LT_FENCE("s_waitcnt lgkmcnt(8)", "+v"(b[0]), "+v"(b[1]));
        if constexpr (N == 3) LT_FENCE("s_waitcnt lgkmcnt(8)", "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
        if constexpr (N == 4) LT_FENCE("s_waitcnt lgkmcnt(8)", "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        if constexpr (N == 5) LT_FENCE("s_waitcnt lgkmcnt(8)", "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]));
        if constexpr (N == 6) LT_FENCE("s_waitcnt lgkmcnt(8)", "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]));
    } else {
        if constexpr (N == 1) LT_FENCE("", "+v"(b[0]));
        if constexpr (N == 2) LT_FENCE("", "+v"(b[0]), "+v"(b[1]));
        if constexpr (N == 3) LT_FENCE("", "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
        if constexpr (N == 4) LT_FENCE("", "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        if constexpr (N == 5) LT_FENCE("", "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]));
        if constexpr (N == 6) LT_FENCE("", "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]));
    }
}
// tile barrier: this wave's share of the next tile has landed (all but the newest VM pieces), every hand-issued
// read has returned (the second-half fragments are named: their MFMAs run after the barrier)
template <int VM>
__device__ __forceinline__ void tile_barrier(f16x8 (&f)[8], f16x8 (&y)[4]) {
#if LT_ABLATE & 4
#define LT_BARRIER_TEXT "s_waitcnt vmcnt(%12) lgkmcnt(" LT_LGKM(0) ")"
#else
#define LT_BARRIER_TEXT "s_waitcnt vmcnt(%12) lgkmcnt(" LT_LGKM(0) ")\n\ts_barrier"
#endif
    asm volatile(LT_BARRIER_TEXT
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                   "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]) : "n"(VM) : "memory");
}

}  // namespace

template <int NT>
__global__ __launch_bounds__(256, 1) void layer_tail_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int H = 128 * NT, KU = 2 * NT, NB = 4 * NT, NQ = 8 * NT;
    constexpr int S_BYTES = 64 * H;                           // 32 tokens x H halfs per wave
    const int I = a.I, NC = I / 64;                           // chunks of 64 intermediate features
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int tok_w = blockIdx.x * 128 + wave * 32;           // first token of this wave

    char *S = smem + wave * S_BYTES;
    char *ring = smem + 4 * S_BYTES;
    float *cbo = (float *)(ring + 3 * LT_TILE);
    float *cg1 = cbo + H, *cbe1 = cg1 + H, *cb2 = cbe1 + H, *cg2 = cb2 + H, *cbe2 = cg2 + H, *cb1 = cbe2 + H;

    // ---- prologue: parameters -> LDS; this wave's x rows -> S (16-byte unit of (token, 8-feature chunk c) at
    // position c*32 + ((token + c) & 31): conflict-free 8-byte reads in the accumulator layout)
    for (int i = tid; i < H; i += 256) {
        cbo[i] = a.bo[i]; cg1[i] = a.g1[i]; cbe1[i] = a.be1[i]; cb2[i] = a.b2[i]; cg2[i] = a.g2[i]; cbe2[i] = a.be2[i];
    }
    for (int i = tid; i < I; i += 256) cb1[i] = a.b1[i];
    {
        const half_t *xw = a.x + (size_t)tok_w * H;
#pragma unroll
        for (int p = 0; p < H / 16; ++p) {
            const int u = p * 64 + lane, c = u >> 5, tok = ((u & 31) - c) & 31;
            __builtin_amdgcn_global_load_lds(AS_GLOBAL(xw + (size_t)tok * H + c * 8), AS_LDS(S + p * 1024), 16, 0, 0);
        }
    }
    // attention context of this wave's tokens as B fragments, straight into registers (k order as stored)
    f16x8 bf[NQ];
    {
        const half_t *cw = a.ctx + (size_t)(tok_w + l31) * H + 8 * hi;
#pragma unroll
        for (int q = 0; q < NQ; ++q) bf[q] = *(const f16x8 *)(cw + 16 * q);
    }

    // ---- weight-tile stream: [128 rows x 64 k] tiles, 16 pieces of 1 KiB, 4 per wave
    // ([64 rows x 128 k] for the up-projection: 256-byte rows, 16-byte chunk ^ (row & 15), 4 rows per piece).
    // The feed-forward's lane offsets and addresses are (re)computed after the out-projection from an opaque copy of
    // the lane id: values that live across the out-projection (where the 96 context registers are alive) get
    // spilled for their whole life, and a reload inside the tile loop queues behind the DMA in flight.
    // Lane offsets of the DMA pieces, kept in few registers: piece i of this wave covers rows (wave*4+i)*8.. of a
    // [128 x 64] tile; the row part of i goes into the scalar base, the swizzle alternates between two values
    // (even / odd i).  For the [64 x 128] up-projection tile (rows (wave*4+i)*4..) the swizzle of piece i is the one
    // of piece 0 with i XORed into bits 6..7.
    unsigned offHe, offHo, offIe, offIo, offUb, offUx;
    auto lane_offsets = [&](int ln) __attribute__((always_inline)) {
        const int ch0 = (ln & 7) ^ (ln >> 4), row = wave * 32 + (ln >> 3);
        offHe = (unsigned)(row * H * 2 + ch0 * 16);
        offHo = (unsigned)(row * H * 2 + (ch0 ^ 4) * 16);
        offIe = (unsigned)(row * I * 2 + ch0 * 16);
        offIo = (unsigned)(row * I * 2 + (ch0 ^ 4) * 16);
        offUb = (unsigned)((wave * 16 + (ln >> 4)) * H * 2);
        offUx = (unsigned)(((ln & 15) ^ (ln >> 4)) * 16);
    };
    lane_offsets(lane);
    const half_t *const wo = a.wo, *const w1p = a.w1p, *const w2p = a.w2p;     // (locals: keeps the argument struct out of scratch)
    // PC = piece of this wave to request (0..3), or -1 for all four
    using ALLP = std::integral_constant<int, -1>;
    auto dma128 = [&](const half_t *base, unsigned oe, unsigned oo, int row_bytes, int slot, auto pc_tag) __attribute__((always_inline)) {
        constexpr int PC = decltype(pc_tag)::value;
        char *dst = ring + slot * LT_TILE + wave * 4096;
        static_for<4>([&](auto i_tag) __attribute__((always_inline)) {
            constexpr int i = decltype(i_tag)::value;
            if constexpr ((PC < 0 || PC == i) && !(LT_ABLATE & 1))
                __builtin_amdgcn_global_load_lds(AS_GLOBAL((const char *)base + (size_t)i * 8 * row_bytes + ((i & 1) ? oo : oe)),
                                                 AS_LDS(dst + i * 1024), 16, 0, 0);
        });
    };
    auto dma_proj = [&](int n3, int kt, int slot, auto pc) __attribute__((always_inline)) { dma128(wo + (size_t)n3 * 128 * H + kt * 64, offHe, offHo, H * 2, slot, pc); };
    auto dma_down = [&](int c, int n3, int slot, auto pc) __attribute__((always_inline)) { dma128(w2p + (size_t)n3 * 128 * I + c * 64, offIe, offIo, I * 2, slot, pc); };
    auto dma_up = [&](int c, int j, int slot, auto pc_tag) __attribute__((always_inline)) {
        constexpr int PC = decltype(pc_tag)::value;
        const half_t *base = w1p + (size_t)c * 64 * H + j * 128;
        char *dst = ring + slot * LT_TILE + wave * 4096;
        static_for<4>([&](auto i_tag) __attribute__((always_inline)) {
            constexpr int i = decltype(i_tag)::value;
            if constexpr ((PC < 0 || PC == i) && !(LT_ABLATE & 1))
                __builtin_amdgcn_global_load_lds(AS_GLOBAL((const char *)base + (size_t)i * 4 * H * 2 + (offUb + (offUx ^ (unsigned)(i << 6)))),
                                                 AS_LDS(dst + i * 1024), 16, 0, 0);
        });
    };

    // ---- per-lane LDS addresses
    // weight fragment of k-step kk: the swizzles are XORs of the 16-byte chunk index 2*kk + hi, so the address of
    // k-step kk is the address of k-step 0 with kk XORed into bits 5.. (one register per tile shape instead of 4 / 8)
    unsigned aA0 = lds_addr(ring) + off64(l31, hi);                                   // [128 x 64]: + ob * 4 KiB + slot * 16 KiB
    unsigned aU0 = lds_addr(ring) + l31 * 256 + ((hi ^ (l31 & 15)) << 4);            // [64 x 128]: + fb * 8 KiB + slot * 16 KiB
    unsigned aY = lds_addr(S) + lane * 16;                    // y fragment q at + q * 1 KiB
    unsigned aB = lds_addr(cb1) + hi * 16;                    // up-projection biases of a lane's features: + chunk * 256 B

    f32x16 acc2[NB], accU[2];
    f16x8 g[2][4];                                            // GELU'ed chunks (two in flight) as B fragments
    f16x8 F[2][8], Y[2][4];                                   // weight fragments [half][...], y fragments [half][i]
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[n][r] = 0.f;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int j = 0; j < 8; ++j) F[hh][j] = (f16x8)(_Float16)0;
        Y[hh][0] = Y[hh][1] = Y[hh][2] = Y[hh][3] = (f16x8)(_Float16)0;
    }

    // fragment reads of one half of the tile in ring slot `slot_off`.  [128 x 64] tiles: k-steps 2*half + i, 4 row
    // blocks, F[half][ob*2 + i].  Up-projection [64 x 128] tiles: k-steps 4*half + i, 2 row blocks, F[half][fb*4 + i],
    // plus the 4 y fragments of those k-steps.
    auto read_half = [&](auto desc, auto half_tag, unsigned slot_off) __attribute__((always_inline)) {
        using D = decltype(desc);
        constexpr int half = decltype(half_tag)::value;
        if constexpr (LT_ABLATE & 8) {
        } else if constexpr (D::kind == K_UP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned ad = (aU0 ^ (unsigned)((half * 4 + i) << 5)) + slot_off;
                F[half][0 + i] = lds_read_b128_u<0>(ad);
                F[half][4 + i] = lds_read_b128_u<8192>(ad);
            }
            Y[half][0] = lds_read_b128_u<(8 * D::i1 + 4 * half + 0) * 1024>(aY);
            Y[half][1] = lds_read_b128_u<(8 * D::i1 + 4 * half + 1) * 1024>(aY);
            Y[half][2] = lds_read_b128_u<(8 * D::i1 + 4 * half + 2) * 1024>(aY);
            Y[half][3] = lds_read_b128_u<(8 * D::i1 + 4 * half + 3) * 1024>(aY);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned ad = (aA0 ^ (unsigned)((half * 2 + i) << 5)) + slot_off;
                F[half][0 + i] = lds_read_b128_u<0>(ad);
                F[half][2 + i] = lds_read_b128_u<4096>(ad);
                F[half][4 + i] = lds_read_b128_u<8192>(ad);
                F[half][6 + i] = lds_read_b128_u<12288>(ad);
            }
        }
    };
    // the 8 MFMAs of one half of a tile.  PROJ <n3, kt>: acc2[n3*4+ob] += Wo x ctx;  UP <0, j>: accU += W1 x y;
    // DOWN <gpar, n3>: acc2[n3*4+ob] += W2 x gelu chunk in g[gpar]
    auto mma_half = [&](auto desc, auto half_tag, auto &&fill) __attribute__((always_inline)) {
        using D = decltype(desc);
        constexpr int half = decltype(half_tag)::value;
        // fill(k): VALU filler issued behind the k-th MFMA of the half (k = 0..7), in program order
        if constexpr (LT_ABLATE & 16) {
            static_for<8>([&](auto k_tag) __attribute__((always_inline)) { fill(k_tag); });
        } else if constexpr (D::kind == K_UP) {
            static_for<8>([&](auto k_tag) __attribute__((always_inline)) {
                constexpr int k = decltype(k_tag)::value, i = k >> 1, fb = k & 1;
                accU[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[half][fb * 4 + i], Y[half][i], accU[fb], 0, 0, 0);
                fill(k_tag);
            });
        } else {
            static_for<8>([&](auto k_tag) __attribute__((always_inline)) {
                constexpr int k = decltype(k_tag)::value, i = k >> 2, ob = k & 3;
                if constexpr (D::kind == K_PROJ)
                    acc2[D::i0 * 4 + ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[half][ob * 2 + i], bf[4 * D::i1 + 2 * half + i], acc2[D::i0 * 4 + ob], 0, 0, 0);
                else
                    acc2[D::i1 * 4 + ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[half][ob * 2 + i], g[D::i0][2 * half + i], acc2[D::i1 * 4 + ob], 0, 0, 0);
                fill(k_tag);
            });
        }
    };

    int slot = 0;                                             // ring slot of the current tile
    [[maybe_unused]] int tl = 1;
    TL_STAMP(0);
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    // One interval = one tile.  On entry the tile is complete in LDS for every wave.  `prev` = the tile whose second
    // half is still owed (NoTile after a drain), `prefetch(slot)` requests tile +2, `filler(group, k)` is VALU work
    // issued behind the k-th MFMA of the owed half (group 0) / of this tile's first half (group 1), VM = DMA pieces of this wave that may stay in flight at the closing barrier.
    auto interval = [&](auto cur, auto prev, auto vm_tag, auto &&prefetch, auto &&filler, auto &&pre, auto &&fence) __attribute__((always_inline)) {
        using C = decltype(cur);
        using P = decltype(prev);
        constexpr int VM = decltype(vm_tag)::value;
        const unsigned so = (unsigned)slot * LT_TILE;
        TL_STAMP(tl++);
        const int pslot = slot == 0 ? 2 : slot - 1;           // slot + 2 (mod 3): read one tile ago, free since the barrier
        // pre(group) issues the LDS reads the fillers of that group need (by hand, like the fragments: a compiler-issued
        // read is retired with lgkmcnt(0), which would wait for every fragment read in flight); fence(group) hands them over
        pre(H0{});
        read_half(cur, H0{}, so);
        fence(H0{});
        // the 4 DMA pieces go behind MFMAs 1, 3, 5, 7 of the owed half (a piece costs >= 60 issue cycles, an MFMA covers 32)
        // LT_DMA_SPREAD: behind MFMAs 1 and 5 of the owed half and of this tile's first half instead
        if constexpr (P::kind != K_NONE)
            mma_half(prev, H1{}, [&](auto k) __attribute__((always_inline)) {
                constexpr int kk = decltype(k)::value;
                if constexpr (LT_DMA_SPREAD) {
                    if constexpr (kk == 1 || kk == 5) prefetch(pslot, std::integral_constant<int, (kk >> 2)>{});
                } else {
                    if constexpr (kk & 1) prefetch(pslot, std::integral_constant<int, (kk >> 1)>{});
                }
                filler(H0{}, k);
            });
        else prefetch(pslot, ALLP{});
        pre(H1{});
        read_half(cur, H1{}, so);
        if constexpr (C::kind == K_UP) wait_lgkm12(F[0], Y[0]); else wait_lgkm8(F[0]);
        fence(H1{});
        mma_half(cur, H0{}, [&](auto k) __attribute__((always_inline)) {
            constexpr int kk = decltype(k)::value;
            if constexpr (LT_DMA_SPREAD && P::kind != K_NONE && (kk == 1 || kk == 5)) prefetch(pslot, std::integral_constant<int, 2 + (kk >> 2)>{});
            filler(H1{}, k);
        });
        // the first-half MFMAs cover the second-half reads; left alone the scheduler sinks them below the barrier, whose
        // lgkmcnt(0) then waits for those reads with an empty matrix pipe
        __builtin_amdgcn_sched_barrier(0);
        tile_barrier<VM>(F[1], Y[1]);
        slot = slot == 2 ? 0 : slot + 1;
    };
    auto nothing = [](auto, auto) __attribute__((always_inline)) {};
    auto nothing1 = [](auto) __attribute__((always_inline)) {};
    auto drain = [&](auto prev) __attribute__((always_inline)) { mma_half(prev, H1{}, [](auto) __attribute__((always_inline)) {}); };
    using VM4 = std::integral_constant<int, 4>;
    using VM0 [[maybe_unused]] = std::integral_constant<int, 0>;

    // GELU of fragment j (features 32*(j>>1) + 16*(j&1) .. +16 of chunk c) of the up-projection accumulators into
    // g[par], in four steps of two elements so that it can be spread behind the MFMAs of an interval (a wave's VALU work
    // only overlaps its own MFMAs if it is issued between them)
    auto gelu_pair = [&](auto par_tag, auto j_tag, auto p_tag, f32x2 b) __attribute__((always_inline)) {
        constexpr int par = decltype(par_tag)::value, j = decltype(j_tag)::value, fb = j >> 1, s = j & 1;
        constexpr int p = decltype(p_tag)::value;             // elements 2p, 2p+1 of the fragment = registers 8s + 2p, +1
        constexpr int e0 = 2 * p;
        // b = the biases of THIS chunk (added here) or, with LT_BIASINIT, of the chunk the accumulators serve next
        const float x0 = LT_BIASINIT ? accU[fb][8 * s + e0] : accU[fb][8 * s + e0] + b[0];
        const float x1 = LT_BIASINIT ? accU[fb][8 * s + e0 + 1] : accU[fb][8 * s + e0 + 1] + b[1];
        if constexpr (LT_ABLATE & 2) {
            g[par][j][e0] = (_Float16)x0; g[par][j][e0 + 1] = (_Float16)x1;
        } else if constexpr (LT_GELU16) {
            const f16x2_t gv = gelu_pk16(x0, x1);
            g[par][j][e0] = gv[0]; g[par][j][e0 + 1] = gv[1];
        } else {
            g[par][j][e0] = (_Float16)gelu_fast(x0); g[par][j][e0 + 1] = (_Float16)gelu_fast(x1);
        }
        accU[fb][8 * s + e0] = LT_BIASINIT ? b[0] : 0.f;
        accU[fb][8 * s + e0 + 1] = LT_BIASINIT ? b[1] : 0.f;
    };
    // bias of pair (j, p) of a chunk: features 32fb + 16s + 4hi + {2p, 2p+1} (p < 2) or + 8 + {2p-4, 2p-3}, as a float offset
    // into the chunk's 64 biases without the 4hi part
    auto bias_off = [](int j, int p) constexpr { return 32 * (j >> 1) + 16 * (j & 1) + (p < 2 ? 2 * p : 4 + 2 * p); };

    // ================================ out-projection ================================
    dma_proj(0, 0, 0, ALLP{});
    if (NT * KU > 1) dma_proj(KU > 1 ? 0 : 1, KU > 1 ? 1 : 0, 1, ALLP{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // x rows, ctx fragments, tiles 0 and 1
    static_for<NT * KU>([&](auto t_tag) __attribute__((always_inline)) {
        constexpr int t = decltype(t_tag)::value, n3 = t / KU, kt = t % KU;
        using Cur = TileDesc<K_PROJ, n3, kt>;
        using Prev = std::conditional_t<t == 0, NoTile, TileDesc<K_PROJ, (t - 1) / KU, (t - 1) % KU>>;
        // tile t+2: still out-projection, or the first up-projection tiles of chunk 0
        auto pf = [&](int s2, auto pc) __attribute__((always_inline)) {
            constexpr int t2 = t + 2;
            if constexpr (t2 < NT * KU) dma_proj(t2 / KU, t2 % KU, s2, pc);
            else dma_up(0, t2 - NT * KU, s2, pc);                 // up-projection tiles 0, 1 of chunk 0 (NT >= 2), see below
        };
        interval(Cur{}, Prev{}, VM4{}, pf, nothing, nothing1, nothing1);
    });
    drain(TileDesc<K_PROJ, NT - 1, KU - 1>{});

    // ================================ LayerNorm 1 (wave-local) -> y fragments in S ================================
    // (the epilogues use opaque copies of the lane ids: otherwise their address arithmetic is computed at kernel start
    // and carried through the tile loop in registers the loop needs)
    {
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int l31 = lane_e & 31, hi = lane_e >> 5, lane = lane_e;
        // one pass over the accumulators for both moments (mean, E[v^2]); a second pass normalises with a per-feature
        // scale and offset.  (A wave alone on its SIMD pays every VALU instruction at full price: the three-pass
        // textbook form cost 7 us per LayerNorm.)  var = E[v^2] - mean^2 in f32: the inputs are O(1) with |mean| < std.
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int c = 4 * n + gq, f0 = 32 * n + 8 * gq + 4 * hi;
                const f32x4 bv = *(const f32x4 *)(cbo + f0);
                const f16x4 xv = *(const f16x4 *)(S + (c * 32 + ((l31 + c) & 31)) * 16 + hi * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc2[n][4 * gq + e] + bv[e] + (float)xv[e];
                    acc2[n][4 * gq + e] = v;
                    s1 += v;
                    s2 = __builtin_fmaf(v, v, s2);
                }
            }
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        const float mean = s1 * (1.0f / H);
        const float var = fmaxf(s2 * (1.0f / H) - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f), nmr = -mean * rstd;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // every x read of the wave is done before S is rewritten
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 o;
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    const int f0 = 32 * n + 16 * s + 8 * hq + 4 * hi;              // registers 8s + 4hq + 0..3
                    const f32x4 gv = *(const f32x4 *)(cg1 + f0), bv = *(const f32x4 *)(cbe1 + f0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 8 * s + 4 * hq + e;
                        // g * ((v - mean) * rstd) + b  =  v * (g * rstd) + (g * (-mean * rstd) + b)
                        o[4 * hq + e] = (_Float16)__builtin_fmaf(acc2[n][r], gv[e] * rstd, __builtin_fmaf(gv[e], nmr, bv[e]));
                        acc2[n][r] = 0.f;
                    }
                }
                *(f16x8 *)(S + (2 * n + s) * 1024 + lane * 16) = o;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // y is read back by hand-issued reads of this wave
    }

    // ================================ feed-forward ================================
    auto refresh_lane_values = [&]() __attribute__((always_inline)) {
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        lane_offsets(lane_e);
        const int l31e = lane_e & 31, hie = lane_e >> 5;
        aA0 = lds_addr(ring) + off64(l31e, hie);
        aU0 = lds_addr(ring) + l31e * 256 + ((hie ^ (l31e & 15)) << 4);
        aY = lds_addr(S) + lane_e * 16;
        aB = lds_addr(cb1) + hie * 16;
    };
    refresh_lane_values();
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r)          // register r of block fb = feature 32 fb + (r & 3) + 8 (r >> 2) + 4 hi of the chunk
            accU[fb][r] = LT_BIASINIT ? cb1[32 * fb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] : 0.f;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int j = 0; j < 4; ++j) g[pp][j] = (f16x8)(_Float16)0;
    // stream order (NT tiles per stage):
    //     UP(0) | gelu(0) | UP(1) | DOWN(0)+gelu(1) | UP(2) | DOWN(1)+gelu(2) | ... | UP(NC-1) | DOWN(NC-2)+gelu(NC-1) | DOWN(NC-1)
    // the GELU of chunk c+1 is VALU filler behind the MFMAs of DOWN(c); only gelu(0) is exposed.  One set of up-
    // projection accumulators, two GELU'ed chunks (g[c & 1]).  NT >= 2 and NC >= 2.
    // ---- UP(0)  (its first two tiles were requested by the last out-projection intervals)
    static_for<NT>([&](auto j_tag) __attribute__((always_inline)) {
        constexpr int j = decltype(j_tag)::value;
        using Cur = TileDesc<K_UP, 0, j>;
        using Prev = std::conditional_t<j == 0, NoTile, TileDesc<K_UP, 0, j - 1>>;
        auto pf = [&](int s2, auto pc) __attribute__((always_inline)) {
            constexpr int j2 = j + 2;
            if constexpr (j2 < NT) dma_up(0, j2, s2, pc); else dma_up(1, j2 - NT, s2, pc);
        };
        interval(Cur{}, Prev{}, VM4{}, pf, nothing, nothing1, nothing1);
    });
    drain(TileDesc<K_UP, 0, NT - 1>{});
    static_for<4>([&](auto j_tag) __attribute__((always_inline)) {
        static_for<4>([&](auto p_tag) __attribute__((always_inline)) {
            gelu_pair(std::integral_constant<int, (LT_GROLE < 0 ? 0 : (LT_GROLE ^ 1))>{}, j_tag, p_tag, *(const f32x2 *)(cb1 + (LT_BIASINIT ? 64 : 0) + bias_off(decltype(j_tag)::value, decltype(p_tag)::value) + 4 * hi));
        });
    });

    // ---- step c: UP(c+1), then DOWN(c) with gelu(c+1) as filler.  GP = c & 1 (g buffer of chunk c).
    // FIRST: c == 0; LAST: c == NC - 2 (no UP(c+2) to request: a separate instantiation keeps the requests branch-free)
    auto step = [&](auto gp_tag, auto first_tag, auto last_tag, int c) __attribute__((always_inline)) {
        constexpr int GPT = decltype(gp_tag)::value;
        constexpr int GP = LT_GROLE < 0 ? GPT : LT_GROLE;          // buffer DOWN(c) reads
        constexpr int GW = GP ^ 1;                                  // buffer gelu(c+1) is written to
        constexpr int GE = LT_GROLE < 0 ? GP ^ 1 : GP;              // buffer the tile owed on entry (of DOWN(c-1)) reads
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
        // fixed roles: chunk c, written to g[GW] during the previous step, moves to g[GP] behind the last owed MFMA
        auto move_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) g[GP][j] = g[GW][j];
        };
        if constexpr (LT_GROLE >= 0 && FIRST) move_chunk();
        auto after_owed = [&](auto grp_tag, auto k_tag) __attribute__((always_inline)) {
            if constexpr (LT_GROLE >= 0 && decltype(grp_tag)::value == 0 && decltype(k_tag)::value == 7) move_chunk();
        };
        static_for<NT>([&](auto j_tag) __attribute__((always_inline)) {
            constexpr int j = decltype(j_tag)::value;
            using Cur = TileDesc<K_UP, 0, j>;
            // owed on entry: nothing after the exposed gelu(0), else the last tile of DOWN(c-1)
            using Entry = std::conditional_t<FIRST, NoTile, TileDesc<K_DOWN, GE, NT - 1>>;
            using Prev = std::conditional_t<j == 0, Entry, TileDesc<K_UP, 0, j - 1>>;
            auto pf = [&](int s2, auto pc) __attribute__((always_inline)) {
                constexpr int j2 = j + 2;
                if constexpr (j2 < NT) dma_up(c + 1, j2, s2, pc); else dma_down(c, j2 - NT, s2, pc);
            };
            if constexpr (j == 0 && !FIRST) interval(Cur{}, Prev{}, VM4{}, pf, after_owed, nothing1, nothing1);
            else interval(Cur{}, Prev{}, VM4{}, pf, nothing, nothing1, nothing1);
        });
        static_for<NT>([&](auto d_tag) __attribute__((always_inline)) {
            constexpr int d = decltype(d_tag)::value;
            using Cur = TileDesc<K_DOWN, GP, d>;
            using Prev = std::conditional_t<d == 0, TileDesc<K_UP, 0, NT - 1>, TileDesc<K_DOWN, GP, d - 1>>;
            auto pf = [&](int s2, auto pc) __attribute__((always_inline)) {
                constexpr int d2 = d + 2;
                if constexpr (d2 < NT) dma_down(c, d2, s2, pc);
                else if constexpr (!LAST) dma_up(c + 2, d2 - NT, s2, pc);
                else dma_down(c + 1, d2 - NT, s2, pc);        // chunk c+1 is the last one: only its DOWN tiles are left
            };
            // the 16 element pairs of gelu(c+1) behind the MFMAs of the NT intervals (lt_pair_of); their biases are read
            // by hand one group of 8 MFMA slots ahead
            f32x2 Bv[2][6];
            const unsigned aBc = aB + (unsigned)(c + 1 + LT_BIASINIT) * 256u;      // (LT_BIASINIT: one chunk past the end in the last step, never used)
            auto pre = [&](auto grp_tag) __attribute__((always_inline)) {
                constexpr int grp = decltype(grp_tag)::value;
                static_for<8>([&](auto k_tag) __attribute__((always_inline)) {
                    constexpr int k = decltype(k_tag)::value, q = lt_pair_of<NT>(d, grp * 8 + k);
                    if constexpr (q >= 0) Bv[grp][lt_pair_rank<NT>(d, grp, k)] = lds_read_b64_u<bias_off(q / 4, q % 4) * 4>(aBc);
                });
            };
            auto fence = [&](auto grp_tag) __attribute__((always_inline)) {
                constexpr int grp = decltype(grp_tag)::value;
                bias_fence_n<lt_pair_rank<NT>(d, grp, 8), grp == 0>(Bv[grp]);
            };
            auto fill = [&](auto grp_tag, auto k_tag) __attribute__((always_inline)) {
                constexpr int grp = decltype(grp_tag)::value, k = decltype(k_tag)::value, q = lt_pair_of<NT>(d, grp * 8 + k);
                if constexpr (q >= 0 && !((LT_ABLATE & 32) && q >= 10) && !(LT_ABLATE & 64))
                    gelu_pair(std::integral_constant<int, GW>{}, std::integral_constant<int, q / 4>{},
                              std::integral_constant<int, q % 4>{}, Bv[grp][lt_pair_rank<NT>(d, grp, k)]);
            };
            interval(Cur{}, Prev{}, VM4{}, pf, fill, pre, fence);
        });
    };
    // NC is even and >= 4: steps c = 0 .. NC-2 (GP = c & 1), the last one with GP = 0
    step(H0{}, std::true_type{}, std::false_type{}, 0);
#if LT_GROLE >= 0
    // fixed roles: one copy of the step in the loop
    for (int c = 1; c + 2 < NC; ++c) step(H1{}, std::false_type{}, std::false_type{}, c);
#else
    for (int c = 1; c + 3 < NC; c += 2) {
        step(H1{}, std::false_type{}, std::false_type{}, c);
        step(H0{}, std::false_type{}, std::false_type{}, c + 1);
    }
    step(H1{}, std::false_type{}, std::false_type{}, NC - 3);
#endif
    step(H0{}, std::false_type{}, std::true_type{}, NC - 2);
    // ---- DOWN(NC-1)
    auto last_down = [&](auto gp_tag) __attribute__((always_inline)) {
        constexpr int GP = LT_GROLE < 0 ? decltype(gp_tag)::value : LT_GROLE;
        constexpr int GE = LT_GROLE < 0 ? GP ^ 1 : GP;
        auto after_owed = [&](auto grp_tag, auto k_tag) __attribute__((always_inline)) {
            if constexpr (LT_GROLE >= 0 && decltype(grp_tag)::value == 0 && decltype(k_tag)::value == 7) {
#pragma unroll
                for (int j = 0; j < 4; ++j) g[GP][j] = g[GP ^ 1][j];
            }
        };
        static_for<NT>([&](auto d_tag) __attribute__((always_inline)) {
            constexpr int d = decltype(d_tag)::value;
            using Cur = TileDesc<K_DOWN, GP, d>;
            using Prev = std::conditional_t<d == 0, TileDesc<K_DOWN, GE, NT - 1>, TileDesc<K_DOWN, GP, d - 1>>;
            auto pf = [&](int s2, auto pc) __attribute__((always_inline)) {
                constexpr int d2 = d + 2;
                if constexpr (d2 < NT) dma_down(NC - 1, d2, s2, pc);
            };
            constexpr int vm = d + 2 < NT ? 4 : 0;
            if constexpr (d == 0) interval(Cur{}, Prev{}, std::integral_constant<int, vm>{}, pf, after_owed, nothing1, nothing1);
            else interval(Cur{}, Prev{}, std::integral_constant<int, vm>{}, pf, nothing, nothing1, nothing1);
        });
        drain(TileDesc<K_DOWN, GP, NT - 1>{});
    };
    last_down(H1{});                                          // NC - 1 is odd

    // ================================ LayerNorm 2 (wave-local) -> rows in S -> HBM ================================
    TL_STAMP(tl++);
    {
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int hi = lane_e >> 5, lane = lane_e;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x8 yv = *(const f16x8 *)(S + (2 * n + s) * 1024 + lane * 16);
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    const f32x4 bv = *(const f32x4 *)(cb2 + 32 * n + 16 * s + 8 * hq + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 8 * s + 4 * hq + e;
                        const float v = acc2[n][r] + bv[e] + (float)yv[4 * hq + e];
                        acc2[n][r] = v;
                        s1 += v;
                        s2 = __builtin_fmaf(v, v, s2);
                    }
                }
            }
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        const float mean = s1 * (1.0f / H);
        const float var = fmaxf(s2 * (1.0f / H) - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f), nmr = -mean * rstd;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // fragment (q, lane) -> position (lane + 2q) & 63 of row q: the row-major read below is conflict-free
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int q = 2 * n + s;
                f16x8 o;
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    const int f0 = 32 * n + 16 * s + 8 * hq + 4 * hi;
                    const f32x4 gv = *(const f32x4 *)(cg2 + f0), bv = *(const f32x4 *)(cbe2 + f0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o[4 * hq + e] = (_Float16)__builtin_fmaf(acc2[n][8 * s + 4 * hq + e], gv[e] * rstd, __builtin_fmaf(gv[e], nmr, bv[e]));
                }
                *(f16x8 *)(S + q * 1024 + ((lane + 2 * q) & 63) * 16) = o;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TL_STAMP(tl++);
        // 16-byte unit (token, 8-feature chunk c8) of the output = bytes h2*8.. of the fragments (q, token) and
        // (q, token + 32), q = c8 >> 1, h2 = c8 & 1
        half_t *ow = a.out + (size_t)tok_w * H;
#pragma unroll
        for (int st = 0; st < H / 16; ++st) {
            const int v = st * 64 + lane, tok = v / (H / 8), c8 = v - tok * (H / 8), q = c8 >> 1, h2 = c8 & 1;
            const char *row = S + q * 1024 + h2 * 8;
            const f16x4 lo = *(const f16x4 *)(row + ((tok + 2 * q) & 63) * 16);
            const f16x4 hi4 = *(const f16x4 *)(row + ((tok + 32 + 2 * q) & 63) * 16);
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi4[e]; }
            *(f16x8 *)(ow + (size_t)tok * H + c8 * 8) = o;
        }
    }
#ifdef BERT_HIP_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TL_STAMP(tl++);
#endif
}

bool layer_tail_supported(const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2) {
    const int H = W1.K, I = W1.N;
    if (Wo.type != GW_F16 || W1.type != GW_F16 || W2.type != GW_F16 || !W1.w16p || !W2.w16p) return false;
    if (Wo.N != H || Wo.K != H || W2.N != H || W2.K != I) return false;
    if (H % 128 != 0 || H < 256 || H > 384 || I % 128 != 0 || I < 256) return false;     // an even number >= 4 of 64-feature chunks
    const size_t lds = (size_t)4 * 64 * H + 3 * LT_TILE + (size_t)(6 * H + I + 64) * sizeof(float);
    return lds <= 160 * 1024;
}

void launch_layer_tail(const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2, const half_t *ctx, const half_t *x,
                       const float *bo, const float *g1, const float *be1, const float *b1, const float *b2,
                       const float *g2, const float *be2, half_t *out, int M_pad, hipStream_t stream) {
    TailArgs a;
    a.ctx = ctx; a.x = x; a.wo = Wo.w16; a.w1p = W1.w16p; a.w2p = W2.w16p;
    a.bo = bo; a.g1 = g1; a.be1 = be1; a.b1 = b1; a.b2 = b2; a.g2 = g2; a.be2 = be2; a.out = out; a.I = W1.N;
    const int H = W1.K, NT = H / 128;
    const size_t lds = (size_t)4 * 64 * H + 3 * LT_TILE + (size_t)(6 * H + a.I + 64) * sizeof(float);
    static bool configured[4][MAX_HIP_DEVICES] = {};
    auto go = [&](auto kernel) __attribute__((always_inline)) {
        if (first_launch_on_device(configured[NT])) {
            (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        hipLaunchKernelGGL(kernel, dim3(M_pad / 128), dim3(256), lds, stream, a);
        TL_DUMP(M_pad >= 128 * 256, 200);
    };
    switch (NT) {
        case 2: go(layer_tail_kernel<2>); break;
        default: go(layer_tail_kernel<3>); break;
    }
}

}  // namespace bert_hip
