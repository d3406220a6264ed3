// gemm256.hip — the large-tile weight mat-mul of the H > 384 models (bert-base / mpnet dimensions, BASELINE configs 3, 4):
//   C[t][n] = epilogue( sum_k A[t][k] * W[n][k] + bias[n] )        A: f16 activations, W: f16 image, or q4_0 / q4_1 planes
//   (BERT_HIP_Q4=fused: the matrix stays 4-bit in HBM / L2 and its blocks are dequantised into the weight tile's LDS image by
//   the tile load itself — same image, same MFMA sequence, same bits as the f16 form on the expanded matrix)
// Same operation and epilogues as gemm.hip (reference bert.cpp:822-839, :859-865, :878-882, :885-891); what changes is the
// shape of the work, chosen for the two things a power-limited MI355X pays for besides the MFMAs themselves — LDS
// fragment reads and L2 -> LDS tile traffic:
//   * 256 tokens x 256 features per workgroup of 8 waves (two per SIMD: while one wave waits for a tile or a fragment
//     the other one has the matrix pipe), a wave owns 128 features x 64 tokens = 4 x 2 accumulator blocks, so a k-step
//     is 6 fragment reads for 8 MFMAs (gemm.hip: 4 for 4) and a 64-deep reduction tile moves 64 KiB through the LDS-DMA
//     for 256 MFMAs (gemm.hip: 32 KiB for 64);
//   * v_mfma_f32_32x32x16_f16 with the WEIGHT tile as the A operand and the ACTIVATION tile as the B operand: a lane's
//     accumulator column is one token, its registers are 4-feature runs (bias / GELU / residual per lane, f32);
//   * tiles go HBM/L2 -> LDS by global_load_lds_dwordx4, double buffered (2 x 64 KiB), one barrier per reduction tile,
//     the 16-byte chunk swizzle on the SOURCE address (the DMA writes lane-linearly) and again on the fragment reads; the
//     fragments of a k-step are read (hand-issued ds_read_b128, counted waits) under the MFMAs of the one before, and the
//     last k-step of a reduction tile runs behind the next tile's barrier, under that tile's first reads;
//   * ONE PERSISTENT WORKGROUP PER CU walks its share of the output tiles (each XCD a contiguous range, all feature tiles
//     of a token tile back to back: an activation tile comes from HBM once, not once per XCD) and the stream of reduction
//     tiles runs across output tiles without a gap — no workgroup launch, prologue latency or drained pipeline per tile;
//   * the accumulators of a tile START from bias (+ residual): those loads are issued between the two phases of the finished
//     tile's epilogue (the bias vectors straight into the dead accumulator tuples) and land under its stores, so a tile
//     boundary has no load round trip of its own; epilogue: GELU in f32 -> packed f16 in the accumulator layout, one rounding,
//     then a wave passes its own 64 tokens x 128 features through a private 4 KiB staging area in four rounds, so every global
//     store is 16 bytes of a full 128-byte row segment; no barrier, no use of the tile buffers.
// Measured (bert-base, 512 x 512 tokens, per reduction tile 1.45 us = 88 % of the matrix rate the board sustains at its
// power limit; hipBLASLt's 256x256x64 kernel on the same shapes: QKV 836 us, this kernel 960): what is left is the epilogue
// — 128 store instructions per tile at 31-52 cycles each per CU (tools/ubench/store_issue.hip) with no MFMA beside them.
#include "tile_stream.h"

#include <type_traits>

namespace bert_hip {

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define G2_GLOBAL(p) ((const __attribute__((address_space(1))) void *)(p))
#define G2_LDS(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 64;
constexpr int G2_TILE = 256 * 128;               // 256 rows x 64 halfs
constexpr int G2_STAGE = 2 * G2_TILE;            // activation tile + weight tile

struct Gemm256Args {
    const half_t *A;        // [M_pad][K], M_pad % 256 == 0
    const half_t *w16;      // [N_pad][K], N % 256 == 0
    const float *bias;      // [N]
    const half_t *resid;    // [M_pad][N] or null
    half_t *C;              // [M_pad][N]
    const uint4 *qs;        // q4 forms: nibble plane / scale plane of W (kernels.h GemmWeight), w16 unused
    const void *sc;
    int N, K, n_tiles_n, n_tiles;
    int n_groups;           // feature-tile groups an XCD pair / quad shares the walk with (1: every XCD walks all feature tiles)
    // ---- LayerNorm folded into its neighbours (LN != 0, kernels.h GemmLnFold): the residual mat-muls write the UN-normalised sum u
    // and per-row partial statistics, the mat-muls that consume LayerNorm(u) read u itself
    const float4 *rows_in;  // LN_IN: per row of A {rstd, -mean rstd, -mean, std} of the LayerNorm that produced A's consumer input
    const half_t *waug;     // LN_IN: [N][16] f16, the weight side of the statistics k-step (columns 0..5: s_hi s_lo s_hi c_hi c_lo c_hi)
    const float4 *rows_res; // LN_RES: per row of resid {rstd, -mean rstd, ..}: the residual is LayerNorm(resid) rebuilt per element
    const unsigned *gb;     // LN_RES: per feature (f16 gamma | f16 (beta + bias) << 16) instead of `bias`
    float2 *stats;          // LN_STATS: [M_pad][stats_p] (sum, sum of squares) of the rounded results per row and 128-feature half
    int stats_p;
};

// LN flags of the kernel (bert_hip::GemmLnFold's): what this launch does for the LayerNorms around it
constexpr int LN_IN = 1, LN_RES = 2, LN_STATS = 4;

}  // namespace

// G2_ABLATE (tuning builds only, results are wrong): bit 0 no global stores in the epilogue, bit 1 no epilogue at all,
// bit 2 no residual loads, bit 3 no LDS staging (the stores write whatever the staging area holds) — what a component costs is the time its removal saves (tools/variant.sh)
#ifndef G2_GELU_RUNS
#define G2_GELU_RUNS 1      // GELU runs of 4 values the scheduler may interleave in the epilogue
#endif
#ifndef G2_ABLATE
#define G2_ABLATE 0
#endif

// G2_DMA_PLAN: how many of a wave's 8 LDS-DMA pieces of the next reduction tile are issued in each of the four 8-MFMA steps
// of a reduction tile (deferred last k-step, k-steps 0, 1, 2), as a 4-digit number.  The CU's vector-memory path takes ~12
// cycles per piece: all 64 pieces of a tile offered within one step's 256 cycles queue up and the waves wait at the
// issue (tools/ubench/store_beside_mfma.hip: +590 cycles per tile bunched, +84 spread); too late a piece misses the barrier.
#ifndef G2_DMA_PLAN
#define G2_DMA_PLAN 4400
#endif
constexpr int g2_plan_count(int step) { return step == 0 ? G2_DMA_PLAN / 1000 : step == 1 ? G2_DMA_PLAN / 100 % 10 : step == 2 ? G2_DMA_PLAN / 10 % 10 : G2_DMA_PLAN % 10; }
static_assert(g2_plan_count(0) + g2_plan_count(1) + g2_plan_count(2) + g2_plan_count(3) == 8, "a wave issues 8 pieces per reduction tile");
// the piece issued behind MFMA m (0..7) of `step`, or -1: a step's pieces are spread evenly over its MFMAs
constexpr int g2_piece_at(int step, int m) {
    int first = 0;
    for (int s = 0; s < step; ++s) first += g2_plan_count(s);
    const int c = g2_plan_count(step);
    for (int k = 0; k < c; ++k)
        if ((k * 8) / c == m) return first + k;
    return -1;
}

// ---- hand-issued fragment reads (the compiler does not track them: every wait names the registers it releases)
template <int OFF>
__device__ __forceinline__ f16x8 g2_read_b128(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
struct G2Frag {
    f16x8 a[4], b[2];                                  // weight rows (4 x 32 features), activation rows (2 x 32 tokens) of one k-step
};
// everything but the newest six reads (the next k-step's) has landed
__device__ __forceinline__ void g2_wait6(G2Frag &f) {
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]) : : "memory");
}
__device__ __forceinline__ void g2_wait0(G2Frag &f) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]) : : "memory");
}
// reduction-tile barrier: this wave's pieces of the tile have landed, its reads of the previous one have returned (the
// fragments of that tile's last k-step are named: their MFMAs run after the barrier, under the first reads of the new tile)
__device__ __forceinline__ void g2_tile_barrier(G2Frag &f) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier"
                 : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]) : : "memory");
}

// A PERSISTENT workgroup per CU walks its share of the output tiles (each XCD a contiguous range: all feature tiles of a
// token tile back to back, so an activation tile comes from HBM once); the stream of reduction tiles runs across output
// tiles without a gap: the first reduction tile of the next output tile is requested during the last one of the current,
// and the epilogue of a finished output tile runs after the NEXT tile's first barrier — its global stores are then
// retired under a whole reduction tile of MFMAs instead of in front of a barrier.
// WT != GW_F16 (q4_0 / q4_1 planes): the weight half of a reduction tile is 512 blocks of 32 weights, one per thread.  A
// thread REQUESTS its block of the tile after next (16 bytes of nibbles + the scale: two plain global loads, coalesced —
// a wave's 64 blocks are 1 KiB + 128 / 256 B of the tile-contiguous planes) in the last k-step of a reduction tile and EXPANDS
// it one tile later, a 16-byte chunk at a time behind MFMAs of the first two k-steps (v_perm_b32 builds (1024 + q) half pairs,
// packed f16 math applies (q - 8) d or q d + m: ~15 VALU + one ds_write_b128 per chunk), into the stage the f16 form fills by
// LDS-DMA — the activation half still arrives that way.  At an output-tile boundary the pending block is expanded in one go in
// front of the finished tile's epilogue, so the raw registers are dead while the epilogue needs every register.
template <int EPI, int WT, int LN = 0>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(Gemm256Args p) {
    static_assert(LN == 0 || WT == GW_F16, "LayerNorm folding runs on f16 images");
    static_assert(!(LN & LN_IN) || EPI != EPI_BIAS_RESID, "a mat-mul either consumes a folded LayerNorm or produces one's input");
    static_assert(!(LN & (LN_RES | LN_STATS)) || EPI == EPI_BIAS_RESID, "");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool Q4 = WT != GW_F16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wf = wave & 1, wq = wave >> 1;         // feature half (128) / token quarter (64) of the tile
    const int l31 = lane & 31, hi = lane >> 5;
    const int K = p.K, nk = K / G2_BK;

    // ---- this workgroup's output tiles: t_first, t_first + S, ... below t_end, in the XCD's own numbering.  An XCD walks a
    // contiguous range of token tiles and, of each, the feature tiles n_begin .. n_begin + cnt_n back to back (an activation
    // tile comes from HBM once per XCD that needs it).  n_groups = 1: cnt_n = all of them, the ranges are cut at tile
    // granularity.  n_groups = G > 1 (weight matrices that do not fit an XCD's 4 MiB L2 beside the activation tiles in flight:
    // the up-projection of the H = 768 models, 4.5 MiB): XCD x takes feature group x % G — its slice of W stays in its L2 —
    // and token range x / G of 8 / G; the activations are read G times (from the memory-side cache after the first).
    const int xcd = blockIdx.x & 7, S = gridDim.x >> 3;
    const int G = p.n_groups, cnt_n = p.n_tiles_n / G, n_begin = (xcd % G) * cnt_n;
    int t_begin, t_end;
    if (G == 1) {
        const int q8 = p.n_tiles >> 3, r8 = p.n_tiles & 7;
        t_begin = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        t_end = t_begin + q8 + (xcd < r8 ? 1 : 0);
    } else {
        const int R = 8 / G, r = xcd / G, tm = p.n_tiles / p.n_tiles_n, q = tm / R, rem = tm % R;
        const int m_begin = r < rem ? r * (q + 1) : rem * (q + 1) + (r - rem) * q;
        t_begin = m_begin * cnt_n;
        t_end = t_begin + (q + (r < rem ? 1 : 0)) * cnt_n;
    }
    int tile = t_begin + (int)(blockIdx.x >> 3);
    if (tile >= t_end) return;

    // ---- LDS-DMA: a reduction tile is 2 x 32 pieces of 1 KiB (8 rows each), 4 + 4 per wave; source offsets (elements)
    unsigned doff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        doff[i] = (unsigned)(r * K + (((lane & 7) ^ ((r >> 1) & 7)) << 3));
    }
    auto dma_piece = [&](const half_t *src, char *stage_base, auto i_tag) __attribute__((always_inline)) {
        constexpr int i = decltype(i_tag)::value;       // 0..3: activation pieces, 4..7: weight pieces
        if constexpr (Q4 && i >= 4) return;             // (q4: the weight half is expanded from blocks, below)
        else __builtin_amdgcn_global_load_lds(G2_GLOBAL(src + doff[i & 3]), G2_LDS(stage_base + (i >> 2) * G2_TILE + (wave * 4 + (i & 3)) * 1024), 16, 0, 0);
    };

    // ---- fragment addresses of the four k-steps of a reduction tile (stage 0; the other stage is address ^ 64 KiB)
    unsigned aW[4], aA[4];
    {
        const int s = (l31 >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const unsigned swz = (unsigned)((((kk * 2 + hi) ^ s) << 4));
            aA[kk] = (unsigned)(size_t)smem + (unsigned)((wq * 64 + l31) * 128) + swz;      // (LDS addresses are 32-bit)
            aW[kk] = (unsigned)(size_t)smem + (unsigned)(G2_TILE + (wf * 128 + l31) * 128) + swz;
        }
    }
    auto read_frag = [&](G2Frag &f, auto kk_tag) __attribute__((always_inline)) {
        constexpr int kk = decltype(kk_tag)::value;
        f.a[0] = g2_read_b128<0>(aW[kk]); f.a[1] = g2_read_b128<4096>(aW[kk]);
        f.a[2] = g2_read_b128<8192>(aW[kk]); f.a[3] = g2_read_b128<12288>(aW[kk]);
        f.b[0] = g2_read_b128<0>(aA[kk]); f.b[1] = g2_read_b128<4096>(aA[kk]);
    };

    // ---- q4: this thread's block of a weight tile.  A wave owns 32 rows x 2 blocks; its lanes are dealt so that the eight lanes
    // of a ds_write_b128 group (8 x 8 contiguous lanes) hit eight different 16-byte columns of the swizzled tile:
    // lane = 8 g + j -> row 2 (j & 3) + (g & 1) + 8 (g >> 1), block j >> 2.
    const int q_row = wave * 32 + 2 * (lane & 3) + ((lane >> 3) & 1) + 8 * (lane >> 4), q_kb = (lane >> 2) & 1;
    unsigned q_dst = (unsigned)(size_t)smem + (unsigned)(G2_TILE + q_row * 128 + (((q_kb * 4) ^ ((q_row >> 1) & 7)) << 4));   // chunk c: ^ (c << 4)
    int q_lane_idx = (q_row & 31) * 2 + q_kb;   // within the wave's 64 consecutive blocks of the planes
    // The request is hand-issued (two asm global loads, scalar base + lane offset): a compiler-visible load is retired by the
    // compiler with a vmcnt that also covers the activation pieces issued behind it — a wait for LDS-DMA in front of the first
    // chunk.  Every path from a request to its expansion crosses a reduction-tile barrier (vmcnt(0)), which names these registers.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 raw_q = {};
    unsigned raw_sc = 0;
    unsigned q_voff = (unsigned)q_lane_idx;
    auto q4_request = [&](int n0_, int kt_) __attribute__((always_inline)) {
        if constexpr (Q4) {
            // planes: 128-row tiles of 256 blocks, tile (n / 128, kt) at ((n / 128) nk + kt) * 256; this wave: rows 32 (wave & 3) ..
            const size_t wave_base = ((size_t)((n0_ >> 7) + (wave >> 2)) * nk + kt_) * 256 + (size_t)(wave & 3) * 64;
            const uint4 *qb = p.qs + wave_base;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(raw_q) : "v"(q_voff << 4), "s"(qb) : "memory");
            if constexpr (WT == GW_Q4_0) {
                const unsigned short *sb = (const unsigned short *)p.sc + wave_base;
                asm volatile("global_load_ushort %0, %1, %2" : "=v"(raw_sc) : "v"(q_voff << 1), "s"(sb) : "memory");
            } else {
                const unsigned *sb = (const unsigned *)p.sc + wave_base;
                asm volatile("global_load_dword %0, %1, %2" : "=v"(raw_sc) : "v"(q_voff << 2), "s"(sb) : "memory");
            }
        }
    };
    // the requested block has landed (behind a vmcnt(0)): hand it to the compiler as the expansion's input
    auto q4_raw = [&]() __attribute__((always_inline)) {
        RawBlock r;
        r.q = make_uint4(raw_q[0], raw_q[1], raw_q[2], raw_q[3]);
        r.sc = raw_sc;
        return r;
    };
    auto tile_barrier = [&](G2Frag &f) __attribute__((always_inline)) {
        if constexpr (Q4)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier"
                         : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(raw_q), "+v"(raw_sc) : : "memory");
        else g2_tile_barrier(f);
    };
    auto q4_expand_one = [&](unsigned stage_xor, auto c_tag) __attribute__((always_inline)) {
        if constexpr (Q4) {
            constexpr int c = decltype(c_tag)::value;
            const uint4 v = q4_expand_chunk<WT>(q4_raw(), c);
            const unsigned dst = (q_dst ^ stage_xor) ^ (unsigned)(c << 4);
            const u32x4 vv = {v.x, v.y, v.z, v.w};
            asm volatile("ds_write_b128 %0, %1" : : "v"(dst), "v"(vv) : "memory");
        }
    };
    auto q4_expand_all = [&](unsigned stage_xor) __attribute__((always_inline)) {
        static_for<4>([&](auto c) __attribute__((always_inline)) { q4_expand_one(stage_xor, c); });
    };

    f32x16 acc[4][2];                                 // [feature block][token block]
    // the 8 MFMAs of a k-step with `fill(m)` pinned behind MFMA m
    auto mfma_step_with = [&](const G2Frag &f, auto fill) __attribute__((always_inline)) {
        static_for<8>([&](auto m_tag) __attribute__((always_inline)) {
            constexpr int m = decltype(m_tag)::value, i = m >> 1, j = m & 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
            fill(m_tag);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto mfma_step = [&](const G2Frag &f) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
    };

    // ---- the accumulators of an output tile START from bias (+ residual): what the round-2 form added in its epilogue —
    // sixteen bias vectors and, for the residual form, 32 scattered 8-byte loads per lane, fetched in six dependent round trips
    // per tile with nothing else to do (14-25 us of a 35 us tile) — is requested for the NEXT tile between the two phases of
    // the finished tile's epilogue and lands under its stores.  (The sum starts from the bias instead of ending with it:
    // f32, same tolerance; the tests compare against float64.)
    // The bias vectors are loaded straight INTO the accumulator tuples of token block 0 (dead between the two phases), token
    // block 1 copies them when they have landed: through registers of their own they would be alive beside all 128
    // accumulators and both fragment sets at the tile's first MFMAs (the compiler feeds them in as C operands) — sixteen
    // spilled behind a vmcnt(0).
    // The residual comes in in the accumulator layout (a lane's 8-byte runs of its token's row: 32 loads per lane and tile, 32
    // lines per instruction — 8-9 us of a 35 us tile go into their issue wherever they are placed; fetching the tile in
    // 16-byte row segments and turning it through the staging area was tried: the register allocator spills all sixteen
    // vectors in front of the stores, and landing them by LDS-DMA has to wait behind the stores round by round).
    f16x4 rv[EPI == EPI_BIAS_RESID ? 4 : 1][2][4];
    // LN_IN: the statistics of this lane's two token rows (rstd kept for the epilogue), the weight side of the statistics k-step;
    // LN_RES: (rstd, -mean rstd) of the two residual rows
    [[maybe_unused]] float row_rstd[2] = {1.f, 1.f};
    [[maybe_unused]] f32x4 row_in[2];
    [[maybe_unused]] f32x4 row_res[2];
    auto init_loads = [&](int im0, int in0) __attribute__((always_inline)) {
        int l31 = lane & 31, hi = lane >> 5;
        asm volatile("" : "+v"(l31), "+v"(hi));
        if constexpr (LN & LN_IN) {
            // (no bias: it sits in the statistics k-step's c column.  The four weight-side fragments go into the dead accumulator
            // tuples like the bias vectors of the plain form)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 wv = *(const f32x4 *)(p.waug + ((size_t)in0 + wf * 128 + i * 32 + l31) * 16 + 8 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][0][e] = wv[e];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) row_in[j] = *(const f32x4 *)(p.rows_in + (size_t)im0 + wq * 64 + j * 32 + l31);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // (LN_RES: the same place holds the packed (gamma, beta + bias) pairs of the features)
                    const f32x4 b = (LN & LN_RES) ? *(const f32x4 *)(p.gb + in0 + wf * 128 + i * 32 + 8 * g + 4 * hi)
                                                  : *(const f32x4 *)(p.bias + in0 + wf * 128 + i * 32 + 8 * g + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][0][4 * g + e] = b[e];
                }
        }
        if (EPI == EPI_BIAS_RESID && !(G2_ABLATE & 4)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const half_t *rrow = p.resid + ((size_t)im0 + wq * 64 + j * 32 + l31) * p.N + in0 + wf * 128 + 4 * hi;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) rv[i][j][g] = *(const f16x4 *)(rrow + i * 32 + 8 * g);
                if constexpr (LN & LN_RES) row_res[j] = *(const f32x4 *)(p.rows_res + (size_t)im0 + wq * 64 + j * 32 + l31);
            }
        }
    };
    auto init_acc = [&]() __attribute__((always_inline)) {
        if constexpr (LN & LN_IN) {
            // the statistics k-step: acc = sum_k W'[n][k] u[t][k] will get  - mean_t s[n] + std_t c[n]  from ONE MFMA per block
            // (hi / lo f16 pairs on both sides: 2^-21 relative), and the epilogue multiplies by rstd_t:
            //     rstd (W' u - mean s) + c  =  W (gamma (u - mean) rstd + beta) + bias
            int hi = lane >> 5;
            asm volatile("" : "+v"(hi));
            f16x8 wa[4], ta[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 wv;
#pragma unroll
                for (int e = 0; e < 4; ++e) wv[e] = acc[i][0][e];
                wa[i] = __builtin_bit_cast(f16x8, wv);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                row_rstd[j] = row_in[j][0];
                const float nm = row_in[j][2], sd = row_in[j][3];
                const _Float16 nm_h = (_Float16)nm, sd_h = (_Float16)sd;
                const _Float16 nm_l = (_Float16)(nm - (float)nm_h), sd_l = (_Float16)(sd - (float)sd_h), z = (_Float16)0.f;
                const f16x8 t = {nm_h, nm_h, nm_l, sd_h, sd_h, sd_l, z, z};
                ta[j] = hi ? (f16x8)z : t;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[i], ta[j], (f32x16)0.f, 0, 0, 0);
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float b = acc[i][0][4 * g + e];
                    if constexpr ((LN & LN_RES) != 0) {
                        // residual = LayerNorm(resid) rebuilt per element: gamma ((r - mean) rstd) + beta, + bias
                        const unsigned w = __builtin_bit_cast(unsigned, b);
                        const float gm = (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
                        const float bb = (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
                        acc[i][1][4 * g + e] = __builtin_fmaf(gm, __builtin_fmaf((float)rv[i][1][g][e], row_res[1][0], row_res[1][1]), bb);
                        acc[i][0][4 * g + e] = __builtin_fmaf(gm, __builtin_fmaf((float)rv[i][0][g][e], row_res[0][0], row_res[0][1]), bb);
                    } else if (EPI == EPI_BIAS_RESID && !(G2_ABLATE & 4)) {
                        acc[i][1][4 * g + e] = b + (float)rv[i][1][g][e];
                        acc[i][0][4 * g + e] = b + (float)rv[i][0][g][e];
                    } else {
                        acc[i][1][4 * g + e] = b;
                    }
                }
    };

    // ---- epilogue of the output tile at (em0, en0): GELU where asked for, one rounding (bias and residual are inside the
    // accumulators already), then through a wave-private 4 KiB staging area in four rounds of [32 tokens][64 features] so
    // that every global store is 16 bytes of a full 128-byte row segment.  No barrier: a wave stages and stores its own
    // 64 tokens x 128 features.  `next`: the tile whose initial values are requested between the phases.
    char *const stg = smem + 2 * G2_STAGE + wave * 4096;
    auto epilogue = [&](int em0, int en0, bool next, int nm0, int nn0) __attribute__((always_inline)) {
        // (opaque copies: every address below is computed here, not hoisted out of the tile loop into registers that
        // the accumulators need)
        int l31 = lane & 31, hi = lane >> 5, lane_e = lane;
        asm volatile("" : "+v"(l31), "+v"(hi), "+v"(lane_e));
        if (G2_ABLATE & 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" : : "v"(acc[i][j]));
            init_loads(next ? nm0 : em0, next ? nn0 : en0);
            return;
        }
        // phase 1, registers only.  The rounded tile is gathered into FOUR 16-register tuples (round (ip, j) = the two
        // accumulator blocks 2 ip, 2 ip + 1 of token block j, converted in place into the first one's registers): left to the
        // allocator the packed halves stay scattered over all eight accumulator tuples, one register in two, and the 64
        // registers that are free hold no 4-register run for the next tile's bias vectors — sixteen of them were spilled.
        typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
        u32x16 o16[4];
        [[maybe_unused]] float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};      // LN_STATS: this lane's (sum, sum of squares) per token block
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x16 &a = acc[2 * ip + ii][j];
                        const float rs = (LN & LN_IN) ? row_rstd[j] : 1.f;      // (LN_IN: the row's 1 / std, see init_acc)
                        f16x4 h;
                        if (EPI == EPI_BIAS_GELU) {
                            // packed f16, as layer_tail.hip evaluates it (the reference reads the GELU from an f16 table)
                            const f16x2_t g0 = (LN & LN_IN) ? gelu_pk16(a[4 * g] * rs, a[4 * g + 1] * rs) : gelu_pk16(a[4 * g], a[4 * g + 1]);
                            const f16x2_t g1 = (LN & LN_IN) ? gelu_pk16(a[4 * g + 2] * rs, a[4 * g + 3] * rs) : gelu_pk16(a[4 * g + 2], a[4 * g + 3]);
                            h[0] = g0[0]; h[1] = g0[1]; h[2] = g1[0]; h[3] = g1[1];
                            if (g % G2_GELU_RUNS == G2_GELU_RUNS - 1) __builtin_amdgcn_sched_barrier(0);   // the GELU temporaries of more runs at a time would spill
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) h[e] = (LN & LN_IN) ? (_Float16)rounded_f32(a[4 * g + e] * rs) : (_Float16)a[4 * g + e];
                        }
                        if constexpr ((LN & LN_STATS) != 0) {
                            // (of the ROUNDED values: what the consumers of u read)
#pragma unroll
                            for (int e = 0; e < 4; ++e) { const float hv = (float)h[e]; st1[j] += hv; st2[j] = __builtin_fmaf(hv, hv, st2[j]); }
                        }
                        const uint2 hb = __builtin_bit_cast(uint2, h);
                        o16[ip * 2 + j][(ii * 4 + g) * 2] = hb.x;
                        o16[ip * 2 + j][(ii * 4 + g) * 2 + 1] = hb.y;
                    }
        if constexpr ((LN & LN_STATS) != 0) {
            // the two lane halves hold the two 4-feature runs of every 8: one partial per (row, 128-feature half of the tile)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                st1[j] = xor32_sum(st1[j]);
                st2[j] = xor32_sum(st2[j]);
            }
            if (hi == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    p.stats[((size_t)em0 + wq * 64 + j * 32 + l31) * p.stats_p + (en0 / G2_BN) * 2 + wf] = make_float2(st1[j], st2[j]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next tile's initial values: every load of the tile boundary is issued here, in front of the first store (vmcnt
        // retires in issue order: a load behind a store would wait for that store's acknowledgement) and BEHIND phase 1 (the
        // tile coordinates are handed over through an asm that reads phase 1's last results: hoisted above it, as the
        // compiler did with four of the loads, they land while all 128 accumulators are still alive and get spilled).  After
        // the last tile the loads are repeated for the tile itself: cheaper than a branch around them.
        {
            int im0 = next ? nm0 : em0, in0 = next ? nn0 : en0;
            asm volatile("" : "+s"(im0), "+s"(in0) : "v"(o16[0][15]), "v"(o16[1][15]), "v"(o16[2][15]), "v"(o16[3][15]), "v"(o16[0][0]), "v"(o16[1][0]), "v"(o16[2][0]), "v"(o16[3][0]) : "memory");
            init_loads(im0, in0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_sched_barrier(0);
        // phase 2, no loads: four rounds of [32 tokens][64 features] through the staging area
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint2 hb = {o16[ip * 2 + j][(ii * 4 + g) * 2], o16[ip * 2 + j][(ii * 4 + g) * 2 + 1]};
                        if (G2_ABLATE & 8) asm volatile("" : : "v"(hb.x), "v"(hb.y));
                        else *(uint2 *)(stg + l31 * 128 + (((ii * 4 + g) ^ (l31 & 7)) << 4) + hi * 8) = hb;
                    }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                half_t *crow = (G2_ABLATE & 16) ? p.C + ((size_t)blockIdx.x * 256 + wq * 64 + j * 32) * p.N + wf * 128 + ip * 64   // (every tile of a workgroup to one place)
                                                : p.C + ((size_t)em0 + wq * 64 + j * 32) * p.N + en0 + wf * 128 + ip * 64;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + (lane_e >> 3), ch = lane_e & 7;
                    const uint4 v = *(const uint4 *)(stg + row * 128 + ((ch ^ (row & 7)) << 4));
                    if (G2_ABLATE & 1) asm volatile("" : : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                    else *(uint4 *)(crow + (size_t)row * p.N + ch * 8) = v;
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
    };

    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>;

    // (a one-time start delay that spreads the workgroups' phases over a tile period was tried: no gain — the cost of an
    // epilogue is its CU's own store issue, 31 B/cycle/CU into L2 and 18 when the whole chip streams to HBM,
    // tools/ubench/store_issue.hip — not the other CUs' bursts)
    int m0 = (tile / cnt_n) * G2_BM, n0 = (n_begin + tile % cnt_n) * G2_BN;
    {   // the first reduction tile of the first output tile
        const half_t *a = p.A + (size_t)m0 * K, *w = p.w16 + (size_t)n0 * K;
        dma_piece(a, smem, I0{}); dma_piece(a, smem, I1{}); dma_piece(a, smem, I2{}); dma_piece(a, smem, I3{});
        dma_piece(w, smem, I4{}); dma_piece(w, smem, I5{}); dma_piece(w, smem, I6{}); dma_piece(w, smem, I7{});
        if constexpr (Q4) {
            q4_request(n0, 0);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw_q), "+v"(raw_sc) : : "memory");
            q4_expand_all(0u);
            q4_request(n0, 1);
        }      // (raw: always the tile after the one being multiplied)
    }
    init_loads(m0, n0);                                // (the first tile's initial values: one exposed round trip per launch)
    init_acc();
    G2Frag f0, f1;
    bool have_prev = false;
    int pm0 = 0, pn0 = 0;
    // k-steps 0..2 of a reduction tile whose first fragments (f0) have been requested; leaves the last k-step's fragments
    // (f1) in flight: its MFMAs run after the next barrier.  Then the fragment addresses move to the other stage.
    // the pieces of the next reduction tile that G2_DMA_PLAN puts into `step`, each behind its MFMA
    auto dma_fill = [&](const half_t *na, const half_t *nw, char *nstage, auto step_tag, auto m_tag) __attribute__((always_inline)) {
        constexpr int piece = g2_piece_at(decltype(step_tag)::value, decltype(m_tag)::value);
        if constexpr (piece >= 0) dma_piece(piece < 4 ? na : nw, nstage, std::integral_constant<int, (piece >= 0 ? piece : 0)>{});
    };
    // q4 (`expand`: the pending block has not been expanded in one go at the top of the period): chunks 0, 1 behind MFMAs 1, 5
    // of k-step 0, chunks 2, 3 behind MFMAs 1, 5 of k-step 1; the block of the tile after next (rn0, rkt) is requested behind
    // MFMA 1 of k-step 2, into the registers chunk 3 has just released.
    auto q4_fill = [&](auto expand_tag, unsigned nxor, int rn0, int rkt, auto step_tag, auto m_tag) __attribute__((always_inline)) {
        if constexpr (Q4) {
            constexpr int step = decltype(step_tag)::value, m = decltype(m_tag)::value;
            if constexpr (decltype(expand_tag)::value && step <= 2 && (m == 1 || m == 5))
                q4_expand_one(nxor, std::integral_constant<int, (step - 1) * 2 + (m == 5 ? 1 : 0)>{});
            if constexpr (step == 3 && m == 1) q4_request(rn0, rkt);
        }
    };
    auto steps_0_to_2 = [&](const half_t *na, const half_t *nw, char *nstage, auto expand_tag, int rn0, int rkt) __attribute__((always_inline)) {
        const unsigned nxor = (unsigned)(nstage - smem);
        read_frag(f1, I1{}); g2_wait6(f0);
        mfma_step_with(f0, [&](auto m) __attribute__((always_inline)) { dma_fill(na, nw, nstage, I1{}, m); q4_fill(expand_tag, nxor, rn0, rkt, I1{}, m); });
        read_frag(f0, I2{}); g2_wait6(f1);
        mfma_step_with(f1, [&](auto m) __attribute__((always_inline)) { dma_fill(na, nw, nstage, I2{}, m); q4_fill(expand_tag, nxor, rn0, rkt, I2{}, m); });
        read_frag(f1, I3{}); g2_wait6(f0);
        mfma_step_with(f0, [&](auto m) __attribute__((always_inline)) { dma_fill(na, nw, nstage, I3{}, m); q4_fill(expand_tag, nxor, rn0, rkt, I3{}, m); });
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { aA[kk] ^= (unsigned)G2_STAGE; aW[kk] ^= (unsigned)G2_STAGE; }
    };
    int stage = 0;
    for (;;) {
        const int next = tile + S;
        const bool more = next < t_end;
        // (after the last output tile the stream requests this tile's first reduction tile once more: a request that
        // is never read costs less than a branch around every request)
        const int nm0 = more ? (next / cnt_n) * G2_BM : m0, nn0 = more ? (n_begin + next % cnt_n) * G2_BN : n0;
        const half_t *ta = p.A + (size_t)m0 * K, *tw = p.w16 + (size_t)n0 * K;
        {   // ---- reduction tile 0: the previous output tile is finished behind its barrier
            tile_barrier(f1);
            const half_t *na = ta + G2_BK, *nw = tw + G2_BK;
            char *nstage = smem + (stage ^ 1) * G2_STAGE;
            // (this tile's step-0 pieces go out in one go: with a previous output tile its last k-step and epilogue follow,
            // without one there is nothing to put them between)
            static_for<g2_plan_count(0)>([&](auto i) __attribute__((always_inline)) {
                constexpr int piece = decltype(i)::value;
                dma_piece(piece < 4 ? na : nw, nstage, i);
            });
            // (q4: this output tile's second weight tile, in one go — its registers must be free during the epilogue)
            if constexpr (Q4) q4_expand_all((unsigned)(nstage - smem));
            if (have_prev) {
                mfma_step(f1);                         // the last k-step of the previous output tile
                epilogue(pm0, pn0, true, m0, n0);
                init_acc();
            }
            read_frag(f0, I0{});
            steps_0_to_2(na, nw, nstage, std::false_type{}, nk > 2 ? n0 : nn0, nk > 2 ? 2 : 0);
            stage ^= 1;
        }
        for (int kt = 1; kt < nk; ++kt) {
            tile_barrier(f1);
            // the next reduction tile (of this output tile, or the first of the next one) into the stage just released
            const bool last = kt + 1 == nk;
            const half_t *na = last ? p.A + (size_t)nm0 * K : ta + (kt + 1) * G2_BK;
            const half_t *nw = last ? p.w16 + (size_t)nn0 * K : tw + (kt + 1) * G2_BK;
            char *nstage = smem + (stage ^ 1) * G2_STAGE;
            read_frag(f0, I0{});
            // the previous reduction tile's last k-step, with the new tile's first requests between its MFMAs
            mfma_step_with(f1, [&](auto m) __attribute__((always_inline)) { dma_fill(na, nw, nstage, I0{}, m); });
            steps_0_to_2(na, nw, nstage, std::true_type{}, kt + 2 < nk ? n0 : nn0, kt + 2 < nk ? kt + 2 : kt + 2 - nk);
            stage ^= 1;
        }
        have_prev = true; pm0 = m0; pn0 = n0;
        if (!more) break;
        tile = next; m0 = nm0; n0 = nn0;
    }
    // the request issued behind the last output tile must not outlive the workgroup
    tile_barrier(f1);
    mfma_step(f1);
    epilogue(pm0, pn0, false, 0, 0);
}

bool gemm256_supported(const GemmWeight &W, int M_pad) {
    return (W.type == GW_F16 ? W.w16 != nullptr : W.qs != nullptr && W.sc != nullptr) && W.N % G2_BN == 0 && W.K % G2_BK == 0 && W.K >= 2 * G2_BK && M_pad % G2_BM == 0 && M_pad > 0;
}

void launch_gemm256(const GemmWeight &W, const half_t *A, const float *bias, const half_t *resid, half_t *C, int M_pad,
                    int epilogue, hipStream_t stream, const GemmLnFold *ln) {
    Gemm256Args a;
    a.A = A; a.w16 = W.w16; a.bias = bias; a.resid = resid; a.C = C; a.qs = W.qs; a.sc = W.sc;
    a.rows_in = ln ? ln->rows_in : nullptr; a.waug = ln ? ln->waug : nullptr; a.rows_res = ln ? ln->rows_res : nullptr;
    a.gb = ln ? ln->gb : nullptr; a.stats = ln ? ln->stats : nullptr; a.stats_p = 2 * (W.N / G2_BN);
    a.N = W.N; a.K = W.K; a.n_tiles_n = W.N / G2_BN;
    a.n_tiles = a.n_tiles_n * (M_pad / G2_BM);
    // feature groups: only where W (N x K f16) overflows an XCD's L2 share and reading the activations twice is the cheaper
    // side (K small against N), with enough token tiles for the 8 / G token ranges to balance
    // (measured, bert-base / mpnet dimensions: up-projection -2.5 %; FETCH_SIZE per launch in profiles/)
    a.n_groups = 1;
    // (a q4 matrix is 9 / 32 or 10 / 32 of that: the up-projection's 1.3 MiB stay resident beside everything else)
    if (W.type == GW_F16 && (size_t)W.N * W.K * 2 > (size_t)3 << 20 && W.N >= 4 * W.K && a.n_tiles_n % 2 == 0 && M_pad / G2_BM >= 64) a.n_groups = 2;
    // one persistent workgroup per CU (256 on an MI355X, a multiple of the 8 XCDs), fewer when there are fewer tiles
    static int n_cu[MAX_HIP_DEVICES] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < MAX_HIP_DEVICES && !n_cu[dev]) {
        hipDeviceProp_t prop;
        n_cu[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8 ? prop.multiProcessorCount / 8 * 8 : 256;
    }
    const int cus = dev >= 0 && dev < MAX_HIP_DEVICES ? n_cu[dev] : 256;
    const int grid = std::min(cus, (a.n_tiles + 7) / 8 * 8);
    const size_t lds = 2 * G2_STAGE + 8 * 4096;        // 128 KiB of reduction tiles + 8 wave-private staging areas
    static DeviceFlags configured[13];
    auto go = [&](auto kernel, int e) {
        configure_once(configured[e], [&] { (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
        BERT_LAUNCH(kernel, dim3(grid), dim3(512), lds, stream, a);
    };
    auto by_type = [&](auto epi_tag) {
        constexpr int E = decltype(epi_tag)::value;
        switch (W.type) {
            case GW_F16: go(gemm256_kernel<E, GW_F16>, E); break;
            case GW_Q4_0: go(gemm256_kernel<E, GW_Q4_0>, 3 + E); break;
            default: go(gemm256_kernel<E, GW_Q4_1>, 6 + E); break;
        }
    };
    if (ln && ln->flags && W.type == GW_F16) {
        // (the folded-LayerNorm forms: f16 images only — the engine does not fold where a matrix stays on 4-bit planes)
        if (epilogue == EPI_BIAS && ln->flags == GemmLnFold::IN) go(gemm256_kernel<EPI_BIAS, GW_F16, LN_IN>, 9);
        else if (epilogue == EPI_BIAS_GELU && ln->flags == GemmLnFold::IN) go(gemm256_kernel<EPI_BIAS_GELU, GW_F16, LN_IN>, 10);
        else if (epilogue == EPI_BIAS_RESID && ln->flags == GemmLnFold::STATS) go(gemm256_kernel<EPI_BIAS_RESID, GW_F16, LN_STATS>, 11);
        else if (epilogue == EPI_BIAS_RESID && ln->flags == (GemmLnFold::RES | GemmLnFold::STATS)) go(gemm256_kernel<EPI_BIAS_RESID, GW_F16, LN_RES | LN_STATS>, 12);
        else fprintf(stderr, "launch_gemm256: unsupported LayerNorm-folding form (epilogue %d, flags %d): nothing launched\n", epilogue, ln->flags);
        return;
    }
    switch (epilogue) {
        case EPI_BIAS: by_type(std::integral_constant<int, EPI_BIAS>{}); break;
        case EPI_BIAS_GELU: by_type(std::integral_constant<int, EPI_BIAS_GELU>{}); break;
        default: by_type(std::integral_constant<int, EPI_BIAS_RESID>{}); break;
    }
}

}  // namespace bert_hip
