// gemm192.hip — the weight mat-mul of the H > 384 models (bert-base / mpnet dimensions, BASELINE configs 3, 4) with the
// output tile's epilogue UNDER the next tile's MFMAs:
//   C[t][n] = epilogue( sum_k A[t][k] * W[n][k] + bias[n] (+ resid[t][n]) )        A: f16 activations, W: f16 image
// Same operation, epilogues and BITS as gemm256.hip (reference bert.cpp:822-839, :859-865, :878-882, :885-891: the sum
// starts from bias (+ residual) in f32, k ascending on v_mfma_f32_32x32x16_f16, one rounding, packed-f16 GELU).  gemm256's main
// loop runs at the rate of the vendor's 256 x 256 x 64 kernel (1.45 us per reduction tile); what both pay on the K = 768 shapes
// is the output-tile boundary — 8 / 11 / 14-25 us (bias / GELU / residual) per 17 us of main loop, 6 us in the vendor kernel —
// with the matrix cores idle: conversion, staging, 128 store instructions per CU and their acknowledgement in front of the next
// barrier.  A wave of gemm256 has no register left to carry a finished tile into the next one (128 accumulators + 64 packed
// results + 48 fragment registers).  This kernel gives up a quarter of the tile's width for exactly that:
//   * 256 tokens x 192 features per workgroup of 8 waves (two per SIMD), a wave owns 96 features x 64 tokens = 3 x 2
//     accumulator blocks (96 registers): 5 fragment reads per 6 MFMAs (gemm256: 6 per 8), 56 KiB of LDS-DMA per 192 MFMAs;
//   * the FIRST k-step of an output tile takes the bias vector (from LDS: the whole bias is copied there once per launch) —
//     plus, for the residual form, the residual tile — as the MFMAs' C operand and writes the accumulators: no init pass; in
//     front of each of these MFMAs the block's old accumulators are rounded into 8 of the wave's 48 RESULT registers (GELU is
//     applied later, on the packed halves: gemm256 rounds before it too);
//   * the 12 x 16-byte store instructions of a wave (v_permlane32_swap pairs turn two 4-feature runs of the accumulator layout
//     into 8-feature runs: no LDS staging) go out two per reduction tile between the MFMAs of the next output tile's first six
//     reduction tiles, with the GELU arithmetic of their four registers as VALU filler in front; a reduction-tile barrier
//     waits for the wave's LDS-DMA pieces only (vmcnt counts the stores issued BEHIND them and leaves those in flight);
//   * the residual of the NEXT output tile is requested (12 loads per wave, the stores' own shape) into the result registers
//     once their stores are out — reduction tiles nk - 5 .. nk - 2 — and swapped back into the accumulator layout at the tile
//     boundary: it costs no register of its own and no load round trip at the boundary.
// Persistent walk, XCD ranges, feature-tile groups, tile swizzle and the hand-issued fragment reads are gemm256's.
// Shapes: N % 192 == 0 (768, 2304, 3072), K % 64 == 0 and K >= 768 (six reduction tiles of stores + four of residual requests
// + two), M_pad % 256 == 0, N <= 8192 (bias in LDS); everything else stays on gemm256 / gemm_mfma.
#include "tile_stream.h"

#include <cstdlib>
#include <type_traits>

namespace bert_hip {

namespace {

#define G3_GLOBAL(p) ((const __attribute__((address_space(1))) void *)(p))
#define G3_LDS(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int G3_BM = 256, G3_BN = 192, G3_BK = 64;
// LDS: a ring of THREE activation tiles (256 rows x 128 bytes: requested two reduction tiles ahead — they come from HBM, and the
// finished tiles' stores share the way), a ring of two weight tiles (192 rows: one ahead, L2), eight 2 KiB staging areas = 160 KiB
constexpr int G3_A_SLOT = 256 * 128, G3_W_BASE = 3 * G3_A_SLOT, G3_W_SLOT = 192 * 128, G3_STG_BASE = G3_W_BASE + 2 * G3_W_SLOT;
constexpr int G3_LDS_BYTES = G3_STG_BASE + 8 * 2048;
static_assert(G3_LDS_BYTES == 160 * 1024, "the whole LDS of a CU");

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Gemm192Args {
    const half_t *A;        // [M_pad][K], M_pad % 256 == 0
    const half_t *w16;      // [N_pad][K], N % 192 == 0
    const float *bias;      // [N]
    const half_t *resid;    // [M_pad][N] or null
    half_t *C;              // [M_pad][N]
    int N, K, n_tiles_n, n_tiles;
    int n_groups;           // feature-tile groups (gemm256.hip)
};

// the order in which a wave's six blocks (2 i + j: block row i, token block j) leave and the next tile's residual blocks arrive: the
// three 64-byte pieces of a token block's rows back to back
constexpr int g3_block(int k) { return (k % 3) * 2 + k / 3; }

template <int OFF>
__device__ __forceinline__ f16x8 g3_read_b128(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ f32x4 g3_read_f32x4(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
struct G3Frag {
    f16x8 a[3], b[2];                                  // weight rows (3 x 32 features), activation rows (2 x 32 tokens) of one k-step
};
// everything but the newest five reads (the next k-step's) has landed
__device__ __forceinline__ void g3_wait5(G3Frag &f) {
    asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.b[0]), "+v"(f.b[1]) : : "memory");
}
// reduction-tile barrier: this wave's LDS-DMA pieces of the tile have landed — everything but the PEND newest vector-memory
// operations, which are the stores / residual requests issued behind the pieces —, its fragment reads of the previous tile
// have returned (the fragments of that tile's last k-step are named: their MFMAs run after the barrier)
template <int PEND>
__device__ __forceinline__ void g3_barrier(G3Frag &f) {
    asm volatile("s_waitcnt vmcnt(%5) lgkmcnt(0)\n\ts_barrier"
                 : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.b[0]), "+v"(f.b[1]) : "n"(PEND) : "memory");
}

}  // namespace

#ifdef BERT_HIP_TIMELINE
static __device__ unsigned long long g3_clock[256 * 8 * 16];
#endif

// G3_ABLATE (tuning builds only, results are wrong): bit 0 no global stores, bit 1 nothing rides on the reduction tiles (no GELU, swaps,
// stores, residual requests), bit 2 no conversion at the tile boundary, bit 3 every tile request to L2-resident addresses
#ifndef G3_ABLATE
#define G3_ABLATE 0
#endif
#ifndef G3_STORE_POLICY
#define G3_STORE_POLICY "nt"
#endif

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm192_kernel(Gemm192Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool RESID = EPI == EPI_BIAS_RESID;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wf = wave & 1, wq = wave >> 1;         // feature half (96) / token quarter (64) of the tile
    const int l31 = lane & 31, hi = lane >> 5;
    const int K = p.K, nk = K / G3_BK, N = p.N;

    // ---- this workgroup's output tiles (gemm256.hip: per XCD a contiguous range, the feature tiles of a token tile back to back)
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

    // ---- LDS-DMA: a reduction tile is 32 + 24 pieces of 1 KiB (8 rows each); a wave issues activation pieces 4 wave .. + 3 and
    // weight pieces 3 wave .. + 2.  Source offsets (elements): row (lane >> 3) of the piece, 16-byte chunk (lane & 7) ^ ((r >> 1) & 7)
    // with r = 8 piece + (lane >> 3): ((lane >> 4) + 4 (piece & 1)) & 7.
    // An odd piece's chunk differs in bit 2: offset ^ 32 (K is a multiple of 64).
    const unsigned off_e = (unsigned)((lane >> 3) * K) + ((((unsigned)lane & 7) ^ (((unsigned)lane >> 4) & 7)) << 3);
    // (slots as LDS byte offsets, scalars: a_off of the activation tile being multiplied, w_off of its weight tile)
    auto dma_a = [&](const half_t *src, int slot_off, auto i_tag) __attribute__((always_inline)) {
        constexpr int i = decltype(i_tag)::value;       // 0..3
        const int pc = wave * 4 + i;
        __builtin_amdgcn_global_load_lds(G3_GLOBAL(src + (size_t)pc * 8 * K + (off_e ^ ((i & 1) ? 32u : 0u))), G3_LDS(smem + slot_off + pc * 1024), 16, 0, 0);
    };
    auto dma_w = [&](const half_t *src, int slot_off, auto i_tag) __attribute__((always_inline)) {
        constexpr int i = decltype(i_tag)::value;       // 0..2
        const int pc = wave * 3 + i;
        __builtin_amdgcn_global_load_lds(G3_GLOBAL(src + (size_t)pc * 8 * K + (off_e ^ (unsigned)((pc & 1) << 5))), G3_LDS(smem + slot_off + pc * 1024), 16, 0, 0);
    };

    // ---- fragment addresses: k-step kk reads at ((address of k-step 0 in slot 0) ^ (kk << 5)) + slot offset — the chunk swizzle is
    // an XOR and a row's 128 bytes are aligned
    const unsigned aA0 = (unsigned)(size_t)smem + (unsigned)((wq * 64 + l31) * 128) + (unsigned)(((hi ^ ((l31 >> 1) & 7)) << 4));
    const unsigned aW0 = (unsigned)(size_t)smem + (unsigned)((wf * 96 + l31) * 128) + (unsigned)(((hi ^ ((l31 >> 1) & 7)) << 4));
    int a_off = 0, w_off = G3_W_BASE;                  // (scalars) the slots the fragment reads go to
    auto read_frag = [&](G3Frag &f, auto kk_tag) __attribute__((always_inline)) {
        constexpr int kk = decltype(kk_tag)::value;
        const unsigned w = (aW0 ^ (unsigned)(kk << 5)) + (unsigned)w_off, a = (aA0 ^ (unsigned)(kk << 5)) + (unsigned)a_off;
        f.a[0] = g3_read_b128<0>(w); f.a[1] = g3_read_b128<4096>(w); f.a[2] = g3_read_b128<8192>(w);
        f.b[0] = g3_read_b128<0>(a); f.b[1] = g3_read_b128<4096>(a);
    };
    // the slots after the ones being multiplied: the weight tile one ahead goes to w_next(), the activation tile two ahead to a_prev()
    // (the slot whose tile was multiplied in the reduction tile before this one)
    auto w_next = [&]() __attribute__((always_inline)) { return w_off == G3_W_BASE ? G3_W_BASE + G3_W_SLOT : G3_W_BASE; };
    auto a_prev = [&]() __attribute__((always_inline)) { return a_off == 0 ? 2 * G3_A_SLOT : a_off - G3_A_SLOT; };
    auto advance_slots = [&]() __attribute__((always_inline)) {
        a_off = a_off == 2 * G3_A_SLOT ? 0 : a_off + G3_A_SLOT;
        w_off = w_next();
    };

    f32x16 acc[3][2];                                 // [feature block][token block]
    // the finished tile, rounded, in the accumulator layout: ou[4 b + g] = features 32 i + 8 g + 4 hi .. + 3 (two f16 pairs) of token
    // 32 j + l31, block b = 2 i + j; the same registers receive the next tile's residual once the block has left
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 ou[24];
    // ---- a finished block [32 features][32 tokens] leaves through the wave's 2 KiB STAGING area (behind the tile rings): four ds_write_b64 in the accumulator layout, two
    // ds_read_b128 in the row layout, two global stores of 16 rows x 64 bytes.  (Stored straight from the accumulator layout —
    // v_permlane32_swap pairs make 16 bytes per lane — an instruction touches 32 rows x 2 x 16 bytes, no two lanes of a quad in one
    // line: 7.5 us per output tile and CU went into the stores' 64 requests each, 8 us into residual loads of that shape.)
    // LDS image [32 rows][64 bytes], the 16-byte chunk c of row r at position c ^ ((r >> 1) & 3).
    char *const stg = smem + G3_STG_BASE + wave * 2048;
    // (the three lane-dependent addresses below are rebuilt from a fresh lane id at every use — v_mbcnt: two instructions, no
    // register held across the reduction tiles that have none to spare, and no scratch reload among the hand-counted requests)
    auto fresh_lane = []() __attribute__((always_inline)) { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); };
    // accumulator-layout address of this lane's run g: (w_addr() ^ (g << 4)); row-layout address of its 16 bytes of rows t 16 + (lane >> 2)
    auto w_addr = [&]() __attribute__((always_inline)) {
        const int ln = fresh_lane(), r = ln & 31;
        return (unsigned)(size_t)stg + (unsigned)(r * 64 + (ln >> 5) * 8 + (((r >> 1) & 3) << 4));
    };
    auto r_addr = [&]() __attribute__((always_inline)) {
        const int ln = fresh_lane();
        return (unsigned)(size_t)stg + (unsigned)((ln >> 2) * 64 + (((ln & 3) ^ ((ln >> 3) & 3)) << 4));
    };
    // byte offset of this lane's 16 bytes inside an output tile (row lane >> 2 of the wave's token quarter, chunk lane & 3 of its
    // feature half); SWZ: the chunk the LDS-DMA's linear write puts at position lane & 3 of its row (a residual request)
    auto st_voff = [&](bool swz) __attribute__((always_inline)) {
        const int ln = fresh_lane();
        const int ch = swz ? ((ln & 3) ^ ((ln >> 3) & 3)) : (ln & 3);
        return (unsigned)(((wq * 64 + (ln >> 2)) * N + wf * 96 + ch * 8) * 2);
    };
    auto block_soff = [&](auto b_tag, int t) __attribute__((always_inline)) {
        constexpr int b = decltype(b_tag)::value, j = b & 1, i = b >> 1;
        return (size_t)(((size_t)(j * 32 + t * 16) * N + i * 32) * 2);
    };
    auto gelu_reg = [&](auto b_tag, auto k_tag) __attribute__((always_inline)) {
        constexpr int b = decltype(b_tag)::value, k = decltype(k_tag)::value;       // register k = 2 g + (0, 1) of block b
        if constexpr (EPI == EPI_BIAS_GELU) {
            const unsigned w = ou[4 * b + (k >> 1)][k & 1];
            ou[4 * b + (k >> 1)][k & 1] = __builtin_bit_cast(unsigned, gelu_pk16h(__builtin_bit_cast(f16x2_t, w)));
        }
    };
    auto stage_write = [&](auto b_tag) __attribute__((always_inline)) {
        constexpr int b = decltype(b_tag)::value;
        const unsigned wa = w_addr();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32x2 d = ou[4 * b + g];
            const unsigned a = wa ^ (unsigned)(g << 4);
            asm volatile("ds_write_b64 %0, %1" : : "v"(a), "v"(d) : "memory");
        }
    };
    u32x4 srow[2];                                     // the block in the row layout, between its LDS reads and its stores
    auto stage_read = [&]() __attribute__((always_inline)) {
        const unsigned ra = r_addr();
        u32x4 t0, t1;
        asm volatile("ds_read_b128 %0, %1" : "=v"(t0) : "v"(ra) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(t1) : "v"(ra) : "memory");
        srow[0] = t0; srow[1] = t1;
    };
    auto store_rows = [&](const char *ctile, auto b_tag) __attribute__((always_inline)) {
        const unsigned vo = st_voff(false);
        u32x4 t0 = srow[0], t1 = srow[1];
        const char *b0 = ctile + block_soff(b_tag, 0), *b1 = ctile + block_soff(b_tag, 1);
        if (G3_ABLATE & 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0), "+v"(t1) : "v"(vo), "s"(b0), "s"(b1) : "memory");
        // (s_nop 4: the base addresses may have come back from a spill lane by v_readlane just in front — a VALU write of an SGPR
        // needs five wait states before a vector-memory instruction reads it, and the compiler pads nothing inside an asm)
        else if (G3_ABLATE & 128) asm volatile("s_nop 4\n\ts_waitcnt lgkmcnt(0)\n\tglobal_store_dwordx4 %2, %0, %3 " G3_STORE_POLICY "\n\tglobal_store_dwordx4 %2, %1, %4 " G3_STORE_POLICY "\n\ts_nop 1"
                          : "+v"(t0), "+v"(t1) : "v"(vo), "s"(b0), "s"(b1) : "memory");
        else asm volatile("s_nop 4\n\ts_waitcnt lgkmcnt(0)\n\tglobal_store_dwordx4 %2, %0, %3\n\tglobal_store_dwordx4 %2, %1, %4\n\ts_nop 1"
                          : "+v"(t0), "+v"(t1) : "v"(vo), "s"(b0), "s"(b1) : "memory");
    };
    // the residual block b of the next output tile: two LDS-DMA instructions (16 rows x 64 bytes each) into the staging area ...
    auto resid_request = [&](const char *rtile, auto b_tag) __attribute__((always_inline)) {
        if constexpr (RESID) {
            const unsigned ld_voff = st_voff(true);
            __builtin_amdgcn_global_load_lds(G3_GLOBAL(rtile + block_soff(b_tag, 0) + ld_voff), G3_LDS(stg), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(G3_GLOBAL(rtile + block_soff(b_tag, 1) + ld_voff), G3_LDS(stg + 1024), 16, 0, 0);
        }
    };
    // ... and, one reduction tile later, from there into the block's result registers in the accumulator layout
    auto resid_fetch = [&](auto b_tag) __attribute__((always_inline)) {
        constexpr int b = decltype(b_tag)::value;
        if constexpr (RESID) {
            const unsigned wa = w_addr();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned a = wa ^ (unsigned)(g << 4);
                u32x2 d;
                asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(a) : "memory");
                ou[4 * b + g] = d;
            }
        }
    };
    // a block's way out as the filler of a reduction tile's k-steps 1 and 2: GELU of its eight registers behind MFMAs 0..5 of
    // k-step 1 and 0, 1 of k-step 2, the staging writes behind MFMA 2, the row reads behind 3, the two stores behind 5
    auto store_fill_1 = [&](auto b_tag, auto m_tag) __attribute__((always_inline)) { gelu_reg(b_tag, m_tag); };
    auto store_fill_2 = [&](const char *ctile, auto b_tag, auto m_tag) __attribute__((always_inline)) {
        constexpr int m = decltype(m_tag)::value;
        if constexpr (m == 0) gelu_reg(b_tag, std::integral_constant<int, 6>{});
        else if constexpr (m == 1) gelu_reg(b_tag, std::integral_constant<int, 7>{});
        else if constexpr (m == 2) stage_write(b_tag);
        else if constexpr (m == 3) stage_read();
        else if constexpr (m == 5) store_rows(ctile, b_tag);
    };

#define G3_OU_ALL "+v"(ou[0]), "+v"(ou[1]), "+v"(ou[2]), "+v"(ou[3]), "+v"(ou[4]), "+v"(ou[5]), "+v"(ou[6]), "+v"(ou[7]), "+v"(ou[8]), "+v"(ou[9]), \
                  "+v"(ou[10]), "+v"(ou[11]), "+v"(ou[12]), "+v"(ou[13]), "+v"(ou[14]), "+v"(ou[15]), "+v"(ou[16]), "+v"(ou[17]), "+v"(ou[18]), \
                  "+v"(ou[19]), "+v"(ou[20]), "+v"(ou[21]), "+v"(ou[22]), "+v"(ou[23])
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>;

    // the 6 MFMAs of a k-step with `fill(m)` pinned behind MFMA m
    auto mfma_step_with = [&](const G3Frag &f, auto fill) __attribute__((always_inline)) {
        static_for<6>([&](auto m_tag) __attribute__((always_inline)) {
            constexpr int m = decltype(m_tag)::value, i = m >> 1, j = m & 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
            fill(m_tag);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto no_fill = [](auto) __attribute__((always_inline)) {};
    // the requests of a reduction tile: the weight tile ONE ahead (three pieces, first: the next barrier waits for them and leaves
    // everything issued behind them in flight) between the MFMAs of the deferred last k-step, the activation tile TWO ahead (four
    // pieces) between those of k-step 0
    auto dma_fill_w = [&](const half_t *nw, auto m_tag) __attribute__((always_inline)) {
        constexpr int m = decltype(m_tag)::value;
        if constexpr (m == 0) dma_w(nw, w_next(), I0{});
        else if constexpr (m == 2) dma_w(nw, w_next(), I1{});
        else if constexpr (m == 4) dma_w(nw, w_next(), I2{});
    };
    auto dma_fill_a = [&](const half_t *na, auto m_tag) __attribute__((always_inline)) {
        constexpr int m = decltype(m_tag)::value;
        if constexpr (m == 0) dma_a(na, a_prev(), I0{});
        else if constexpr (m == 1) dma_a(na, a_prev(), I1{});
        else if constexpr (m == 3) dma_a(na, a_prev(), I2{});
        else if constexpr (m == 4) dma_a(na, a_prev(), I3{});
    };

    G3Frag f0, f1;
    // k-steps 1 and 2 of a reduction tile (f1 requested, f0 consumed by the caller's k-step 0) with the work that rides on them, fixed
    // at compile time (a block's registers are): KIND 1 = block IDX of the previous output tile leaves, KIND 2 = residual block
    // IDX of the next one is requested (and block IDX - 1, requested one reduction tile earlier, fetched: by the caller, in front
    // of the tile's first MFMAs), KIND 0 = nothing.  Returns the number of vector-memory operations issued behind the tile's pieces
    // that the next barrier may leave in flight.
    auto steps_1_2 = [&](auto kind_tag, auto idx_tag, const char *ctile, const char *rtile) __attribute__((always_inline)) -> int {
        constexpr int KIND = (G3_ABLATE & 2) ? 0 : decltype(kind_tag)::value, IDX = decltype(idx_tag)::value;
        using B = std::integral_constant<int, g3_block(IDX)>;
        read_frag(f0, I2{}); g3_wait5(f1);
        mfma_step_with(f1, [&](auto m) __attribute__((always_inline)) {
            constexpr int mm = decltype(m)::value;
            if constexpr (KIND == 1) store_fill_1(B{}, m);
            else if constexpr (KIND == 2 && mm == 0) resid_request(rtile, B{});
        });
        read_frag(f1, I3{}); g3_wait5(f0);
        mfma_step_with(f0, [&](auto m) __attribute__((always_inline)) {
            if constexpr (KIND == 1) store_fill_2(ctile, B{}, m);
        });
        advance_slots();
        return KIND == 1 ? ((G3_ABLATE & 1) ? 0 : 2) : 0;      // (a residual request must have landed at the next barrier)
    };
#ifdef BERT_HIP_TIMELINE
    // phase clock of the tuning build: per kind of reduction tile (0 plain, 1 a block leaves, 2 residual request, 3 the first tile of
    // an output tile) the cycles a wave spends in front of the barrier's vmcnt wait, in the wait, in the barrier, and in the tile's work
    unsigned long long clk[4][4] = {};
    unsigned long long t_prev = 0;
    int clk_kind = 0;
    auto clocked_barrier = [&](int pend, int next_kind) __attribute__((always_inline)) {
        unsigned long long t0, t1, t2;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
        if (pend == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)"
                     : "=s"(t1), "=s"(t2), "+v"(f1.a[0]), "+v"(f1.a[1]), "+v"(f1.a[2]), "+v"(f1.b[0]), "+v"(f1.b[1]) : : "memory");
        if (t_prev) { clk[clk_kind][0] += t0 - t_prev; clk[clk_kind][1] += t1 - t0; clk[clk_kind][2] += t2 - t1; clk[clk_kind][3] += 1; }
        t_prev = t2; clk_kind = next_kind;
    };
    auto barrier_pend = [&](int pend) __attribute__((always_inline)) { clocked_barrier(pend, 0); };
#else
    auto barrier_pend = [&](int pend) __attribute__((always_inline)) {
        // (the four activation pieces of the tile after next were issued last, or in front of a leaving block's two stores)
        if (pend == 0) g3_barrier<4>(f1);
        else g3_barrier<6>(f1);
    };
#endif

    // ---- the rounded results of block (i, j): accumulators -> units 4 i + 2 j, 4 i + 2 j + 1 (GELU later, on the packed halves)
    auto convert_block = [&](auto m_tag) __attribute__((always_inline)) {
        constexpr int m = decltype(m_tag)::value, i = m >> 1, j = m & 1;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const f16x2_t h = {(_Float16)acc[i][j][4 * g + 2 * r], (_Float16)acc[i][j][4 * g + 2 * r + 1]};
                ou[4 * m + g][r] = __builtin_bit_cast(unsigned, h);
            }
    };
    // the C operand of block (i, j)'s first MFMA: register e = bias[feature 32 i + 8 (e >> 2) + 4 hi + (e & 3)] (+ the residual that
    // waits in the block's registers).  The bias comes through SCALAR loads (a wave-uniform address: 32 consecutive floats per
    // block row, the lane halves pick theirs) — the LDS has no byte left for it beside the rings and the staging areas.
    // (hand-issued s_load_dwordx16: the compiler takes a uniform address through the VECTOR memory path when stores may alias it,
    // and waits for such a load with a vmcnt(0) that drains the LDS-DMA pieces in flight)
    auto bias_row = [&](const float *brow, bool upper) __attribute__((always_inline)) -> f32x16 {
        f32x16 s0, s1;
        asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(s0), "=&s"(s1) : "s"(brow) : "memory");
        f32x16 c;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = 8 * (e >> 2) + (e & 3);
            const float lo = k < 16 ? s0[k] : s1[k - 16], hi_ = k + 4 < 16 ? s0[k + 4] : s1[k + 4 - 16];
            c[e] = upper ? hi_ : lo;
        }
        return c;
    };
    auto first_c = [&](const f32x16 &cb, auto m_tag) __attribute__((always_inline)) -> f32x16 {
        constexpr int m = decltype(m_tag)::value;
        f32x16 c = cb;
        if constexpr (RESID) {
            // c = bias + (float)residual in ONE instruction per value (v_fma_mix_f32: f16 x 1.0 + f32, the f32 add's bits): no
            // converted copy of the block beside the bias runs
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned w = ou[4 * m + (e >> 2)][(e & 3) >> 1];
                const float bb = cb[e];
                float r;
                if (e & 1) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(bb));
                else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(bb));
                c[e] = r;
            }
        }
        return c;
    };
    // k-step 0 of an output tile (fragments f0 requested; F1_EARLY: f1 = k-step 1 requested behind them): per block, the old
    // accumulators into the result registers (PREV), then the first MFMA with C = bias (+ residual); the activation pieces of the
    // tile after next ride along
    // (the residual form has no registers for k-step 1's fragments beside the residual and the C operand: it requests them behind
    // this step, F1_EARLY false)
    constexpr bool F1_EARLY = !RESID;
    auto first_step = [&](auto prev_tag, const float *btile, const half_t *na) __attribute__((always_inline)) {
        constexpr bool PREV = decltype(prev_tag)::value;
        const bool upper = fresh_lane() >= 32;
        if constexpr (F1_EARLY) g3_wait5(f0);
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0.a[0]), "+v"(f0.a[1]), "+v"(f0.a[2]), "+v"(f0.b[0]), "+v"(f0.b[1]) : : "memory");
        static_for<3>([&](auto i_tag) __attribute__((always_inline)) {
            constexpr int i = decltype(i_tag)::value;
            const f32x16 cb = bias_row(btile + wf * 96 + i * 32, upper);
            static_for<2>([&](auto j_tag) __attribute__((always_inline)) {
                constexpr int j = decltype(j_tag)::value;
                using M = std::integral_constant<int, i * 2 + j>;
                const f32x16 c = first_c(cb, M{});
                if constexpr (PREV && !(G3_ABLATE & 4)) convert_block(M{});
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0.a[i], f0.b[j], c, 0, 0, 0);
                dma_fill_a(na, M{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        if constexpr (!F1_EARLY) read_frag(f1, I1{});
    };

    int m0 = (tile / cnt_n) * G3_BM, n0 = (n_begin + tile % cnt_n) * G3_BN;
    {   // the first two activation tiles and the first weight tile of the first output tile
        const half_t *a = p.A + (size_t)m0 * K, *w = p.w16 + (size_t)n0 * K;
        static_for<4>([&](auto i) __attribute__((always_inline)) { dma_a(a, 0, i); });
        static_for<4>([&](auto i) __attribute__((always_inline)) { dma_a(a + G3_BK, G3_A_SLOT, i); });
        static_for<3>([&](auto i) __attribute__((always_inline)) { dma_w(w, G3_W_BASE, i); });
    }
    if constexpr (RESID) {   // the first tile's residual, block by block through the staging area: six exposed round trips per launch
        const char *rt = (const char *)p.resid + ((size_t)m0 * N + n0) * 2;
        static_for<6>([&](auto b) __attribute__((always_inline)) {
            resid_request(rt, b);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            resid_fetch(b);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        });
        asm volatile("" : G3_OU_ALL : : "memory");       // (named outside the generic lambda: it would not capture what only an asm operand uses)
    }
    {   // ---- reduction tile 0 of the first output tile
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const half_t *ta0 = p.A + (size_t)m0 * K, *tw0 = p.w16 + (size_t)n0 * K;
        dma_w(tw0 + G3_BK, w_next(), I0{}); dma_w(tw0 + G3_BK, w_next(), I1{}); dma_w(tw0 + G3_BK, w_next(), I2{});
        read_frag(f0, I0{});
        if constexpr (F1_EARLY) read_frag(f1, I1{});
        first_step(std::false_type{}, p.bias + n0, ta0 + 2 * G3_BK);
        (void)steps_1_2(I0{}, I0{}, nullptr, nullptr);
    }
    int pend = 0;
    bool have_prev = false;
    const char *ctile = nullptr;                       // the output tile whose results travel in the result registers
    for (;;) {
        const int next = tile + S;
        const bool more = next < t_end;
        // (after the last output tile the stream requests this tile's first reduction tiles once more: requests that are
        // never read cost less than a branch around every request)
        const int nm0 = more ? (next / cnt_n) * G3_BM : m0, nn0 = more ? (n_begin + next % cnt_n) * G3_BN : n0;
        const half_t *ta = p.A + (size_t)m0 * K, *tw = p.w16 + (size_t)n0 * K;
        const half_t *nta = p.A + (size_t)nm0 * K, *ntw = p.w16 + (size_t)nn0 * K;
        const char *rtile = (const char *)p.resid + ((size_t)nm0 * N + nn0) * 2;
        // reduction tile kt >= 1 of the current output tile
        auto period = [&](int kt, auto kind_tag, auto idx_tag) __attribute__((always_inline)) {
            constexpr int KIND = decltype(kind_tag)::value, IDX = decltype(idx_tag)::value;
#ifdef BERT_HIP_TIMELINE
            clocked_barrier(pend, KIND);
#else
            barrier_pend(pend);
#endif
            // the weight tile one ahead and the activation tile two ahead (of this output tile, or the first ones of the next) into
            // the slots the barrier has just released
            // (a reduction tile that carries a block's way out is one of the first six: nk >= 12)
            const half_t *nw = KIND != 1 && kt + 1 >= nk ? ntw : tw + (kt + 1) * G3_BK;
            const half_t *na = KIND != 1 && kt + 2 >= nk ? nta + (kt + 2 - nk) * G3_BK : ta + (kt + 2) * G3_BK;
            if (G3_ABLATE & 8) { na = p.A + (size_t)(blockIdx.x & 7) * 256 * K; nw = p.w16; }      // (every request to tiles that stay in the L2)
            read_frag(f0, I0{});
            // (the residual block requested one reduction tile ago has landed behind the barrier's vmcnt(0): into its registers)
            if constexpr (KIND == 2 && IDX > 0 && !(G3_ABLATE & 2)) resid_fetch(std::integral_constant<int, g3_block(IDX > 0 ? IDX - 1 : 0)>{});
            // the previous reduction tile's last k-step, with the weight requests between its MFMAs
            mfma_step_with(f1, [&](auto m) __attribute__((always_inline)) { dma_fill_w(nw, m); });
            read_frag(f1, I1{}); g3_wait5(f0);
            mfma_step_with(f0, [&](auto m) __attribute__((always_inline)) { dma_fill_a(na, m); });
            pend = steps_1_2(kind_tag, idx_tag, ctile, rtile);
        };
        const bool loads = RESID && more;              // the next tile's residual: requests in reduction tiles nk - 6 .. nk - 1
        int kt = 1;
        if (have_prev) {                               // the previous tile's blocks 1 .. 5 leave (block 0 left in reduction tile 0)
            period(1, I1{}, I1{}); period(2, I1{}, I2{}); period(3, I1{}, I3{}); period(4, I1{}, I4{}); period(5, I1{}, I5{});
            kt = 6;
        }
        for (const int kt_end = loads ? nk - 6 : nk; kt < kt_end; ++kt) period(kt, I0{}, I0{});
        if (loads) {
            period(nk - 6, I2{}, I0{}); period(nk - 5, I2{}, I1{}); period(nk - 4, I2{}, I2{});
            period(nk - 3, I2{}, I3{}); period(nk - 2, I2{}, I4{}); period(nk - 1, I2{}, I5{});
        }
        if (!more) break;
        // ---- reduction tile 0 of the next output tile: the finished tile's last k-step, then k-step 0 with the hand-over
        ctile = (const char *)p.C + ((size_t)m0 * N + n0) * 2;
        if (G3_ABLATE & 16) ctile = (const char *)p.C;
        tile = next; m0 = nm0; n0 = nn0;
        // (the residual form: the last residual request has landed: vmcnt(0))
        if constexpr (RESID)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(f1.a[0]), "+v"(f1.a[1]), "+v"(f1.a[2]), "+v"(f1.b[0]), "+v"(f1.b[1]), G3_OU_ALL : : "memory");
#ifdef BERT_HIP_TIMELINE
        else clocked_barrier(pend, 3);
#else
        else barrier_pend(pend);
#endif
        read_frag(f0, I0{});
        if constexpr (RESID && !(G3_ABLATE & 2)) resid_fetch(std::integral_constant<int, g3_block(5)>{});
        mfma_step_with(f1, [&](auto m) __attribute__((always_inline)) { dma_fill_w(ntw + G3_BK, m); });
        if constexpr (F1_EARLY) read_frag(f1, I1{});
        first_step(std::true_type{}, p.bias + n0, nta + 2 * G3_BK);
        have_prev = true;
        pend = steps_1_2(I1{}, I0{}, ctile, nullptr);      // (block 0 leaves)
    }
    // ---- the last output tile: its last k-step, then its six blocks at once (the requests issued behind the last tile must
    // not outlive the workgroup: vmcnt(0))
#ifdef BERT_HIP_TIMELINE
    clocked_barrier(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *o = g3_clock + ((size_t)(blockIdx.x & 255) * 8 + wave) * 16;
        for (int k = 0; k < 4; ++k)
            for (int q = 0; q < 4; ++q) o[k * 4 + q] = clk[k][q];
    }
#else
    g3_barrier<0>(f1);
#endif
    mfma_step_with(f1, no_fill);
    static_for<6>([&](auto m) __attribute__((always_inline)) { convert_block(m); });
    ctile = (const char *)p.C + ((size_t)m0 * N + n0) * 2;
    static_for<6>([&](auto k) __attribute__((always_inline)) {
        using B = std::integral_constant<int, g3_block(decltype(k)::value)>;
        static_for<8>([&](auto r) __attribute__((always_inline)) { gelu_reg(B{}, r); });
        stage_write(B{});
        stage_read();
        store_rows(ctile, B{});
    });
}

bool gemm192_supported(const GemmWeight &W, int M_pad) {
    return W.type == GW_F16 && W.w16 != nullptr && W.N % G3_BN == 0 && W.K % G3_BK == 0 && W.K >= 12 * G3_BK &&      // (six reduction tiles of blocks leaving + six of residual requests)
          
           M_pad % G3_BM == 0 && M_pad > 0 && (size_t)M_pad * W.N * 2 < ((size_t)1 << 31);
}

void launch_gemm192(const GemmWeight &W, const half_t *A, const float *bias, const half_t *resid, half_t *C, int M_pad,
                    int epilogue, hipStream_t stream) {
    Gemm192Args a;
    a.A = A; a.w16 = W.w16; a.bias = bias; a.resid = resid; a.C = C;
    a.N = W.N; a.K = W.K; a.n_tiles_n = W.N / G3_BN;
    a.n_tiles = a.n_tiles_n * (M_pad / G3_BM);
    // feature groups as in gemm256.hip: only where W (N x K f16) overflows an XCD's L2 share and reading the activations twice is
    // the cheaper side
    a.n_groups = 1;
    if ((size_t)W.N * W.K * 2 > (size_t)3 << 20 && W.N >= 4 * W.K && a.n_tiles_n % 2 == 0 && M_pad / G3_BM >= 64) a.n_groups = 2;
    if (const char *g = getenv("BERT_HIP_G3_GROUPS")) {      // (tuning) feature groups for matrices beyond 3 MiB
        const int v = atoi(g);
        if ((v == 1 || v == 2 || v == 4) && a.n_tiles_n % v == 0 && (size_t)W.N * W.K * 2 > (size_t)3 << 20 && M_pad / G3_BM >= 64) a.n_groups = v;
    }
    static int n_cu[MAX_HIP_DEVICES] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < MAX_HIP_DEVICES && !n_cu[dev]) {
        hipDeviceProp_t prop;
        n_cu[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8 ? prop.multiProcessorCount / 8 * 8 : 256;
    }
    const int cus = dev >= 0 && dev < MAX_HIP_DEVICES ? n_cu[dev] : 256;
    const int grid = std::min(cus, (a.n_tiles + 7) / 8 * 8);
    const size_t lds = G3_LDS_BYTES;
    static DeviceFlags configured[3];
    auto go = [&](auto kernel, int e) {
        configure_once(configured[e], [&] { (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES); });
        BERT_LAUNCH(kernel, dim3(grid), dim3(512), lds, stream, a);
    };
    switch (epilogue) {
        case EPI_BIAS: go(gemm192_kernel<EPI_BIAS>, 0); break;
        case EPI_BIAS_GELU: go(gemm192_kernel<EPI_BIAS_GELU>, 1); break;
        default: go(gemm192_kernel<EPI_BIAS_RESID>, 2); break;
    }
#ifdef BERT_HIP_TIMELINE
    {   // the phase clock of launches 40..43 (one of each shape of a bert-base layer), summed over the waves of a few workgroups
        static int shots = 0;
        if (shots >= 40 && shots < 44) {
            (void)hipStreamSynchronize(stream);
            static unsigned long long h[256 * 8 * 16];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g3_clock), sizeof(h));
            for (int wg : {0, 100, 255}) {
                unsigned long long sum[4][4] = {};
                for (int w = 0; w < 8; ++w)
                    for (int k = 0; k < 16; ++k) sum[k / 4][k % 4] += h[((size_t)wg * 8 + w) * 16 + k];
                fprintf(stderr, "gemm192 clock N=%d K=%d epi=%d wg %3d:", W.N, W.K, epilogue, wg);
                static const char *kinds[4] = {"plain", "leave", "resid", "first"};
                for (int k = 0; k < 4; ++k)
                    if (sum[k][3]) fprintf(stderr, "  %s x%llu: work %.0f vmwait %.0f barrier %.0f", kinds[k], sum[k][3] / 8, (double)sum[k][0] / sum[k][3], (double)sum[k][1] / sum[k][3], (double)sum[k][2] / sum[k][3]);
                fprintf(stderr, "\n");
            }
        }
        ++shots;
    }
#endif
}

}  // namespace bert_hip
