// layer_tail.hip — everything of an encoder layer after the attention in ONE kernel, two SPECIALIST waves per SIMD
// (gfx950, f16 weights, H = 256 / 384):
//     y     = LayerNorm(ctx Wo^T + bo + x) * g1 + be1                     (reference bert.cpp:859-875)
//     x_out = LayerNorm(gelu(y W1^T + b1) W2^T + b2 + y) * g2 + be2       (reference bert.cpp:878-901)
//
// History: the round-1/2 form of this kernel gave ONE wave per SIMD 32 tokens x all features; its steady state carried 5.7
// non-MFMA instructions per MFMA (tools/isa_gaps.py: 1.9 VALU + 0.7 transcendental + 0.7 accvgpr moves + 1.4 LDS + 0.25 DMA +
// 0.5 SALU) against the <= 5 a lone wave can hide per v_mfma_f32_32x32x16 gap, and the wave issued all of it in order: 75
// cycles per MFMA instead of 32.  A round-2 experiment put two SYMMETRIC waves on a SIMD (both up-project, split by K, both
// down-project): per chunk a 32 x 32 f32 partial block and half a GELU'ed chunk crossed LDS in both directions with two
// dependent hand-overs, a quarter of its time, and it ended level.  Both are gone; this is the only form.
//
// Here a token block of 32 belongs to a pair of waves with DIFFERENT jobs (workgroup = 128 tokens = 4 pairs, 512 threads):
//   * wave U (up):   holds the LayerNorm'ed y of its 32 tokens as 8 NT B fragments in REGISTERS for the whole feed-forward
//                    (no y staging in LDS, no y re-reads), multiplies the up-projection tiles ([64 rows x 128 k], 16 MFMAs per
//                    tile into one of TWO sets of 32 accumulators), and runs the GELU of the previous chunk as VALU filler
//                    between those MFMAs, in packed f16, straight into the accumulator layout that IS the down-projection's
//                    B fragment (GemmWeight::w16p);
//   * wave D (down): holds ALL 4 NT output accumulator blocks of the same 32 tokens (192 registers at H = 384, initial value
//                    y + b2: the residual), multiplies the down-projection tiles ([128 rows x 64 k], 16 MFMAs per tile) with
//                    the GELU'ed chunk U published — four ds_write_b128 by U, four ds_read_b128 by D per chunk of 64
//                    intermediate features, ONE direction, riding on the tile barriers (no extra synchronisation);
//                    it issues no VALU work at all in the loop, so its stream is MFMAs + fragment reads + its share of the DMA.
// The two streams are decoupled by two chunks (U: UP(c) + gelu(c-1), D: DOWN(c-2)); one tile interval = one UP tile for U
// and one DOWN tile for D = 32 MFMAs per SIMD between two workgroup barriers (layer_tail: 16), and what one wave issues
// beside its MFMAs (fragment reads, GELU, DMA requests, waits) sits under the other's MFMAs.
// Out-projection: both waves, split by output features (2 NT blocks each, x + bo as the initial value), two tiles per
// interval; LayerNorm 1: row sums cross LDS (one float pair per token), then each wave's half of y crosses once (4 NT
// fragments each way) — U needs all of y as B fragments, D all of y as its accumulators' initial value.
// LayerNorm 2 is local to D (it owns whole rows); both waves store the rows.
//
// LDS: two rings of 3 x 16 KiB (UP tiles / DOWN tiles; the out-projection uses both), 4 x 2 x 4 KiB GELU'ed chunks,
// parameters.  Registers (H = 384): U 96 (y) + 64 (two sets of up-projection accumulators) + 48 (fragments) + 16 (chunk
// being GELU'ed); D 192 + 32 (fragments) + 16 (chunk).
#include "tile_stream.h"

namespace bert_hip {

namespace {

constexpr int LT_TILE = 16384;
// LT_ABLATE (tuning builds only, results are wrong): bit 0 no weight DMA, bit 1 no GELU arithmetic, bit 2 no fragment reads,
// bit 3 no MFMAs, bit 4 no chunk publish / fetch, bit 5 LayerNorm 1 without statistics and parameter reads, bit 6 no out-projection
// fragment reads / MFMAs, bit 7 LayerNorm 2 without parameter reads
#ifndef LT_ABLATE
#define LT_ABLATE 0
#endif
// priority of the U waves in the feed-forward loop (D stays at 0)
#ifndef LT_UPRIO
#define LT_UPRIO 1
#endif
#ifndef LT_GELU_BATCH
#define LT_GELU_BATCH 0
#endif
// D's requests of an interval: 0 = all eight pieces behind its first MFMA group, 1 = the up-projection tile's behind the first,
// the down-projection tile's behind the second
#ifndef LT_DMA_SPLIT
#define LT_DMA_SPLIT 1
#endif
#ifndef LT_DPRIO
#define LT_DPRIO 0
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Tuning aid (-DBERT_HIP_TIMELINE, `make timeline`): shader-clock stamps of pair 0 of every workgroup: [0, 128) U behind
// the barrier of interval i, [128, 256) U in front of it, [256, 384) D in front of it, [384, 512) phases outside the loops
#ifdef BERT_HIP_TIMELINE
static __device__ unsigned long long g_tl_tail[256 * 512];
#define LT_STAMP(sel, i) do { const int tl_i = (i); if ((sel) && tl_i < 512) g_tl_tail[(blockIdx.x & 255) * 512 + tl_i] = __builtin_readcyclecounter(); } while (0)
#ifdef LT_FINE
#define LT_STAMP_FINE(sel, i) LT_STAMP(sel, i)
#else
#define LT_STAMP_FINE(sel, i) do { } while (0)
#endif
#else
#define LT_STAMP_FINE(sel, i) do { } while (0)
#define LT_STAMP(sel, i) do { } while (0)
#endif

struct TailArgs {
    const half_t *ctx, *x;            // [T_pad][H]
    const half_t *wo;                 // [H_pad][H] f16
    const half_t *w1p;                // [I_pad][H] f16, k order permuted inside groups of 16 (GemmWeight::w16p)
    const half_t *w2p;                // [H_pad][I] f16, same
    // q4_0 / q4_1 weights (kernels.h GemmWeight: nibble plane + scale plane per matrix, tile-contiguous): the matrices stay
    // 4-bit in HBM and L2 and are expanded into the ring slots on chip
    const uint4 *wo_qs, *w1_qs, *w2_qs;
    const void *wo_sc, *w1_sc, *w2_sc;
    const float *bo, *g1, *be1, *b1, *b2, *g2, *be2;
    half_t *out;                      // [T_pad][H]
    int I;
};

// retire all but the newest N hand-issued LDS reads; the four fragments named are the ones whose MFMAs follow
template <int N>
__device__ __forceinline__ void wait_frags(f16x8 (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory");
}
// a hand-read register handed to its users (behind the wait that retired the read)
template <class V>
__device__ __forceinline__ void landed(V &v) { asm volatile("" : "+v"(v)); }
// closing barrier of a tile interval: this wave's DMA pieces of the NEXT tile have landed (all but the newest VM), its LDS
// reads and writes are complete
template <int VM>
__device__ __forceinline__ void close_interval() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" : : "n"(VM) : "memory");
}
template <int VM>
__device__ __forceinline__ void close_interval(f16x8 (&f)[4]) {
    asm volatile("s_waitcnt vmcnt(%4) lgkmcnt(0)\n\ts_barrier" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(VM) : "memory");
}
template <int OFF>
__device__ __forceinline__ f32x2 lds_read_b64_h(unsigned addr) {
    f32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_read_f32x4_h(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ f16x8 frag_read(unsigned addr) {
    if (LT_ABLATE & 4) { f16x8 z = (f16x8)(_Float16)0.f; asm volatile("" : "+v"(z) : "v"(addr)); return z; }
    return lds_read_b128_u<OFF>(addr);
}

// LDS-DMA by hand (global_load_lds_dwordx4, 1 KiB per wave-instruction): scalar base + 32-bit lane offset, so no 64-bit
// address arithmetic on the vector unit and one VGPR per distinct lane pattern.  M0 = LDS address of the wave's first piece,
// set once per tile; the instruction offset (which moves BOTH the global and the LDS address) selects the piece, the
// scalar base compensates.  The compiler never touches M0 in this kernel (no LDS-DMA builtin, no indirect indexing).
__device__ __forceinline__ void dma_m0(unsigned lds) {
    if (LT_ABLATE & 1) return;
    asm volatile("s_mov_b32 m0, %0" : : "s"(lds) : "memory");
}
template <int PIECE>
__device__ __forceinline__ void dma_piece(const char *gbase, unsigned voff) {
    if (LT_ABLATE & 1) return;
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(voff), "s"(gbase - PIECE * 1024), "n"(PIECE * 1024) : "memory");
}

__device__ __forceinline__ f32x16 mfma16(const f16x8 &a, const f16x8 &b, const f32x16 &c) {
    if (LT_ABLATE & 8) { f32x16 r = c; asm volatile("" : "+v"(r[0]) : "v"(a), "v"(b)); return r; }
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

}  // namespace

// (the kernel's body as a device function of (arguments, the workgroup's LDS, 128-token block index): model_kernel.hip runs it as
// one phase of a launch that carries a window through all layers)
// RAGGED: the block is rows tok0 .. tok0 + rows - 1 only (a window of whole sentences with fewer than 128 tokens; the rows behind
// belong to another workgroup): the loads of the missing rows repeat the last one, their stores are skipped.
template <int NT, int WT, bool RAGGED = false>
__device__ __forceinline__ void layer_tail_body(const TailArgs &a, char *smem, const int tok0, const int rows, const int tid) {
    constexpr bool Q4 = WT != GW_F16;
    constexpr int VMQ = Q4 ? 63 : 0;                          // (q4: no LDS-DMA in flight, nothing for a barrier to wait for)
    constexpr int H = 128 * NT, NBH = 2 * NT, NB = 4 * NT, NQ = 8 * NT, NYH = 4 * NT;
    constexpr int P = NT * NT;                                // out-projection intervals (two [128 x 64] tiles each)
    constexpr int NG = NT;                                    // intervals over which the GELU of a chunk is spread
    constexpr int LAGT = NT + NG;                             // D runs this many intervals behind U
    const int I = a.I, NC = I / 64;                           // chunks of 64 intermediate features (even, >= 2)
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = wave & 3, role = wave >> 2;                 // token block; 0 = U (up-projection + GELU), 1 = D (down-projection)
    const int l31 = lane & 31, hi = lane >> 5;
    const int tok_w = tok0 + t * 32;                          // first token of this pair
    const int row_l = RAGGED ? tok0 + min(t * 32 + l31, rows - 1) : tok_w + l31;      // the lane's row of x / ctx

    char *ringU = smem;                                       // 3 x 16 KiB
    char *ringD = smem + 3 * LT_TILE;                         // 3 x 16 KiB
    char *G = smem + 6 * LT_TILE;                             // [4 pairs][2][4 KiB] GELU'ed chunks
    float *ST = (float *)(G + 32768);                         // 8 x 32 x {sum, sum of squares}
    float *cbo = ST + 8 * 64;
    float *cg1 = cbo + H, *cbe1 = cg1 + H, *cb2 = cbe1 + H, *cg2 = cb2 + H, *cbe2 = cg2 + H, *cb1 = cbe2 + H;   // cb1: I + 128 floats

    {   // parameters -> LDS: every load of a thread in flight together (one round trip, not one per loop iteration)
        constexpr int NPB = 4;                                // b1: up to 512 * 4 * NPB floats
        const float *const src[6] = {a.bo, a.g1, a.be1, a.b2, a.g2, a.be2};
        float pv[6];
        f32x4 bq[NPB];
#pragma unroll
        for (int k = 0; k < 6; ++k) pv[k] = tid < H ? src[k][tid] : 0.f;
#pragma unroll
        for (int k = 0; k < NPB; ++k) {
            const int i4 = (k * 512 + tid) * 4;
            bq[k] = i4 < I ? *(const f32x4 *)(a.b1 + i4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (tid < H) {
#pragma unroll
            for (int k = 0; k < 6; ++k) cbo[k * H + tid] = pv[k];
        }
#pragma unroll
        for (int k = 0; k < NPB; ++k) {
            const int i4 = (k * 512 + tid) * 4;
            if (i4 < I + 128) *(f32x4 *)(cb1 + i4) = bq[k];
        }
    }
    // ---- out-projection accumulators start from x + bo: own block b = n3 * 2 + obp holds features
    // n3*128 + role*64 + obp*32 + 8 (r >> 2) + 4 hi + (r & 3) of token l31 in register r.  x now (one round trip), the bias
    // from LDS behind the first barrier: requesting both from HBM at once needs 144 registers next to the 96 of the context.
    f32x16 accp[NBH];
    f16x4 xv[NBH][4];
    {
        const half_t *xr = a.x + (size_t)row_l * H + role * 64 + 4 * hi;
#pragma unroll
        for (int b = 0; b < NBH; ++b)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) xv[b][gq] = *(const f16x4 *)(xr + (b >> 1) * 128 + (b & 1) * 32 + 8 * gq);
    }
    // attention context of the pair's tokens as B fragments, straight into registers (k order as stored)
    f16x8 bf[NQ];
    {
        const half_t *cw = a.ctx + (size_t)row_l * H + 8 * hi;
#pragma unroll
        for (int q = 0; q < NQ; ++q) bf[q] = *(const f16x8 *)(cw + 16 * q);
    }

    // ---- weight-tile stream.  A tile is 16 pieces of 1 KiB; the four waves of a role move the tiles of "their" ring, four
    // pieces each: [128 rows x 64 k] tiles (128-byte rows): piece i of wave t covers rows (t*4+i)*8 .. +8, 16-byte chunk c of a
    // row at chunk c ^ ((row >> 1) & 7); [64 rows x 128 k] tiles (256-byte rows): rows (t*4+i)*4 .. +4, chunk c at c ^ (row & 15).
    // Lane offsets in few registers: the row part of i goes into the scalar base, the swizzle alternates between two values
    // (even / odd i) resp. is the one of piece 0 with i XORed into bits 6..7.
    // offR[0/1]: even / odd pieces of a [128 x 64] tile with row pitch `pitch` halfs; offU: piece 0 of a [64 x 128] tile
    // (piece i: ^ (i << 6); the row part is a multiple of 256 bytes).  Computed per phase from an opaque copy of the lane id:
    // values that live across phases get spilled for their whole life.
    auto rows_offset = [&](int pitch) __attribute__((always_inline)) {        // even pieces; odd pieces: ^ 64 (the row part is a multiple of 128)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int ch0 = (ln & 7) ^ (ln >> 4), row = t * 32 + (ln >> 3);
        return (unsigned)(row * pitch * 2 + ch0 * 16);
    };
    const half_t *const wo = a.wo, *const w1p = a.w1p, *const w2p = a.w2p;
    auto dma128 = [&](const half_t *base, unsigned oe, int row_bytes, unsigned tile) __attribute__((always_inline)) {
        const char *b = (const char *)base;
        dma_m0(tile + (unsigned)(t * 4096));
        dma_piece<0>(b, oe);
        dma_piece<1>(b + (size_t)8 * row_bytes, oe ^ 64u);
        dma_piece<2>(b + (size_t)16 * row_bytes, oe);
        dma_piece<3>(b + (size_t)24 * row_bytes, oe ^ 64u);
    };
    const unsigned ldsU = lds_addr(ringU), ldsD = lds_addr(ringD);
    const unsigned offH = rows_offset(H);
    auto up_offset = [&]() __attribute__((always_inline)) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        return (unsigned)((t * 16 + (ln >> 4)) * H * 2 + (((ln & 15) ^ (ln >> 4)) * 16));
    };
    auto dma_up_f16 = [&](int c, int j, unsigned offU, unsigned tile) __attribute__((always_inline)) {
        const char *b = (const char *)(w1p + (size_t)c * 64 * H + j * 128);
        dma_m0(tile + (unsigned)(t * 4096));
        dma_piece<0>(b, offU);
        dma_piece<1>(b + (size_t)4 * H * 2, offU ^ 64u);
        dma_piece<2>(b + (size_t)8 * H * 2, offU ^ 128u);
        dma_piece<3>(b + (size_t)12 * H * 2, offU ^ 192u);
    };

    // ---- q4 weights: what the f16 form requests by LDS-DMA during interval i (the tiles of interval i + 2), this form LOADS as
    // raw blocks into registers during interval i and EXPANDS at the start of interval i + 1 into the same ring slot (free
    // since the barrier before; readable behind the next one).  A wave has two raw slots (A: the [128 x 64] tiles of "its"
    // ring — out-projection, down-projection; B: the up-projection tiles, D waves only); a thread expands one block per slot.
    // What is pending at the start of an interval is known at compile time (the requests of the interval before).
    [[maybe_unused]] RawBlock rawA, rawB;
    auto role_thread = [&]() __attribute__((always_inline)) {  // 0 .. 255 inside the role (made where needed, not kept)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        return t * 64 + ln;
    };
    auto q4_request_rows = [&](const uint4 *qs, const void *sc, int tile_index) __attribute__((always_inline)) {
        if constexpr (Q4) rawA = q4_load_block<WT>(qs, sc, (size_t)tile_index * 256 + role_thread());
    };
    auto q4_request_up = [&](int c, int j) __attribute__((always_inline)) {
        if constexpr (Q4) {
            // [64 rows x 128 k] of chunk c, k-tile j: rows (c & 1) * 64 .. of q-tile row c >> 1, k-tiles 2 j and 2 j + 1
            const int tidR = role_thread(), b_hi = tidR >> 7, rem = tidR & 127;
            const size_t idx = ((size_t)((c >> 1) * (H / 64) + 2 * j + b_hi) * 128 + (c & 1) * 64) * 2 + rem;
            rawB = q4_load_block<WT>(a.w1_qs, a.w1_sc, idx);
        }
    };
    // slot A -> [128 x 64] tile at `tile` (PERM: the k order of GemmWeight::w16p); slot B -> [64 x 128] tile
    auto q4_expand_rows = [&](auto perm_tag, char *tile) __attribute__((always_inline)) {
        if constexpr (Q4) {
            // chunk 4 blk + k of row `row` sits at chunk (4 blk + k) ^ ((row >> 1) & 7) = (4 blk ^ ..) ^ k
            const int tidR = role_thread(), row = tidR >> 1, blk = tidR & 1;
            const int o = row * 128 + (((4 * blk) ^ ((row >> 1) & 7)) << 4);
            q4_expand_block<WT, decltype(perm_tag)::value>(rawA, [&](int k) __attribute__((always_inline)) { return tile + (o ^ (k << 4)); });
        }
    };
    auto q4_expand_up = [&](char *tile) __attribute__((always_inline)) {
        if constexpr (Q4) {
            const int tidR = role_thread(), b_hi = tidR >> 7, rem = tidR & 127, r = rem >> 1, blk = rem & 1;
            const int o = r * 256 + (((4 * (2 * b_hi + blk)) ^ (r & 15)) << 4);
            q4_expand_block<WT, true>(rawB, [&](int k) __attribute__((always_inline)) { return tile + (o ^ (k << 4)); });
        }
    };
    auto dma_proj = [&](int n3, int kt, unsigned tile) __attribute__((always_inline)) {
        if constexpr (Q4) q4_request_rows(a.wo_qs, a.wo_sc, n3 * (H / 64) + kt);
        else dma128(wo + (size_t)n3 * 128 * H + kt * 64, offH, H * 2, tile);
    };
    auto dma_up = [&](int c, int j, unsigned offU, unsigned tile) __attribute__((always_inline)) {
        if constexpr (Q4) q4_request_up(c, j);
        else dma_up_f16(c, j, offU, tile);
    };

    // ---- fragment addresses: k-step kk of a tile = the address of k-step 0 with kk XORed into bits 5.. (the slot offsets are
    // multiples of 8 KiB, so the XOR commutes with adding them)
    const unsigned aR = lds_addr(ringU) + (unsigned)off64(l31, hi);                      // [128 x 64]: + ob * 4096
    const unsigned aUp = lds_addr(ringU) + (unsigned)(l31 * 256 + ((hi ^ (l31 & 15)) << 4));   // [64 x 128]: + fb * 8192

    int slot = 0;                                             // ring slot of the current interval (both rings)
    [[maybe_unused]] int tl = 1;
    [[maybe_unused]] const bool tlU = tid == 0, tlD = tid == 256;
    LT_STAMP(tlU, 0);
    // (stamps: in front of the closing barrier per role, behind it for U)
    auto pre_close = [&]() __attribute__((always_inline)) { LT_STAMP_FINE(tlU, 128 + tl); LT_STAMP_FINE(tlD, 256 + tl); };
    auto next_slot = [&]() __attribute__((always_inline)) { slot = slot == 2 ? 0 : slot + 1; LT_STAMP(tlU, tl); ++tl; };
    auto slot_plus2 = [&]() __attribute__((always_inline)) { return slot == 0 ? 2 : slot - 1; };
    [[maybe_unused]] auto slot_plus1 = [&]() __attribute__((always_inline)) { return slot == 2 ? 0 : slot + 1; };
    using PLAIN = std::false_type; using PERMUTED = std::true_type;

    // ================================ out-projection ================================
    // interval p: tiles (n3, kt = 2 q) in ringU[slot] and (n3, 2 q + 1) in ringD[slot], n3 = p / NT, q = p % NT; a wave
    // multiplies row blocks 2 role, 2 role + 1 of both.  U waves request the ringU tiles, D waves the ringD tiles.
    auto proj_request = [&](int p, int s) __attribute__((always_inline)) {
        const int n3 = p / NT, q = p % NT;
        if (role == 0) dma_proj(n3, 2 * q, ldsU + s * LT_TILE);
        else dma_proj(n3, 2 * q + 1, ldsD + s * LT_TILE);
    };
    proj_request(0, 0);
    q4_expand_rows(PLAIN{}, role ? ringD : ringU);            // (q4: interval 0 expanded here, interval 1 at the start of interval 0)
    proj_request(1, 1);
    LT_STAMP(tlU, 384);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // parameters, x, ctx fragments, intervals 0 and 1
    LT_STAMP(tlU, 385);
#pragma unroll
    for (int b = 0; b < NBH; ++b)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 bv = *(const f32x4 *)(cbo + (b >> 1) * 128 + role * 64 + (b & 1) * 32 + 8 * gq + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) accp[b][4 * gq + e] = (float)xv[b][gq][e] + bv[e];
        }
#pragma unroll
    for (int b = 0; b < NBH; ++b) asm volatile("" : "+v"(accp[b]));     // (made here: left alone the compiler keeps x alive and converts it block by block)
    __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0), visible to the compiler (see the note in U's branch)
    static_for<P>([&](auto p_tag) __attribute__((always_inline)) {
        constexpr int p = decltype(p_tag)::value, n3 = p / NT, q = p % NT, p2 = p + 2;
        const unsigned bA = aR + (unsigned)(slot * LT_TILE + role * 8192), bB = bA + 3 * LT_TILE;
        f16x8 Fa[4], Fb[4];
        // group = two k-steps x the wave's two row blocks
        auto rd = [&](unsigned base, int g2, f16x8 (&F)[4]) __attribute__((always_inline)) {
            const unsigned a0 = base ^ (unsigned)((2 * g2) << 5), a1 = base ^ (unsigned)((2 * g2 + 1) << 5);
            F[0] = frag_read<0>(a0); F[1] = frag_read<4096>(a0);
            F[2] = frag_read<0>(a1); F[3] = frag_read<4096>(a1);
        };
        auto mm = [&](f16x8 (&F)[4], int ks) __attribute__((always_inline)) {      // ks = first ctx k-step of the group
            accp[n3 * 2 + 0] = mfma16(F[0], bf[ks], accp[n3 * 2 + 0]);
            accp[n3 * 2 + 1] = mfma16(F[1], bf[ks], accp[n3 * 2 + 1]);
            accp[n3 * 2 + 0] = mfma16(F[2], bf[ks + 1], accp[n3 * 2 + 0]);
            accp[n3 * 2 + 1] = mfma16(F[3], bf[ks + 1], accp[n3 * 2 + 1]);
        };
        // q4: the blocks requested one interval ago are expanded (into the slot of interval p + 1) in the MIDDLE of this
        // interval and the next request follows at once: a request has a whole interval to land, and the expansion's
        // VALU work sits between this wave's MFMA groups, under the partner wave's MFMAs.  (Four LDS stores join the queue
        // of hand-issued reads: the wait behind them counts them.)
        auto q4_turn = [&]() __attribute__((always_inline)) {
            if constexpr (p + 1 < P) q4_expand_rows(PLAIN{}, (role ? ringD : ringU) + slot_plus1() * LT_TILE);
            else if (role == 1) q4_expand_up(ringU + slot_plus1() * LT_TILE);
            const int s2 = slot_plus2();
            if constexpr (p2 < P) proj_request(p2, s2);
            else if (role == 1) dma_up(0, p2 - P, up_offset(), ldsU + s2 * LT_TILE);
        };
        constexpr int QW = Q4 && p + 1 < P ? 4 : 0;           // (the two intervals in which only D expands: the strict count for both roles)
        if constexpr (LT_ABLATE & 64) {
            if constexpr (Q4) q4_turn();
            else {
                const int s2 = slot_plus2();
                if constexpr (p2 < P) proj_request(p2, s2);
                else if (role == 1) dma_up(0, p2 - P, up_offset(), ldsU + s2 * LT_TILE);
            }
            pre_close();
            if (p2 < P || role == 1) close_interval<Q4 ? 63 : 4>(); else close_interval<VMQ>();
            next_slot();
            return;
        }
        rd(bA, 0, Fa);
        rd(bA, 1, Fb);
        if constexpr (!Q4) {   // requests for interval p + 2 (out-projection, or U's first two up-projection tiles)
            const int s2 = slot_plus2();
            if constexpr (p2 < P) proj_request(p2, s2);
            else if (role == 1) dma_up(0, p2 - P, up_offset(), ldsU + s2 * LT_TILE);
        }
        wait_frags<4>(Fa);
        mm(Fa, 8 * q + 0);
        __builtin_amdgcn_sched_barrier(0);
        rd(bB, 0, Fa);
        wait_frags<4>(Fb);
        mm(Fb, 8 * q + 2);
        __builtin_amdgcn_sched_barrier(0);
        rd(bB, 1, Fb);
        if constexpr (Q4) { q4_turn(); __builtin_amdgcn_sched_barrier(0); }
        wait_frags<4 + QW>(Fa);
        mm(Fa, 8 * q + 4);
        __builtin_amdgcn_sched_barrier(0);
        wait_frags<0>(Fb);
        mm(Fb, 8 * q + 6);
        __builtin_amdgcn_sched_barrier(0);
        pre_close();
        if (p2 < P || role == 1) close_interval<Q4 ? 63 : 4>(); else close_interval<VMQ>();
        next_slot();
    });

    if (role == 1) q4_expand_up(ringU + slot_plus1() * LT_TILE);           // (q4: the second up-projection tile)
    LT_STAMP(tlU, 386); LT_STAMP(tlD, 396);
    // ================================ LayerNorm 1 ================================
    // own half of the features summed here, the partner's through ST; then each role normalises its half into f16 fragments:
    // yh[4 n3 + m] is k-step 8 n3 + 4 role + m of the up-projection (registers 8 s .. 8 s + 7 of own block b = 2 n3 + (m >> 1),
    // s = m & 1).  D's half crosses to U through ringD (U needs all of y as B fragments); U's half crosses to D at the END
    // (D adds the residual to the accumulator blocks of U's features in front of LayerNorm 2), when LDS and D's registers are free.
    float ln_rstd, ln_nmr;
    {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int b = 0; b < ((LT_ABLATE & 32) ? 0 : NBH); ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s1 += accp[b][r]; s2 = __builtin_fmaf(accp[b][r], accp[b][r], s2); }
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (hi == 0) *(f32x2 *)(ST + wave * 64 + l31 * 2) = f32x2{s1, s2};
        LT_STAMP(tlU, 387); LT_STAMP(tlD, 397);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        LT_STAMP(tlU, 388);
        const f32x2 o = *(const f32x2 *)(ST + (wave ^ 4) * 64 + l31 * 2);
        s1 += o[0]; s2 += o[1];
        layernorm_scale(s1, s2, 1.0f / H, ln_rstd, ln_nmr);
    }
    // own block b -> fragments y0 (registers 0..7), y1 (8..15); the eight parameter reads of a block are in flight together
    auto normalise_block = [&](auto b_tag, f16x8 &y0, f16x8 &y1) __attribute__((always_inline)) {
        constexpr int b = decltype(b_tag)::value;
        const int f0 = (b >> 1) * 128 + role * 64 + (b & 1) * 32 + 4 * hi;
        if constexpr (LT_ABLATE & 32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { y0[e] = (_Float16)accp[b][e]; y1[e] = (_Float16)accp[b][8 + e]; }
            return;
        }
        f32x4 gv[4], bv[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) { gv[gq] = *(const f32x4 *)(cg1 + f0 + 8 * gq); bv[gq] = *(const f32x4 *)(cbe1 + f0 + 8 * gq); }
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // g * ((v - mean) * rstd) + b  =  v * (g * rstd) + (g * (-mean * rstd) + b), rounded to f32 and THEN to f16 (never the
                // single-rounding v_fma_mixlo_f16: skinny.hip's LayerNorm must give the same bits, whatever the compiler prefers there)
                const _Float16 y = (_Float16)rounded_f32(__builtin_fmaf(accp[b][4 * gq + e], gv[gq][e] * ln_rstd, __builtin_fmaf(gv[gq][e], ln_nmr, bv[gq][e])));
                if (gq < 2) y0[4 * gq + e] = y; else y1[4 * (gq - 2) + e] = y;
            }
        asm volatile("" ::: "memory");                        // (one block's parameters at a time: unfenced, the loads of all blocks are hoisted to the front)
    };
    // the exchange areas: pair t owns NYH KiB, fragment m of lane l at m KiB + 16 l
    char *const xch = ringD + t * (NYH * 1024) + lane * 16;    // LayerNorm 1: D's half (ringD is idle until D's first requests)
    LT_STAMP(tlU, 389); LT_STAMP(tlD, 399);

    // ================================ rows -> HBM (both waves of the pair, alternate 1 KiB pieces) ================================
    // (called at the end of EACH role's branch, which then returns: with a common tail behind the branches the register
    // allocator carries U's fragments "through" D's branch — stores in front of D's loop, dead reloads behind it)
    typedef unsigned store_u32x4 __attribute__((ext_vector_type(4)));
    auto store_rows = [&]() __attribute__((always_inline)) {
        LT_STAMP(tlU, 394);
        const char *S = smem + t * (64 * H);
        int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane));                     // (made here, every time: as a phase of model_kernel.hip the address arithmetic below is loop-invariant)
        // 16-byte unit (token, 8-feature chunk c8) of the output = bytes h2*8.. of the fragments (q, token) and
        // (q, token + 32), q = c8 >> 1, h2 = c8 & 1
        half_t *ow = a.out + (size_t)tok_w * H;
        // RAGGED: a buffer of the pair's rows that exist — the hardware drops the stores behind its end (a branch around each
        // store costs the combined translation unit 450 bytes of scratch)
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t out_rows =
            __builtin_amdgcn_make_buffer_rsrc(ow, 0, max(min(rows - t * 32, 32), 0) * H * 2, 0x00020000);
#pragma unroll
        for (int st2 = 0; st2 < H / 32; ++st2) {
            const int st = 2 * st2 + role;
            const int v = st * 64 + lane, tok = v / (H / 8), c8 = v - tok * (H / 8), q = c8 >> 1, h2 = c8 & 1;
            const char *row = S + q * 1024 + h2 * 8;
            const f16x4 lo = *(const f16x4 *)(row + ((tok + 2 * q) & 63) * 16);
            const f16x4 hi4 = *(const f16x4 *)(row + ((tok + 32 + 2 * q) & 63) * 16);
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi4[e]; }
            if constexpr (RAGGED) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(store_u32x4, o), out_rows, (tok * H + c8 * 8) * 2, 0, 0);
            else *(f16x8 *)(ow + (size_t)tok * H + c8 * 8) = o;
        }
#ifdef BERT_HIP_TIMELINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LT_STAMP(tlU, 395);
#endif
    };

    if (role == 0) {
        // =====================================================================================================
        // U: y fragments of all k-steps, up-projection, GELU
        // =====================================================================================================
        f16x8 Y[NQ];
        static_for<NBH>([&](auto b_tag) __attribute__((always_inline)) {
            constexpr int b = decltype(b_tag)::value;             // own block b = k-steps 8 (b >> 1) + 2 (b & 1), + 1
            normalise_block(b_tag, Y[8 * (b >> 1) + 2 * (b & 1)], Y[8 * (b >> 1) + 2 * (b & 1) + 1]);
        });
        LT_STAMP(tlU, 390);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // [B2] D's half is in LDS
        LT_STAMP(tlU, 391);
#pragma unroll
        for (int m = 0; m < NYH; ++m) Y[8 * (m >> 2) + 4 + (m & 3)] = *(const f16x8 *)(xch + m * 1024);
        // (a wait the COMPILER sees: its own bookkeeping of these loads would otherwise stay "pending" around the loop's back
        // edge — the waits inside asm statements are invisible to it — and it would add lgkmcnt waits in front of every MFMA
        // that reads y, draining the hand-issued fragment reads each time)
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0)
        LT_STAMP(tlU, 392);
        // two sets of up-projection accumulators (chunk parity), initial value = the chunk's bias
        f32x16 accU[2][2];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int r = 0; r < 16; ++r)          // register r of block fb = feature 32 fb + (r & 3) + 8 (r >> 2) + 4 hi of the chunk
                    accU[pp][fb][r] = cb1[pp * 64 + 32 * fb + (r & 3) + 8 * (r >> 2) + 4 * hi];
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0), visible to the compiler (see above)
        f16x8 Fc[4];                                          // the fragment group carried across the barrier
#pragma unroll
        for (int j = 0; j < 4; ++j) Fc[j] = (f16x8)(_Float16)0;
        const unsigned aB = lds_addr(cb1) + hi * 16;          // up-projection biases of a lane's features: + chunk * 256 B
        char *const Gw = G + t * 8192 + lane * 16;            // + parity * 4096 + fragment * 1024

        // GELU of fragment jf of the chunk in accumulator set PG (features 32 (jf >> 1) + 16 (jf & 1) .. + 16 = registers
        // 8 s .. 8 s + 7 of block jf >> 1), as ONE batch of four element pairs: a pair is a chain of ten dependent instructions
        // (two quarter-rate transcendentals twice), and an in-order wave that runs the pairs one behind the other, each
        // between two of its MFMAs, pays the whole chain every time (250-350 cycles per pair, measured with stamps);
        // four chains side by side hide each other's latency.  The result is the down-projection's B fragment: published at
        // once (D reads it behind the barrier that closes the chunk's last GELU interval); the accumulators restart from the
        // bias of the chunk two ahead (b0: elements 0..3 of the fragment, b1: 4..7).
        auto gelu_pair = [&](auto pg_tag, auto jf_tag, auto p_tag, const f32x4 &b0, const f32x4 &b1, f16x8 &o) __attribute__((always_inline)) {
            constexpr int PG = decltype(pg_tag)::value, jf = decltype(jf_tag)::value, fb = jf >> 1, s8 = 8 * (jf & 1), p = decltype(p_tag)::value;
            const float x0 = accU[PG][fb][s8 + 2 * p], x1 = accU[PG][fb][s8 + 2 * p + 1];
            if constexpr (LT_ABLATE & 2) {
                o[2 * p] = (_Float16)x0; o[2 * p + 1] = (_Float16)x1;
            } else {
                const f16x2_t gv = gelu_pk16(x0, x1);
                o[2 * p] = gv[0]; o[2 * p + 1] = gv[1];
            }
            accU[PG][fb][s8 + 2 * p] = p < 2 ? b0[2 * p] : b1[2 * p - 4];
            accU[PG][fb][s8 + 2 * p + 1] = p < 2 ? b0[2 * p + 1] : b1[2 * p - 3];
            if constexpr (p == 3 && !(LT_ABLATE & 16)) *(f16x8 *)(Gw + PG * 4096 + jf * 1024) = o;
        };
        // fragments of GELU interval jg: fstart(jg) .. fstart(jg + 1) - 1 of 4
        auto fstart = [](int jg) constexpr { return (4 * jg + NG - 1) / NG; };
        constexpr int MAXF = (4 + NG - 1) / NG;

        // One interval of U.  C_PAR: parity of chunk c (accumulator set of UP(c)); MMA: UP tile (c, j) exists; GELU: chunk c-1
        // exists and fragments fstart(j) .. of it are GELU'ed between this interval's MFMAs; CARRY: the last group of the
        // previous tile is owed.
        auto u_interval = [&](auto par_tag, auto j_tag, auto mma_tag, auto gelu_tag, auto carry_tag, int c) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_tag)::value, j = decltype(j_tag)::value;
            constexpr bool MMA = decltype(mma_tag)::value, GELU = decltype(gelu_tag)::value, CARRY = decltype(carry_tag)::value;
            constexpr int f0 = GELU ? fstart(j) : 0, nf = GELU ? fstart(j + 1) - fstart(j) : 0;
            constexpr int PG = PAR ^ 1;                                           // accumulator set of chunk c - 1
            f16x8 Fa[4], Fb[4];
            [[maybe_unused]] const bool fine = tlU && tl >= 40 && tl < 46;                 // (timeline builds: stamps inside six intervals)
            [[maybe_unused]] const int fb0 = 402 + (tl - 40) * 10;
            LT_STAMP_FINE(fine, fb0 + 0);
            unsigned bU = aUp + (unsigned)(slot * LT_TILE);
            asm volatile("" : "+v"(bU));                                          // (made here, not kept per slot)
            // group g4 = k-steps 2 g4, 2 g4 + 1 x both row blocks
            auto rd = [&](int g4, f16x8 (&F)[4]) __attribute__((always_inline)) {
                unsigned a0 = bU ^ (unsigned)((2 * g4) << 5), a1 = bU ^ (unsigned)((2 * g4 + 1) << 5);
                asm volatile("" : "+v"(a0), "+v"(a1));
                F[0] = frag_read<0>(a0); F[1] = frag_read<8192>(a0);
                F[2] = frag_read<0>(a1); F[3] = frag_read<8192>(a1);
            };
            // biases of this interval's fragments (of the chunk two ahead of the one being GELU'ed = c + 1), by hand, first in the
            // queue: the wait for the first fragment group retires them too
            f32x4 Bv[MAXF][2];
            if constexpr (GELU) {
                const unsigned aBc = aB + (unsigned)(c + 1) * 256u;
                static_for<nf>([&](auto k_tag) __attribute__((always_inline)) {
                    constexpr int k = decltype(k_tag)::value, jf = f0 + k, off = 4 * (32 * (jf >> 1) + 16 * (jf & 1));
                    Bv[k][0] = lds_read_f32x4_h<off>(aBc);
                    Bv[k][1] = lds_read_f32x4_h<off + 32>(aBc);
                });
            }
            if constexpr (MMA) { rd(0, Fa); rd(1, Fb); }
            LT_STAMP_FINE(fine, fb0 + 1);
            // pair q (0 .. 4 nf - 1; fragment q / 4, pair q % 4) rides behind this wave's MFMA slot gslot(q) of the interval (4 carried,
            // then 12 own).  LT_GELU_BATCH: the four pairs of a fragment together behind the last MFMA of a group (their ten-deep
            // dependency chains side by side), else spread evenly over the own MFMAs
            f16x8 og[MAXF];
            auto gslot = [](int q) constexpr { return LT_GELU_BATCH ? 7 + 4 * (q / 4) : 4 + (q * 12) / (4 * (nf > 0 ? nf : 1)); };
            auto fill = [&](auto slot_tag) __attribute__((always_inline)) {
                constexpr int sl = decltype(slot_tag)::value;
                static_for<4 * nf>([&](auto q_tag) __attribute__((always_inline)) {
                    constexpr int q = decltype(q_tag)::value, k = q / 4;
                    if constexpr (gslot(q) == sl)
                        gelu_pair(std::integral_constant<int, PG>{}, std::integral_constant<int, f0 + k>{}, std::integral_constant<int, q % 4>{}, Bv[k][0], Bv[k][1], og[k]);
                });
            };
            // the carried group: k-steps 6, 7 of the previous tile (jp) into the accumulator set of ITS chunk
            if constexpr (CARRY) {
                constexpr int jp = j == 0 ? NT - 1 : j - 1, PC = j == 0 ? PAR ^ 1 : PAR;
                accU[PC][0] = mfma16(Fc[0], Y[8 * jp + 6], accU[PC][0]);
                accU[PC][1] = mfma16(Fc[1], Y[8 * jp + 6], accU[PC][1]);
                accU[PC][0] = mfma16(Fc[2], Y[8 * jp + 7], accU[PC][0]);
                accU[PC][1] = mfma16(Fc[3], Y[8 * jp + 7], accU[PC][1]);
                __builtin_amdgcn_sched_barrier(0);
                LT_STAMP_FINE(fine, fb0 + 2);
            }
            auto mm = [&](f16x8 (&F)[4], auto g4_tag) __attribute__((always_inline)) {
                constexpr int g4 = decltype(g4_tag)::value, sb = 4 + 4 * g4;
                accU[PAR][0] = mfma16(F[0], Y[8 * j + 2 * g4], accU[PAR][0]);
                fill(std::integral_constant<int, sb + 0>{});
                accU[PAR][1] = mfma16(F[1], Y[8 * j + 2 * g4], accU[PAR][1]);
                fill(std::integral_constant<int, sb + 1>{});
                accU[PAR][0] = mfma16(F[2], Y[8 * j + 2 * g4 + 1], accU[PAR][0]);
                fill(std::integral_constant<int, sb + 2>{});
                accU[PAR][1] = mfma16(F[3], Y[8 * j + 2 * g4 + 1], accU[PAR][1]);
                fill(std::integral_constant<int, sb + 3>{});
                __builtin_amdgcn_sched_barrier(0);
            };
            auto bias_landed = [&]() __attribute__((always_inline)) {
                static_for<nf>([&](auto k_tag) __attribute__((always_inline)) { landed(Bv[decltype(k_tag)::value][0]); landed(Bv[decltype(k_tag)::value][1]); });
            };
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
            if constexpr (MMA) {
                // (the first own group's wait retires the bias reads too; by then the carried MFMAs, which in a chunk's first
                // interval still write the accumulator set being read, are through)
                wait_frags<4>(Fa);
                bias_landed();
                LT_STAMP_FINE(fine, fb0 + 3);
                mm(Fa, I0{});
                LT_STAMP_FINE(fine, fb0 + 4);
                rd(2, Fa);
                wait_frags<4>(Fb);
                LT_STAMP_FINE(fine, fb0 + 5);
                mm(Fb, I1{});
                LT_STAMP_FINE(fine, fb0 + 6);
                rd(3, Fc);
                wait_frags<4>(Fa);
                LT_STAMP_FINE(fine, fb0 + 7);
                mm(Fa, I2{});
                LT_STAMP_FINE(fine, fb0 + 8);
            } else {
                wait_lgkm<0>();
                bias_landed();
                static_for<12>([&](auto s_tag) __attribute__((always_inline)) { fill(std::integral_constant<int, 4 + decltype(s_tag)::value>{}); });
            }
            static_assert(MAXF <= 2, "two GELU batches per interval at most");
            pre_close();
            if constexpr (MMA) close_interval<VMQ>(Fc); else close_interval<VMQ>();
            next_slot();
        };
        using TT = std::true_type; using FF = std::false_type;
        LT_STAMP(tlU, 393);
        __builtin_amdgcn_s_setprio(LT_UPRIO);                 // U's stream is the longer one: it wins the arbitration for the SIMD
        // chunk c: UP(c) with gelu(c - 1) as filler
        auto u_chunk = [&](auto par_tag, auto first_tag, int c) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;            // c == 0
            static_for<NT>([&](auto j_tag) __attribute__((always_inline)) {
                constexpr int j = decltype(j_tag)::value;
                constexpr bool carry = !(FIRST && j == 0), gelu = !FIRST && j < NG;
                u_interval(par_tag, j_tag, TT{}, std::integral_constant<bool, gelu>{}, std::integral_constant<bool, carry>{}, c);
            });
        };
        u_chunk(std::integral_constant<int, 0>{}, TT{}, 0);
        for (int c = 1; c + 1 < NC; c += 2) {
            u_chunk(std::integral_constant<int, 1>{}, FF{}, c);
            u_chunk(std::integral_constant<int, 0>{}, FF{}, c + 1);
        }
        u_chunk(std::integral_constant<int, 1>{}, FF{}, NC - 1);
        // tail: the carried group of the last tile, gelu(NC - 1) (parity of "chunk NC" = 0), then idle barriers
        static_for<LAGT>([&](auto k_tag) __attribute__((always_inline)) {
            constexpr int k = decltype(k_tag)::value;
            if constexpr (k < NG)
                u_interval(std::integral_constant<int, 0>{}, std::integral_constant<int, k>{}, FF{}, TT{}, std::integral_constant<bool, k == 0>{}, NC);
            else {
                // first idle interval: this wave's half of y (the residual of its features) goes to ringU (idle since its last
                // tile: pair t owns NYH KiB like above); D adds it in front of LayerNorm 2
                if constexpr (k == NG) {
                    char *const xu = ringU + t * (NYH * 1024) + lane * 16;
#pragma unroll
                    for (int m = 0; m < NYH; ++m) *(f16x8 *)(xu + m * 1024) = Y[8 * (m >> 2) + (m & 3)];
                }
                pre_close(); close_interval<VMQ>(); next_slot();
            }
        });
        static_assert(LAGT > NG, "U needs an idle interval to hand its half of y over");
        // LayerNorm 2 is D's; U stores half of the rows below
        asm volatile("s_barrier" ::: "memory");                                   // [E0] every D wave has taken U's half of y in
        asm volatile("s_barrier" ::: "memory");                                   // [E1] D's rows are staged
        store_rows();
        return;
    } else {
        // =====================================================================================================
        // D: all output accumulators, down-projection
        // =====================================================================================================
        // accumulators start from b2 (+ y, the residual, for D's own features: blocks ob = 2, 3 of every n3; U's half of y is
        // added at the end): block n = 4 n3 + ob holds features 32 n + 8 (r >> 2) + 4 hi + (r & 3)
        f16x8 yh[NYH];
        static_for<NBH>([&](auto b_tag) __attribute__((always_inline)) {
            constexpr int b = decltype(b_tag)::value;
            normalise_block(b_tag, yh[2 * b], yh[2 * b + 1]);
            *(f16x8 *)(xch + (2 * b) * 1024) = yh[2 * b];
            *(f16x8 *)(xch + (2 * b + 1) * 1024) = yh[2 * b + 1];
        });
        LT_STAMP(tlD, 400);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // [B2] D's half is in LDS
        // (in this order: the 96 out-projection accumulators are dead before the 192 of the feed-forward come to life)
        f32x16 acc2[NB];
        static_for<NBH>([&](auto b_tag) __attribute__((always_inline)) {
            constexpr int b = decltype(b_tag)::value, n3 = b >> 1, obp = b & 1;
            f32x4 bu[4], bd[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                bu[gq] = *(const f32x4 *)(cb2 + n3 * 128 + obp * 32 + 8 * gq + 4 * hi);
                bd[gq] = *(const f32x4 *)(cb2 + n3 * 128 + 64 + obp * 32 + 8 * gq + 4 * hi);
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc2[4 * n3 + obp][4 * gq + e] = bu[gq][e];
                    acc2[4 * n3 + 2 + obp][4 * gq + e] = (float)yh[2 * b + (gq >> 1)][4 * (gq & 1) + e] + bd[gq][e];
                }
            asm volatile("" ::: "memory");
        });
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0), visible to the compiler
        // fragment groups: one k-step = the four row blocks of the weight tile + the matching B fragment of the GELU'ed chunk
        // (re-read per tile: holding the chunk's four fragments would cost 12 registers more than the 256 there are)
        const unsigned aD = aR + 3 * LT_TILE;                 // ringD
        const unsigned offI = rows_offset(I);
        auto dma_down = [&](int c, int n3, unsigned tile) __attribute__((always_inline)) {
            if constexpr (Q4) q4_request_rows(a.w2_qs, a.w2_sc, n3 * (I / 64) + c);
            else dma128(w2p + (size_t)n3 * 128 * I + c * 64, offI, I * 2, tile);
        };
        const unsigned aG = lds_addr(G) + (unsigned)(t * 8192 + lane * 16);
        // D requests EVERY feed-forward tile (U's stream is the longer one: GELU, bias reads): in feed-forward interval i the
        // up-projection tile i + 2 (into ringU) and the down-projection tile i - LAGT + 2 (into ringD), four pieces each
        const unsigned offU = up_offset();
        const int NUP = NC * NT;                              // up-projection tiles (= down-projection tiles)
        auto request = [&](auto up_tag, auto down_tag, int iu, int id) __attribute__((always_inline)) {
            const int s2 = slot_plus2();
            if constexpr (decltype(up_tag)::value) dma_up(iu / NT, iu % NT, offU, ldsU + s2 * LT_TILE);
            if constexpr (decltype(down_tag)::value) dma_down(id / NT, id % NT, ldsD + s2 * LT_TILE);
        };
        using TT = std::true_type; using FF = std::false_type;
        if (LT_DPRIO) __builtin_amdgcn_s_setprio(LT_DPRIO);
        // head: LAGT intervals without a tile of its own
        static_for<LAGT>([&](auto k_tag) __attribute__((always_inline)) {
            constexpr int k = decltype(k_tag)::value;
            constexpr bool dn = k >= LAGT - 2;
            if constexpr (k >= 1) q4_expand_up(ringU + slot_plus1() * LT_TILE);
            if constexpr (k == LAGT - 1) q4_expand_rows(PERMUTED{}, ringD + slot_plus1() * LT_TILE);
            request(TT{}, std::integral_constant<bool, dn>{}, k + 2, k - (LAGT - 2));
            // at the close the pieces of the interval before have landed: all but the newest 4 (+ 4)
            pre_close();
            close_interval<Q4 ? 63 : (dn ? 8 : 4)>();
            next_slot();
        });
        // DOWN(c), tile d: acc2[4 d + ob] += W2(row block ob, k-step kk) x g(c)[kk].  UPQ: how many of the chunk's intervals
        // still have an up-projection tile to request (NT, or fewer near the end); LAST: c == NC - 1
        auto d_chunk = [&](auto upq_tag, auto last_tag, auto prev_up_tag, int c) __attribute__((always_inline)) {
            constexpr int UPQ = decltype(upq_tag)::value;
            constexpr bool LAST = decltype(last_tag)::value;
            constexpr bool PREV_UP = decltype(prev_up_tag)::value;    // the chunk before requested an up-projection tile in its last interval
            static_for<NT>([&](auto d_tag) __attribute__((always_inline)) {
                constexpr int d = decltype(d_tag)::value;
                constexpr bool up = d < UPQ, dn = !(LAST && d + 2 >= NT);
                f16x8 Fa[4], Fb[4], ga, gb;
                unsigned bD = aD + (unsigned)(slot * LT_TILE), bG = aG + (unsigned)((c & 1) * 4096);
                asm volatile("" : "+v"(bD), "+v"(bG));         // (made here, not kept per slot)
                auto rd = [&](auto kk_tag, f16x8 (&F)[4], f16x8 &gv) __attribute__((always_inline)) {
                    constexpr int kk = decltype(kk_tag)::value;
                    unsigned a0 = bD ^ (unsigned)(kk << 5);
                    asm volatile("" : "+v"(a0));
                    F[0] = frag_read<0>(a0); F[1] = frag_read<4096>(a0); F[2] = frag_read<8192>(a0); F[3] = frag_read<12288>(a0);
                    gv = lds_read_b128_u<kk * 1024>(bG);
                };
                auto wait_group = [&](auto n_tag, f16x8 (&F)[4], f16x8 &gv) __attribute__((always_inline)) {
                    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(gv) : "n"(decltype(n_tag)::value) : "memory");
                };
                auto mm = [&](f16x8 (&F)[4], const f16x8 &gv) __attribute__((always_inline)) {
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob) acc2[4 * d + ob] = mfma16(F[ob], gv, acc2[4 * d + ob]);
                    __builtin_amdgcn_sched_barrier(0);
                };
                using N5 = std::integral_constant<int, 5>; using N0 = std::integral_constant<int, 0>;
                // q4: the blocks requested one interval ago are expanded (into the slots of the next interval) first thing, each
                // followed at once by its next request: a request has a whole interval to land, the expansion's VALU work runs
                // under U's MFMAs, and no fragment is live yet (the registers are all taken further down)
                constexpr bool pend_up = Q4 && (d >= 1 ? d - 1 < UPQ : PREV_UP), pend_dn = Q4 && (d >= 1 ? !(LAST && d + 1 >= NT) : true);
                if constexpr (Q4) {
                    if constexpr (pend_up) q4_expand_up(ringU + slot_plus1() * LT_TILE);
                    request(std::integral_constant<bool, up>{}, FF{}, c * NT + d + LAGT + 2, 0);
                    if constexpr (pend_dn) q4_expand_rows(PERMUTED{}, ringD + slot_plus1() * LT_TILE);
                    request(FF{}, std::integral_constant<bool, dn>{}, 0, c * NT + d + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                rd(std::integral_constant<int, 0>{}, Fa, ga);
                rd(std::integral_constant<int, 1>{}, Fb, gb);
                wait_group(N5{}, Fa, ga);
                mm(Fa, ga);
                rd(std::integral_constant<int, 2>{}, Fa, ga);
                if constexpr (!Q4) {
                    if constexpr (LT_DMA_SPLIT) request(std::integral_constant<bool, up>{}, FF{}, c * NT + d + LAGT + 2, 0);
                    else request(std::integral_constant<bool, up>{}, std::integral_constant<bool, dn>{}, c * NT + d + LAGT + 2, c * NT + d + 2);
                }
                wait_group(N5{}, Fb, gb);
                mm(Fb, gb);
                rd(std::integral_constant<int, 3>{}, Fb, gb);
                if constexpr (!Q4 && LT_DMA_SPLIT) request(FF{}, std::integral_constant<bool, dn>{}, 0, c * NT + d + 2);
                wait_group(N5{}, Fa, ga);
                mm(Fa, ga);
                wait_group(N0{}, Fb, gb);
                mm(Fb, gb);
                pre_close();
                close_interval<Q4 ? 63 : (up ? 4 : 0) + (dn ? 4 : 0)>();
                next_slot();
            });
        };
        // up-projection tile of DOWN tile id: id + LAGT + 2 < NUP, i.e. every tile of the chunks c <= NC - 4, the first NT - 2 of
        // chunk NC - 3 (LAGT = 2 NT), none later
        static_assert(LAGT == 2 * NT, "the request schedule below assumes D runs two chunks behind");
        (void)NUP;
        for (int c = 0; c + 3 < NC; ++c) d_chunk(std::integral_constant<int, NT>{}, FF{}, TT{}, c);
        d_chunk(std::integral_constant<int, NT - 2>{}, FF{}, TT{}, NC - 3);
        d_chunk(std::integral_constant<int, 0>{}, FF{}, FF{}, NC - 2);
        d_chunk(std::integral_constant<int, 0>{}, TT{}, FF{}, NC - 1);

        // ================================ LayerNorm 2 (wave-local) -> rows staged in LDS ================================
        {
            char *S = smem + t * (64 * H);                    // 32 tokens x H halfs (the rings are idle: every wave is past its last read)
            int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // (not kept through the loop)
            asm volatile("" : "+v"(lane));
            const int hi = lane >> 5;
            // U's half of the residual: fragment m = 4 n3 + 2 obp + s of pair t = registers 8 s .. of block 4 n3 + obp
            {
                const char *const xu = ringU + t * (NYH * 1024) + lane * 16;
#pragma unroll
                for (int n3 = 0; n3 < NT; ++n3) {
                    f16x8 yu[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) yu[m] = *(const f16x8 *)(xu + (4 * n3 + m) * 1024);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc2[4 * n3 + (m >> 1)][8 * (m & 1) + e] += (float)yu[m][e];
                    asm volatile("" ::: "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // [E0] the staging below overwrites other pairs' areas
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) { s1 += acc2[n][r]; s2 = __builtin_fmaf(acc2[n][r], acc2[n][r], s2); }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            float rstd, nmr;
            layernorm_scale(s1, s2, 1.0f / H, rstd, nmr);
            // fragment (q, lane) -> position (lane + 2q) & 63 of row q: the row-major read of the store loop is conflict-free.
            // The eight parameter reads of a block are in flight together.
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                f32x4 gv[4], bv[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) { gv[g4] = *(const f32x4 *)(cg2 + 32 * n + 8 * g4 + 4 * hi); bv[g4] = *(const f32x4 *)(cbe2 + 32 * n + 8 * g4 + 4 * hi); }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int q = 2 * n + s;
                    f16x8 o;
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            o[4 * hq + e] = (_Float16)rounded_f32(__builtin_fmaf(acc2[n][8 * s + 4 * hq + e], gv[2 * s + hq][e] * rstd, __builtin_fmaf(gv[2 * s + hq][e], nmr, bv[2 * s + hq][e])));
                    *(f16x8 *)(S + q * 1024 + ((lane + 2 * q) & 63) * 16) = o;
                }
                asm volatile("" ::: "memory");
            }
            LT_STAMP(tlD, 401);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // [E1]
        }
        store_rows();
    }
}

template <int NT, int WT>
__global__ __launch_bounds__(512) void layer_tail_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    layer_tail_body<NT, WT>(a, smem, (int)blockIdx.x * 128, 128, (int)threadIdx.x);
}

static size_t layer_tail_lds(int H, int I) {
    return (size_t)6 * LT_TILE + 32768 + 8 * 64 * 4 + (size_t)(6 * H + I + 128) * sizeof(float);
}

#ifndef BERT_HIP_PHASES_ONLY

bool layer_tail_supported(const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2) {
    const int H = W1.K, I = W1.N;
    if (Wo.type != W1.type || W1.type != W2.type) return false;
    if (W1.type == GW_F16 ? (!Wo.w16 || !W1.w16p || !W2.w16p) : (!Wo.qs || !W1.qs || !W2.qs)) return false;
    if (Wo.N != H || Wo.K != H || W2.N != H || W2.K != I) return false;
    if (H % 128 != 0 || H < 256 || H > 384 || I % 128 != 0 || I < 256) return false;     // an even number >= 4 of 64-feature chunks
    return layer_tail_lds(H, I) <= 160 * 1024;
}

void launch_layer_tail(const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2, const half_t *ctx, const half_t *x,
                        const float *bo, const float *g1, const float *be1, const float *b1, const float *b2,
                        const float *g2, const float *be2, half_t *out, int M_pad, hipStream_t stream) {
    TailArgs a;
    a.ctx = ctx; a.x = x; a.wo = Wo.w16; a.w1p = W1.w16p; a.w2p = W2.w16p;
    a.wo_qs = Wo.qs; a.w1_qs = W1.qs; a.w2_qs = W2.qs; a.wo_sc = Wo.sc; a.w1_sc = W1.sc; a.w2_sc = W2.sc;
    a.bo = bo; a.g1 = g1; a.be1 = be1; a.b1 = b1; a.b2 = b2; a.g2 = g2; a.be2 = be2; a.out = out;
    a.I = W1.N;
    const int H = W1.K;
    const size_t lds = layer_tail_lds(H, a.I);
    static DeviceFlags configured[6];
    auto go = [&](auto kernel, int which) {
        configure_once(configured[which], [&] { (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        BERT_LAUNCH(kernel, dim3(M_pad / 128), dim3(512), lds, stream, a);
#ifdef BERT_HIP_TIMELINE
        static int shots = 0;
        if (M_pad >= 128 * 256 && shots++ == 20) {
            (void)hipDeviceSynchronize();
            static unsigned long long h[256 * 512];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tl_tail), sizeof(h));
            for (int b : {0, 100}) {
                const unsigned long long *r = h + b * 512, t0 = r[0];
                auto rel = [&](int i) { return r[i] ? (long long)(r[i] - t0) : -1LL; };
                fprintf(stderr, "tail wg %d phases:", b);
                for (int i = 384; i < 402; ++i) fprintf(stderr, " %d:%lld", i, rel(i));
                fprintf(stderr, "\nt3 wg %d inside U's intervals 40..45 (start, reads issued, carried done, g0 landed, g0 done, g1 landed, g1 done, g2 landed, g2 done):", b);
                for (int k = 0; k < 6; ++k) {
                    fprintf(stderr, " [");
                    for (int i = 1; i < 9; ++i) fprintf(stderr, "%lld ", r[402 + k * 10 + i] ? (long long)(r[402 + k * 10 + i] - r[402 + k * 10]) : -1LL);
                    fprintf(stderr, "]");
                }
                fprintf(stderr, "\nt3 wg %d intervals (start | U arrives +, D arrives +, length):", b);
                for (int i = 1; i < 128 && r[i]; ++i)
                    fprintf(stderr, " [%d %lld | %lld %lld %lld]", i, rel(i - 1), (long long)(r[128 + i] - r[i - 1]), r[256 + i] ? (long long)(r[256 + i] - r[i - 1]) : -1LL, (long long)(r[i] - r[i - 1]));
                fprintf(stderr, "\n");
            }
        }
#endif
    };
    const int nt3 = H == 384;
    switch (W1.type) {
    case GW_F16:  if (nt3) go(layer_tail_kernel<3, GW_F16>, 1); else go(layer_tail_kernel<2, GW_F16>, 0); break;
    case GW_Q4_0: if (nt3) go(layer_tail_kernel<3, GW_Q4_0>, 3); else go(layer_tail_kernel<2, GW_Q4_0>, 2); break;
    default:      if (nt3) go(layer_tail_kernel<3, GW_Q4_1>, 5); else go(layer_tail_kernel<2, GW_Q4_1>, 4); break;
    }
}

#endif  // BERT_HIP_PHASES_ONLY

}  // namespace bert_hip
