// qkv_attention2.hip — Q|K|V projection and self-attention of a 128-slot token WINDOW in one kernel (gfx950, d_head = 32,
// f16 or q4 weights, H = 128 / 256 / 384; reference bert.cpp:822-856).  (Its predecessor gave a workgroup ONE sentence and
// paid for 128 tokens whatever the length; it is gone.)
//
// A workgroup owns a window of 128 token slots that holds one or SEVERAL whole sentences of the packed batch (each
// starts at a multiple of 16 slots, see below), so batches of short sentences no longer fall back to the path that
// moves Q|K|V through HBM.  Eight waves, two per SIMD:
//   waves 4..7  "projection" waves: wave t owns token block t (32 slots).  Its rows of the hidden state live in
//               REGISTERS for the whole kernel (H/16 MFMA fragments, loaded once straight from HBM), so the hidden
//               state never goes through LDS and a k-step costs 3 fragment reads (the Q, K and V weight rows of the
//               head) for 3 MFMAs.  The weight tiles ([96 rows x 64 k] = the head's Q, K, V rows) are shared by the
//               four waves: they stream through a ring of three SLABS (half a head each) by LDS-DMA, two slabs
//               ahead, every wave requesting a quarter of the pieces; one barrier per slab.
//   waves 0..3  "attention" waves: wave a owns query block a of the head projected one step earlier: S^T = K Q^T,
//               softmax over the keys of the query's own sentence, O^T = V^T P^T, normalise, store (as attention.hip).
// Wave a and wave a+4 share a SIMD: the projection MFMAs run under the softmax VALU work, and every SIMD carries the
// same number of MFMAs (72 + 16 per head).  Q_h / K_h / V_h^T are double-buffered in LDS (head h in buffer h & 1), so
// the projection waves never wait for the attention of the previous head; per head the two groups meet at two barriers:
// the slab barrier in the middle of the head and, at its end, the barrier that both opens the next head's first slab and
// publishes Q/K/V^T.  The attention waves execute the first one in the middle of their own work (Q2_PM).  In the head's
// last k-tile the MFMAs run matrix by matrix (Q, then K, then V), so that the conversion of Q and K to f16 overlaps
// with the remaining MFMAs.
//
// Windows and bit-exactness: sentence j of the window starts at slot off_j with off_0 = 0, off_{j+1} = off_j + n_j
// rounded up to 16.  A softmax row only ever sees the keys of its own sentence (the others are masked with -inf before
// the running maximum, their probabilities are exact zeros), and because 16 slots are one k-step of the P V MFMAs a
// sentence's keys occupy the same k positions inside every instruction wherever the sentence sits: the zeros in front
// of them add exactly nothing, so a sentence gives the same bits in any window, alone or not (tested).  Key tiles
// that no query of a block needs are skipped, tiles that lie inside the sentence of every query of the block are not
// masked at all.
#include "tile_stream.h"

// tuning knobs (A/B builds): the attention waves execute the head's first barrier after step Q2_PM of their 8 steps
// (0..3 = S^T key tiles, 4..7 = P V key tiles); Q2_PRIO = s_setprio level of the attention waves
#ifndef Q2_PM
#define Q2_PM 3
#endif
#ifndef Q2_PRIO
#define Q2_PRIO 0
#endif

namespace bert_hip {

namespace {

constexpr int Q2_WIN = 128;                      // token slots per workgroup
constexpr int Q2_TILE = 12288;                   // [96 rows x 64 halfs]: Q_h, K_h, V_h rows of one k-tile
constexpr int Q2_VT_LD = Q2_WIN + 4;             // halfs per V^T row (8-byte skew: conflict-free ds_read_b64)

struct Qkv2Args {
    const half_t *x;         // [T_pad][H] hidden state, packed sentences
    const half_t *w;         // [3H (padded)][H] f16: Q rows, K rows, V rows
    const uint4 *qs;         // q4_0 / q4_1 weights instead (kernels.h GemmWeight: nibble plane + scale plane, tile-contiguous)
    const void *sc;
    const float *bias;       // [3H]
    const int32_t *cu;       // [n_sent + 1]
    const int2 *groups;      // per workgroup {first sentence, count}; nullptr: `spw` sentences per workgroup
    const int *n_groups;     // device word holding the number of windows (device-built windows: the grid is an upper bound), or nullptr
    half_t *out;             // [T_pad][H] attention context
    int n_head, n_sent, spw;
    int slot_mask;           // sentence places in a window start at multiples of slot_mask + 1 slots (15; 7 with BERT_HIP_WINDOW_SLOTS=8)
};

__device__ __forceinline__ int q2_off32(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int OFF>
__device__ __forceinline__ f16x8 q2_read_b128(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// everything but the newest six hand-issued reads has landed: hands one half's fragments to its MFMAs
__device__ __forceinline__ void q2_wait6(f16x8 (&f)[2][3]) {
    asm volatile("s_waitcnt lgkmcnt(6)"
                 : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[0][2]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[1][2]) : : "memory");
}
// slab barrier: this wave's pieces of the next slab have landed (all but the newest VM pieces), every read of the
// slab that is about to be overwritten has returned
template <int VM>
__device__ __forceinline__ void q2_slab_barrier(f16x8 (&f)[2][3]) {
    asm volatile("s_waitcnt vmcnt(%6) lgkmcnt(0)\n\ts_barrier"
                 : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[0][2]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[1][2]) : "n"(VM) : "memory");
}
// all but the newest fourteen reads have landed (the last half's six fragment reads and the eight bias vectors behind them)
__device__ __forceinline__ void q2_wait14(f16x8 (&f)[2][3]) {
    asm volatile("s_waitcnt lgkmcnt(14)"
                 : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[0][2]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[1][2]) : : "memory");
}
// every read has landed: the last half's fragments and the bias vectors
__device__ __forceinline__ void q2_wait0_bias(f16x8 (&f)[2][3], f32x4 (&bq)[4], f32x4 (&bk)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[0][2]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[1][2]),
                   "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(bk[0]), "+v"(bk[1]), "+v"(bk[2]), "+v"(bk[3]) : : "memory");
}
template <int OFF>
__device__ __forceinline__ f32x4 q2_read_f32x4(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// hands over the V bias (an untracked global load older than every DMA piece that is still in flight at the head's first barrier)
__device__ __forceinline__ void q2_take(float &bv) { asm volatile("" : "+v"(bv)); }
// end of a head: Q/K/V^T are written, this wave's pieces of the next head's first slab have landed (all but the newest VM)
template <int VM>
__device__ __forceinline__ void q2_head_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" : : "n"(VM) : "memory");
}

}  // namespace

// H = 64 * KT = 32 * n_head; a slab = GB k-tiles, NBAR = KT / GB slabs per head.  WT: GW_F16, or GW_Q4_0 / GW_Q4_1 — the
// weights stay 4-bit in HBM and L2; the projection waves fetch the raw blocks of a slab into registers one slab period ahead
// of expanding them into the ring slot the f16 form fills by LDS-DMA (same tile image, same MFMA sequence, same bits).
// (the kernel's body as a device function of (arguments, the workgroup's LDS, window index): model_kernel.hip runs it as one
// phase of a launch that carries a window through all layers)
template <int KT, int GB, int WT>
__device__ __forceinline__ void qkv_attention2_body(const Qkv2Args &a, char *smem, const int window, const int tid) {
    constexpr bool Q4 = WT != GW_F16;
    constexpr int H = 64 * KT, NBAR = KT / GB, SLAB = GB * Q2_TILE, PPS = 3 * GB;   // PPS = DMA pieces per slab and wave
    static_assert(KT == 2 * GB && NBAR == 2, "");
    constexpr int QKV_BYTES = 2 * Q2_WIN * 64 + 32 * Q2_VT_LD * 2;   // Q [128][32] + K [128][32] (q2_off32 swizzle) + V^T [32][Q2_VT_LD]
    char *RING = smem;                                        // 3 slabs
    char *QKV = smem + 3 * SLAB;                              // two copies: head h in copy h & 1
    float *BS = (float *)(QKV + 2 * QKV_BYTES);               // [2H] Q and K bias (the V bias comes from global memory)

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int blk = wave & 3;                                 // token block (projection) / query block (attention)
    const int n_head = a.n_head;
    TL_STAMP_AT(tid == 256, 120);
    TL_REALTIME_AT(tid == 256, 121);
    TL_STAMP_AT(tid == 0, 250);
    TL_REALTIME_AT(tid == 0, 251);

    // ---- weight DMA of the projection waves: piece i of a slab for wave slot wp = tile i / 3, row block rb = i % 3 (Q, K, V
    // rows), rows wp*8 .. +8 of it.  Slab 0 is requested before anything else
    const int wp = blk;
    const unsigned loff = (unsigned)(((wp * 8 + (lane >> 3)) * H + (((lane & 7) ^ (((wp & 1) << 2) | ((lane >> 4) & 3))) * 8)) * 2);
    // piece i of slab j (compile time) of the head whose Q rows start at `hbase`
    auto dma_piece = [&](const char *hbase, auto j_tag, int dslot, auto i_tag) __attribute__((always_inline)) {
        constexpr int i = decltype(i_tag)::value, j = decltype(j_tag)::value, t = i / 3, rb = i % 3;
        const char *src = hbase + ((size_t)rb * H * H + (size_t)(j * GB + t) * 64) * 2;
        __builtin_amdgcn_global_load_lds(AS_GLOBAL(src + loff), AS_LDS(RING + dslot * SLAB + t * Q2_TILE + rb * 4096 + wp * 1024), 16, 0, 0);
    };
    const char *const wbase = (const char *)a.w;
    if constexpr (!Q4) {
        if (wave >= 4) static_for<PPS>([&](auto i) __attribute__((always_inline)) { dma_piece(wbase, std::integral_constant<int, 0>{}, 0, i); });
    }

    // ---- the window: sentences first .. first+count-1, sentence j at slots [off_j, off_j + n_j)
    int first, count;
    if (a.groups) {
        if (a.n_groups && window >= *a.n_groups) {   // beyond the windows the device-side builder produced
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        const int2 g = a.groups[window];
        first = g.x; count = g.y;
    }
    else { first = window * a.spw; count = min(a.spw, a.n_sent - first); }
    if (count <= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (never taken by the launcher's grids) the DMA must not outlive the workgroup
        return;
    }
    const int slot = blk * 32 + l31;
    int gtok = -1, k0 = 0, k1 = 0;                            // this lane's slot: global token, key range of its sentence
    {
        // (uniform rule, no window list: spw = 128 / round16(max_len) sentences per window.  A sentence longer than the
        // promised max_len is cut to round16(max_len) slots here — it is the one that gets a NaN row from the length guard —
        // instead of pushing its window neighbours past slot 127, where their rows would never be written)
        const int place = a.groups ? 128 : 128 / a.spw;
        int off = 0;
        for (int j = 0; j < count; ++j) {
            const int t0 = a.cu[first + j], n = min(a.cu[first + j + 1] - t0, place);
            if (slot >= off && slot < off + n) { gtok = t0 + slot - off; k0 = off; k1 = off + n; }
            off = (off + n + a.slot_mask) & ~a.slot_mask;
        }
    }

    if (wave >= 4) {
        // =============================== projection wave: token block `blk` ===============================
        // rows of the hidden state as MFMA fragments (token = l31, k = 16 ks + 8 hi ..): B operand of the Q / K
        // projections, A operand of the V projection.  Empty slots read the window's first token (finite values).
        f16x8 bf[4 * KT];
        {
            const int gt = gtok >= 0 ? gtok : a.cu[first];
            const half_t *xr = a.x + (size_t)gt * H + 8 * hi;
            // (untracked loads: the first barrier waits for them and for slab 0 only, slab 1 stays in flight)
#pragma unroll
            for (int ks = 0; ks < 4 * KT; ++ks) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bf[ks]) : "v"(xr + 16 * ks));
        }
        // (slab 0 was requested at kernel entry) slab 1 behind the x rows: the first barrier leaves it in flight
        if constexpr (!Q4) static_for<PPS>([&](auto i) __attribute__((always_inline)) { dma_piece(wbase, std::integral_constant<int, NBAR - 1>{}, 1, i); });

        // ---- q4: a k-tile of a slab is [96 rows x 2 blocks of 32 weights] = three row blocks (the head's Q, K, V rows) of 64
        // blocks: one block per lane of a wave.  Three of the four projection waves take a row block of tile t each, rotated
        // with the tile and the slab so that the idle turn goes round: wave w takes row block rb = (w + t + slab) % 4, idle at 3.
        // Lane l = (row r = l >> 1, k half kb = l & 1):
        //   plane index = ((rb H/128 + h/4) H/64 + k-tile) 256 + (h % 4) 64 + l        (rows rb H + 32 h + r of Q|K|V: scalar + lane)
        //   LDS image   = tile t, row block rb, row r: chunk c of the row at c ^ ((r >> 1) & 7), like the DMA's source swizzle
        // The blocks of a slab are REQUESTED in the second half of the slab period before the one in whose first half they are
        // EXPANDED (into the slot the f16 form's DMA of that period targets: two slabs ahead of the one being multiplied).
        constexpr int NRAW = GB;
        [[maybe_unused]] RawBlock raw[NRAW];
        // (the lane's part of the image address is made where it is used: kept, it would be the 257th register)
        auto q4_image_of_lane = [&]() __attribute__((always_inline)) {
            int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(ln));
            return (ln >> 1) * 128 + (((4 * (ln & 1)) ^ ((ln >> 2) & 7)) << 4);
        };
        // tile i of slab j of head hq: this wave's row block (3: none)
        auto q4_rb = [&](int hq, int j, int i) __attribute__((always_inline)) { return (blk + i + 2 * hq + j) & 3; };
        auto q4_request = [&](int hq, int j, int i) __attribute__((always_inline)) {
            if constexpr (Q4) {
                const int rb = q4_rb(hq, j, i);
                if (rb == 3) return;
                const int sbase = ((rb * (H / 128) + (hq >> 2)) * (H / 64) + j * GB + i) * 256 + (hq & 3) * 64;
                raw[i] = q4_load_block<WT>(a.qs + sbase, WT == GW_Q4_0 ? (const void *)((const unsigned short *)a.sc + sbase) : (const void *)((const unsigned *)a.sc + sbase), (size_t)lane);
            }
        };
        auto q4_expand = [&](int hq, int j, int dslot, int i) __attribute__((always_inline)) {
            if constexpr (Q4) {
                const int rb = q4_rb(hq, j, i);
                if (rb == 3) return;
                char *const base = RING + dslot * SLAB + i * Q2_TILE + rb * 4096;
                const int q4_image = q4_image_of_lane();
                q4_expand_block<WT, false>(raw[i], [&](int k) __attribute__((always_inline)) { return base + (q4_image ^ (k << 4)); });
            }
        };
        if constexpr (Q4) {
            // slabs 0 and 1 of head 0: fetched together, expanded; then the first pending request (head 1's first slab)
            RawBlock first[2][NRAW];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < NRAW; ++i) {
                    q4_request(0, j, i);
                    first[j][i] = raw[i];
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < NRAW; ++i) {
                    raw[i] = first[j][i];
                    q4_expand(0, j, j, i);
                }
        }
        // per-lane LDS address of the weight fragment of k-step kk of a tile: the chunk swizzle is an XOR of 2*kk + hi
        const unsigned aX0 = lds_addr(RING) + off64(l31, hi);
        unsigned aS[4];
        int rslot = 0;                                        // ring slot of the slab being multiplied
        auto set_slot = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) aS[kk] = (aX0 ^ (unsigned)(kk << 5)) + (unsigned)rslot * SLAB;
        };
        set_slot();
        f16x8 F[2][2][3];                                     // [parity][k-step of the half][row block]
        f32x16 acc[3];
        auto read_half = [&](auto par_tag, auto tt_tag, auto half_tag) __attribute__((always_inline)) {
            constexpr int par = decltype(par_tag)::value, tt = decltype(tt_tag)::value, half = decltype(half_tag)::value;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                F[par][i][0] = q2_read_b128<tt * Q2_TILE>(aS[2 * half + i]);
                F[par][i][1] = q2_read_b128<tt * Q2_TILE + 4096>(aS[2 * half + i]);
                F[par][i][2] = q2_read_b128<tt * Q2_TILE + 8192>(aS[2 * half + i]);
            }
        };
        // x rows and slab 0 have landed (slab 1 may still be in flight)
        [[maybe_unused]] const bool tl_sel = tid == 256;
        TL_STAMP_AT(tl_sel, 0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" : : "n"(Q4 ? 0 : PPS) : "memory");
#pragma unroll
        for (int ks = 0; ks < 4 * KT; ++ks) asm volatile("" : "+v"(bf[ks]));
        if constexpr (Q4) {
#pragma unroll
            for (int i = 0; i < NRAW; ++i) q4_request(min(1, n_head - 1), 0, i);
        }
        TL_STAMP_AT(tl_sel, 1);
        read_half(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});

        const unsigned aBias = lds_addr(BS) + 16 * hi;          // Q bias of features 8 gg + 4 hi .. of head h at + (h * 32 + 8 gg) * 4
        for (int h = 0; h < n_head; ++h) {
            char *QS = QKV + (h & 1) * QKV_BYTES, *KS = QS + Q2_WIN * 64;
            half_t *VT = (half_t *)(KS + Q2_WIN * 64);
            // V bias of this lane's feature: an untracked global load, older than every DMA piece requested during the
            // head, so the counted waits of the slab barriers cover it
            float bv;
            if constexpr (Q4) bv = a.bias[2 * H + h * 32 + l31];     // (q4: no hand-counted DMA queue; the compiler's own wait)
            else asm volatile("global_load_dword %0, %1, off" : "=v"(bv) : "v"(a.bias + 2 * H + h * 32 + l31));
            f32x4 bq[4], bk[4];
            // slab j of this head requests slab j of the next head (two slabs ahead); past the end the last head's
            // slabs are requested again, into slots nobody reads any more
            const char *const hnext = wbase + (size_t)min(h + 1, n_head - 1) * 32 * H * 2;
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
            constexpr int LT0 = 2 * KT - 2, LT1 = 2 * KT - 1;      // the halves of the head's last k-tile
            auto pieces = [&](auto hh_tag, auto m_tag) __attribute__((always_inline)) {
                // the pieces of the slab after next go behind MFMAs 1 and 4 of a half (piece i in half (2 i) / 3 of the slab),
                // into the slot the previous slab was read from
                constexpr int hh = decltype(hh_tag)::value, m = decltype(m_tag)::value, hs = hh % (2 * GB);
                constexpr int i0 = (3 * hs + 1) / 2, i1 = (3 * (hs + 1) + 1) / 2;
                const int dslot = (hh == 2 * GB - 1) ? (rslot == 2 ? 0 : rslot + 1) : (rslot == 0 ? 2 : rslot - 1);
                if constexpr (Q4) {
                    // half i of the slab period (i < GB): tile i of slab j of the NEXT head goes into the ring; half GB + i: tile i of the
                    // slab after that one (slab 1 of the next head, slab 0 of the head after next) is requested.  (Nothing is expanded
                    // in the head's last two halves, where the bias vectors and the Q / K conversions need the registers.)
                    constexpr int j = (hh >> 1) / GB;
                    if constexpr (m == 1) {
                        if constexpr (hs < GB) { if (h + 1 < n_head) q4_expand(h + 1, j, dslot, hs); }
                        else q4_request(min(j == 0 ? h + 1 : h + 2, n_head - 1), j ^ 1, hs - GB);
                    }
                } else {
                if constexpr (m == 1) dma_piece(hnext, std::integral_constant<int, (hh >> 1) / GB>{}, dslot, std::integral_constant<int, i0>{});
                if constexpr (m == 4 && i1 - i0 == 2) dma_piece(hnext, std::integral_constant<int, (hh >> 1) / GB>{}, dslot, std::integral_constant<int, i0 + 1>{});
                }
            };
            auto mma = [&](auto par_tag, auto i_tag, auto rb_tag, auto ks_tag) __attribute__((always_inline)) {
                constexpr int par = decltype(par_tag)::value, i = decltype(i_tag)::value, rb = decltype(rb_tag)::value, ks = decltype(ks_tag)::value;
                if constexpr (rb < 2) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[par][i][rb], bf[ks], acc[rb], 0, 0, 0);
                else acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[ks], F[par][i][2], acc[2], 0, 0, 0);
            };
            static_for<2 * KT>([&](auto hh_tag) __attribute__((always_inline)) {
                constexpr int hh = decltype(hh_tag)::value, par = hh & 1;
                constexpr int nhh = hh + 1, ntt = (nhh >> 1) % GB, nhalf = nhh & 1;
                using PAR = std::integral_constant<int, par>;
                if constexpr (hh == 2 * GB - 1) {
                    // the slab barrier in the middle of the head sits in front of the first slab's last half.  Outstanding here,
                    // oldest first: the V bias, the second slab (complete), then the next head's first slab WITHOUT the one
                    // piece this half is about to request
                    if (h < 8) TL_STAMP_AT(tl_sel, 2 + 6 * h);
                    q2_slab_barrier<Q4 ? 63 : PPS - 1>(F[par]);
                    if (h < 8) TL_STAMP_AT(tl_sel, 2 + 6 * h + 1);
                    q2_take(bv);
                    rslot = rslot == 2 ? 0 : rslot + 1;
                    set_slot();
                }
                if constexpr (hh < LT1) read_half(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, ntt>{}, std::integral_constant<int, nhalf>{});
                if constexpr (hh < LT0) {
                    q2_wait6(F[par]);
                    static_for<6>([&](auto m_tag) __attribute__((always_inline)) {      // k-step by k-step: Q K V Q K V
                        constexpr int m = decltype(m_tag)::value;
                        mma(PAR{}, std::integral_constant<int, m / 3>{}, std::integral_constant<int, m % 3>{}, std::integral_constant<int, 2 * hh + m / 3>{});
                        pieces(hh_tag, m_tag);
                    });
                } else if constexpr (hh == LT0) {
                    // the head's Q / K biases, read by hand behind the last half's fragments: they land under this half's MFMAs
                    const unsigned ab = aBias + (unsigned)h * 128u;
                    bq[0] = q2_read_f32x4<0>(ab); bq[1] = q2_read_f32x4<32>(ab); bq[2] = q2_read_f32x4<64>(ab); bq[3] = q2_read_f32x4<96>(ab);
                    bk[0] = q2_read_f32x4<H * 4>(ab); bk[1] = q2_read_f32x4<H * 4 + 32>(ab);
                    bk[2] = q2_read_f32x4<H * 4 + 64>(ab); bk[3] = q2_read_f32x4<H * 4 + 96>(ab);
                    q2_wait14(F[par]);
                    static_for<6>([&](auto m_tag) __attribute__((always_inline)) {      // last k-tile, matrix by matrix: Q Q K K V V ...
                        constexpr int m = decltype(m_tag)::value;
                        mma(PAR{}, std::integral_constant<int, m % 2>{}, std::integral_constant<int, m / 2>{}, std::integral_constant<int, 2 * hh + m % 2>{});
                        pieces(hh_tag, m_tag);
                    });
                } else {
                    q2_wait0_bias(F[par], bq, bk);
                    // ... Q Q (Q is complete: its conversion runs under the K and V MFMAs) K K (the same for K) V V
                    static_for<2>([&](auto i) __attribute__((always_inline)) { mma(PAR{}, i, std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * hh + decltype(i)::value>{}); });
                    pieces(hh_tag, std::integral_constant<int, 1>{});
                    static_for<2>([&](auto i) __attribute__((always_inline)) { mma(PAR{}, i, std::integral_constant<int, 1>{}, std::integral_constant<int, 2 * hh + decltype(i)::value>{}); });
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) {
                        f16x4 oq;
#pragma unroll
                        for (int e = 0; e < 4; ++e) oq[e] = (_Float16)(acc[0][4 * gg + e] + bq[gg][e]);
                        *(f16x4 *)(QS + q2_off32(slot, gg) + hi * 8) = oq;
                    }
                    static_for<2>([&](auto i) __attribute__((always_inline)) { mma(PAR{}, i, std::integral_constant<int, 2>{}, std::integral_constant<int, 2 * hh + decltype(i)::value>{}); });
                    pieces(hh_tag, std::integral_constant<int, 4>{});
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) {
                        f16x4 ok;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ok[e] = (_Float16)(acc[1][4 * gg + e] + bk[gg][e]);
                        *(f16x4 *)(KS + q2_off32(slot, gg) + hi * 8) = ok;
                    }
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) {
                        f16x4 ov;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[e] = (_Float16)(acc[2][4 * gg + e] + bv);
                        *(f16x4 *)(VT + l31 * Q2_VT_LD + blk * 32 + 8 * gg + 4 * hi) = ov;
                    }
                }
            });
            // ---- Q_h, K_h (row-major, swizzled) and V_h^T are in copy h & 1 (the attention waves are two heads behind at
            // most: they finished head h-2 before they passed the end of head h-1).  One barrier publishes them and opens the
            // next head's first slab; in flight behind it: this wave's pieces of the next head's second slab
            if (h < 8) TL_STAMP_AT(tl_sel, 2 + 6 * h + 4);
            q2_head_barrier<Q4 ? 63 : PPS>();
            if (h < 8) TL_STAMP_AT(tl_sel, 2 + 6 * h + 5);
            rslot = rslot == 2 ? 0 : rslot + 1;
            set_slot();
            read_half(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        }
        TL_STAMP_AT(tl_sel, 60);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the dead prefetches must not outlive the LDS allocation
    } else {
        // =============================== attention wave: query block `blk` ===============================
        // key tiles by class, the same for every head: needed by some query of the block / inside every query's sentence
        unsigned need = 0, inner = 0;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const bool nd = gtok >= 0 && k0 < 32 * kt + 32 && k1 > 32 * kt;
            const bool in = gtok < 0 || (k0 <= 32 * kt && k1 >= 32 * kt + 32);
            if (__any(nd)) need |= 1u << kt;
            if (__all(in)) inner |= 1u << kt;
        }
        need = __builtin_amdgcn_readfirstlane(need);
        inner = __builtin_amdgcn_readfirstlane(inner);
        const unsigned klen = (unsigned)(k1 - k0);
        const int kbase = 4 * hi - k0;
        const float sc = 1.44269504088896340736f / __builtin_sqrtf(32.0f);   // log2(e) / sqrt(d)
        for (int i = tid; i < 2 * H; i += 256) BS[i] = a.bias[i];             // Q and K bias -> LDS (these waves have nothing else to do yet)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the projection waves' first barrier

        if constexpr (Q2_PRIO > 0) __builtin_amdgcn_s_setprio(Q2_PRIO);
        [[maybe_unused]] const bool tl_sel = tid == 0;
        // attention of head hh (copy hh & 1 of Q / K / V^T).  WB: the slab barrier of the head the projection waves work on
        // meanwhile is executed here, after step Q2_PM
        // FAST: every key tile is needed and inside every query's sentence (full windows): straight-line code, no masks
        auto attend = [&](int hh, auto wb_tag, auto fast_tag, int tb) __attribute__((always_inline)) {
            constexpr bool WB = decltype(wb_tag)::value, FAST = decltype(fast_tag)::value;
            const char *QS = QKV + (hh & 1) * QKV_BYTES, *KS = QS + Q2_WIN * 64;
            const half_t *VT = (const half_t *)(KS + Q2_WIN * 64);
            f32x16 s[4];
            // ---- S^T = K Q^T, mask, row maximum (of the raw scores: the scale is positive)
            f16x8 qf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) qf[kk] = *(const f16x8 *)(QS + q2_off32(slot, kk * 2 + hi));
            float mx = -INFINITY;
            // (opaque copies: left alone the compiler hoists the 64 mask comparisons out of the head loop into SGPR pairs,
            // spills them to VGPR lanes and reads them back with two v_readlane per element)
            int kb = kbase;
            unsigned kl = klen;
            if constexpr (!FAST) asm volatile("" : "+v"(kb), "+v"(kl));
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (FAST || (need & (1u << kt))) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const f16x8 kf = *(const f16x8 *)(KS + q2_off32(kt * 32 + l31, kk * 2 + hi));
                        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], s[kt], 0, 0, 0);
                    }
                    if (FAST || (inner & (1u << kt))) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            // key = kt*32 + (r&3) + 8*(r>>2) + 4*hi is in [k0, k1)
                            const bool ok = (unsigned)(kt * 32 + (r & 3) + 8 * (r >> 2) + kb) < kl;
                            const float v = ok ? s[kt][r] : -INFINITY;
                            s[kt][r] = v;
                            mx = fmaxf(mx, v);
                        }
                    }
                }
                if constexpr (WB) {
                    if (kt == Q2_PM) {
                        if (tb < 8) TL_STAMP_AT(tl_sel, 128 + 6 * tb + 1);
                        asm volatile("s_barrier" ::: "memory");
                        if (tb < 8) TL_STAMP_AT(tl_sel, 128 + 6 * tb + 2);
                    }
                }
            }
            mx = xor32_max(mx) * sc;          // = the maximum of the scaled scores, as attention.hip
            mx = fmaxf(mx, -3.0e38f);                         // empty slots (no keys): keeps exp2(-inf - mx) = 0, no NaN
            // ---- softmax (one fma + exp2 per score), O^T = V^T P^T
            float psum = 0.f;
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (FAST || (need & (1u << kt))) {
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        // (softmax_p8, kernels.h: fp16 argument and exponential like the reference's table, f32 row sum)
                        const f16x8 pf = softmax_p8(s[kt][8 * st], s[kt][8 * st + 1], s[kt][8 * st + 2], s[kt][8 * st + 3], s[kt][8 * st + 4],
                                                    s[kt][8 * st + 5], s[kt][8 * st + 6], s[kt][8 * st + 7], sc, mx, psum);
                        const int key0 = kt * 32 + 16 * st + 4 * hi;              // keys key0..+3 and key0+8..+11
                        const half_t *vr = VT + l31 * Q2_VT_LD + key0;
                        const f16x4 v0 = *(const f16x4 *)vr, v1 = *(const f16x4 *)(vr + 8);
                        f16x8 vf;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
                        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o, 0, 0, 0);
                    }
                }
                if constexpr (WB) {
                    if (kt + 4 == Q2_PM) {
                        if (tb < 8) TL_STAMP_AT(tl_sel, 128 + 6 * tb + 1);
                        asm volatile("s_barrier" ::: "memory");
                        if (tb < 8) TL_STAMP_AT(tl_sel, 128 + 6 * tb + 2);
                    }
                }
            }
            // ---- normalise, store
            psum = xor32_sum(psum);
            if (gtok >= 0) {
                const float inv = 1.0f / psum;
                half_t *op = a.out + (size_t)gtok * H + hh * 32;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f16x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (_Float16)rounded_f32(o[4 * gq + e] * inv);
                    *(f16x4 *)(op + 8 * gq + 4 * hi) = ov;
                }
            }
        };
        const bool fast = need == 0xFu && inner == 0xFu;        // wave-uniform
        // the projection waves' barriers of head 0 (nothing to attend yet), then head h-1 is attended while head h is
        // projected, then the last head
        asm volatile("s_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        for (int h = 1; h < n_head; ++h) {
            if (h < 8) TL_STAMP_AT(tl_sel, 128 + 6 * h);
            if (fast) attend(h - 1, std::true_type{}, std::true_type{}, h);
            else attend(h - 1, std::true_type{}, std::false_type{}, h);
            if (h < 8) TL_STAMP_AT(tl_sel, 128 + 6 * h + 5);
            asm volatile("s_barrier" ::: "memory");             // end of head h: it is published
        }
        TL_STAMP_AT(tl_sel, 190);
        if (fast) attend(n_head - 1, std::false_type{}, std::true_type{}, 0);
        else attend(n_head - 1, std::false_type{}, std::false_type{}, 0);
        TL_STAMP_AT(tl_sel, 192);
        TL_STAMP_AT(tl_sel, 252);
        TL_REALTIME_AT(tl_sel, 253);
    }
}

template <int KT, int GB, int WT>
__global__ __launch_bounds__(512, 2) void qkv_attention2_kernel(Qkv2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    qkv_attention2_body<KT, GB, WT>(a, smem, (int)blockIdx.x, (int)threadIdx.x);
}

#ifndef BERT_HIP_PHASES_ONLY
// ---------------------------------------------------------------------------------------------------------------------
// Next-fit windows built ON THE DEVICE (the asynchronous device API has the sentence lengths in HBM only): the same rule as
// Engine::build_windows — sentences in order, each starting at a multiple of 16 slots, the open window is closed when the next
// sentence does not fit into its 128 slots.  Next-fit is a sequential automaton, but its state is tiny (the fill of the open
// window: 0, 16, ... 128 slots), so it parallelises as a scan over state maps: every thread runs its block of sentences from
// ALL nine entry states at once and records where each ends up, how many windows it closes and which sentence opened the
// window it leaves open; thread 0 chains the per-block maps; every thread then replays its block from its true entry state
// and writes the windows it closes.
constexpr int BW_THREADS = 512;

// SLOT: the place granularity (16; 8 with BERT_HIP_WINDOW_SLOTS=8): 128 / SLOT + 1 fill states
template <int SLOT>
__global__ __launch_bounds__(BW_THREADS) void build_windows_kernel(const int32_t *__restrict__ cu, int B, int2 *__restrict__ windows,
                                                                   int *__restrict__ n_windows) {
    constexpr int BW_STATES = 128 / SLOT + 1;
    __shared__ unsigned char exit_state[BW_THREADS][BW_STATES];   // fill / SLOT after the block
    __shared__ int closes[BW_THREADS][BW_STATES];                 // windows closed inside the block
    __shared__ int opened[BW_THREADS][BW_STATES];                 // first sentence of the window left open; -1: the entry window
    __shared__ int entry_fill[BW_THREADS], entry_first[BW_THREADS], entry_index[BW_THREADS];
    const int t = threadIdx.x, per = (B + BW_THREADS - 1) / BW_THREADS, n_blocks = per ? (B + per - 1) / per : 0;
    const int b0 = min(B, t * per), b1 = min(B, b0 + per);

    int fill[BW_STATES], n_closed[BW_STATES], first[BW_STATES];
#pragma unroll
    for (int e = 0; e < BW_STATES; ++e) { fill[e] = e * SLOT; n_closed[e] = 0; first[e] = -1; }
    for (int b = b0; b < b1; ++b) {
        const int n = cu[b + 1] - cu[b];
#pragma unroll
        for (int e = 0; e < BW_STATES; ++e) {
            if (fill[e] == 0) first[e] = b;                       // an empty window opens at this sentence
            else if (fill[e] + n > 128) { ++n_closed[e]; first[e] = b; fill[e] = 0; }
            fill[e] = min(128, (fill[e] + n + SLOT - 1) & ~(SLOT - 1));         // (no sentence is longer than a window here)
        }
    }
#pragma unroll
    for (int e = 0; e < BW_STATES; ++e) {
        exit_state[t][e] = (unsigned char)(fill[e] / SLOT); closes[t][e] = n_closed[e]; opened[t][e] = first[e];
    }
    __syncthreads();
    if (t == 0) {
        int e = 0, open_first = 0, index = 0;
        for (int k = 0; k < n_blocks; ++k) {
            entry_fill[k] = e * SLOT; entry_first[k] = open_first; entry_index[k] = index;
            index += closes[k][e];
            if (opened[k][e] >= 0) open_first = opened[k][e];
            e = exit_state[k][e];
        }
        if (B > 0) windows[index++] = make_int2(open_first, B - open_first);   // the window still open after the last sentence
        *n_windows = index;
    }
    __syncthreads();
    if (b0 < b1) {
        int f = entry_fill[t], open_first = entry_first[t], index = entry_index[t];
        for (int b = b0; b < b1; ++b) {
            const int n = cu[b + 1] - cu[b];
            if (f == 0) open_first = b;
            else if (f + n > 128) { windows[index++] = make_int2(open_first, b - open_first); open_first = b; f = 0; }
            f = min(128, (f + n + SLOT - 1) & ~(SLOT - 1));
        }
    }
}

// The place granularity of the windows — the process-wide DEFAULT (BERT_HIP_WINDOW_SLOTS / bert_hip_set_option "window_slots"); a
// forward pass reads it ONCE (Engine::eval_packed_*) and hands that value to the window builders, the grid bound and the
// launchers, so a change from another thread or context cannot land between a window list and the kernel that places by it: 16 — one k-step
// of the P·V MFMAs, so that a sentence's bits do not depend on where it sits in a window — or 8: a quarter fewer windows for
// mean-25-token batches, and bits that depend on a sentence's place (tools/ubench/mfma_shift.hip).
static std::atomic<int> g_window_slots{16};
int window_slots() { return g_window_slots.load(std::memory_order_relaxed); }
void set_window_slots(int slots) { g_window_slots.store(slots == 8 ? 8 : 16, std::memory_order_relaxed); }

void launch_build_windows(const int32_t *cu_seqlens, int n_sentences, int2 *windows, int *n_windows, int slots, hipStream_t stream) {
    if (slots == 8) BERT_LAUNCH(build_windows_kernel<8>, dim3(1), dim3(BW_THREADS), 0, stream, cu_seqlens, n_sentences, windows, n_windows);
    else BERT_LAUNCH(build_windows_kernel<16>, dim3(1), dim3(BW_THREADS), 0, stream, cu_seqlens, n_sentences, windows, n_windows);
}

int qkv_attention2_max_windows(int n_sentences, int n_tokens, int slot) {
    // two consecutive next-fit windows hold more than 128 slots together
    const long long slots = (long long)n_tokens + (long long)(slot - 1) * n_sentences;
    const long long bound = 2 * (slots / 128) + 2;
    return (int)(bound < n_sentences ? bound : n_sentences);
}

bool qkv_attention2_supported(const GemmWeight &Wqkv, int n_head, int d_head, int max_len) {
    const int H = n_head * d_head;
    return (Wqkv.type == GW_F16 ? Wqkv.w16 != nullptr : Wqkv.qs != nullptr) && d_head == 32 && Wqkv.K == H && Wqkv.N == 3 * H && (H == 128 || H == 256 || H == 384) &&
           max_len <= Q2_WIN && max_len > 0;
}

int qkv_attention2_sentences_per_window(int max_len, int slots) { const int m = slots - 1; return Q2_WIN / ((max_len + m) & ~m); }

void launch_qkv_attention2(const GemmWeight &Wqkv, const half_t *x, const float *bias, const int32_t *cu_seqlens,
                           int n_sentences, const int2 *groups, int n_groups, const int *n_groups_dev, int max_len, int n_head,
                           int slots, half_t *out, hipStream_t stream) {
    Qkv2Args a;
    a.x = x; a.w = Wqkv.w16; a.qs = Wqkv.qs; a.sc = Wqkv.sc; a.bias = bias; a.cu = cu_seqlens; a.groups = groups; a.n_groups = n_groups_dev; a.out = out;
    a.n_head = n_head; a.n_sent = n_sentences; a.slot_mask = slots - 1;
    a.spw = qkv_attention2_sentences_per_window(max_len, slots);
    const int grid = groups ? n_groups : (n_sentences + a.spw - 1) / a.spw;
    const int KT = Wqkv.K / 64, GB = KT / 2;
    const size_t lds = (size_t)3 * GB * Q2_TILE + 2 * (2 * Q2_WIN * 64 + 32 * Q2_VT_LD * 2) + (size_t)2 * Wqkv.K * sizeof(float);
    static DeviceFlags configured[3][8];
    auto go = [&](auto kernel) {
        configure_once(configured[Wqkv.type][KT], [&] { (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        BERT_LAUNCH(kernel, dim3(grid), dim3(512), lds, stream, a);
        TL_DUMP_RAW(grid >= 256, 256);
    };
    switch (KT * 4 + Wqkv.type) {
        case 2 * 4 + GW_F16: go(qkv_attention2_kernel<2, 1, GW_F16>); break;
        case 4 * 4 + GW_F16: go(qkv_attention2_kernel<4, 2, GW_F16>); break;
        case 6 * 4 + GW_F16: go(qkv_attention2_kernel<6, 3, GW_F16>); break;
        case 2 * 4 + GW_Q4_0: go(qkv_attention2_kernel<2, 1, GW_Q4_0>); break;
        case 4 * 4 + GW_Q4_0: go(qkv_attention2_kernel<4, 2, GW_Q4_0>); break;
        case 6 * 4 + GW_Q4_0: go(qkv_attention2_kernel<6, 3, GW_Q4_0>); break;
        case 2 * 4 + GW_Q4_1: go(qkv_attention2_kernel<2, 1, GW_Q4_1>); break;
        case 4 * 4 + GW_Q4_1: go(qkv_attention2_kernel<4, 2, GW_Q4_1>); break;
        default: go(qkv_attention2_kernel<6, 3, GW_Q4_1>); break;
    }
}

#endif  // BERT_HIP_PHASES_ONLY

}  // namespace bert_hip
