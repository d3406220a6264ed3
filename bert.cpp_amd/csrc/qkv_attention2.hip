// qkv_attention2.hip — Q|K|V projection and self-attention of a 128-slot token WINDOW in one kernel (gfx950, d_head = 32,
// f16 weights, H = 128 / 256 / 384).  Second generation of qkv_attention.hip (reference bert.cpp:822-856).
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
// same number of MFMAs (72 + 16 per head).  Per head the two groups meet at NBAR+1 barriers: the slab barriers (the
// last one doubles as "attention is done with the previous head's Q/K/V^T") and "Q/K/V^T of this head are published".
//
// Windows and bit-exactness: sentence j of the window starts at slot off_j with off_0 = 0, off_{j+1} = off_j + n_j
// rounded up to 16.  A softmax row only ever sees the keys of its own sentence (the others are masked with -inf before
// the running maximum, their probabilities are exact zeros), and because 16 slots are one k-step of the P V MFMAs a
// sentence's keys occupy the same k positions inside every instruction wherever the sentence sits: the zeros in front
// of them add exactly nothing, so a sentence gives the same bits in any window, alone or not (tested).  Key tiles
// that no query of a block needs are skipped, tiles that lie inside the sentence of every query of the block are not
// masked at all.
#include "tile_stream.h"

namespace bert_hip {

namespace {

constexpr int Q2_WIN = 128;                      // token slots per workgroup
constexpr int Q2_TILE = 12288;                   // [96 rows x 64 halfs]: Q_h, K_h, V_h rows of one k-tile
constexpr int Q2_VT_LD = Q2_WIN + 4;             // halfs per V^T row (8-byte skew: conflict-free ds_read_b64)

struct Qkv2Args {
    const half_t *x;         // [T_pad][H] hidden state, packed sentences
    const half_t *w;         // [3H (padded)][H] f16: Q rows, K rows, V rows
    const float *bias;       // [3H]
    const int32_t *cu;       // [n_sent + 1]
    const int2 *groups;      // per workgroup {first sentence, count}; nullptr: `spw` sentences per workgroup
    half_t *out;             // [T_pad][H] attention context
    int n_head, n_sent, spw;
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
__device__ __forceinline__ void q2_publish_barrier(f16x8 (&f)[2][3]) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier"
                 : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[0][2]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[1][2]) : : "memory");
}

}  // namespace

// H = 64 * KT = 32 * n_head; a slab = GB k-tiles, NBAR = KT / GB slabs per head
template <int KT, int GB>
__global__ __launch_bounds__(512, 2) void qkv_attention2_kernel(Qkv2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int H = 64 * KT, NBAR = KT / GB, SLAB = GB * Q2_TILE, PPS = 3 * GB;   // PPS = DMA pieces per slab and wave
    static_assert(KT % GB == 0 && NBAR >= 1 && NBAR <= 2, "");
    char *RING = smem;                                        // 3 slabs
    char *QS = smem + 3 * SLAB;                               // [128][32] halfs, q2_off32 swizzle
    char *KS = QS + Q2_WIN * 64;
    half_t *VT = (half_t *)(KS + Q2_WIN * 64);                // [32][Q2_VT_LD]
    float *BS = (float *)((char *)VT + 32 * Q2_VT_LD * 2);    // [3H] bias

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int blk = wave & 3;                                 // token block (projection) / query block (attention)
    const int n_head = a.n_head;

    // ---- the window: sentences first .. first+count-1, sentence j at slots [off_j, off_j + n_j)
    int first, count;
    if (a.groups) { const int2 g = a.groups[blockIdx.x]; first = g.x; count = g.y; }
    else { first = blockIdx.x * a.spw; count = min(a.spw, a.n_sent - first); }
    if (count <= 0) return;
    const int slot = blk * 32 + l31;
    int gtok = -1, k0 = 0, k1 = 0;                            // this lane's slot: global token, key range of its sentence
    {
        int off = 0;
        for (int j = 0; j < count; ++j) {
            const int t0 = a.cu[first + j], n = a.cu[first + j + 1] - t0;
            if (slot >= off && slot < off + n) { gtok = t0 + slot - off; k0 = off; k1 = off + n; }
            off = (off + n + 15) & ~15;
        }
    }
    for (int i = tid; i < 3 * H; i += 512) BS[i] = a.bias[i];

    if (wave >= 4) {
        // =============================== projection wave: token block `blk` ===============================
        const int wp = blk;
        // rows of the hidden state as MFMA fragments (token = l31, k = 16 ks + 8 hi ..): B operand of the Q / K
        // projections, A operand of the V projection.  Empty slots read the window's first token (finite values).
        f16x8 bf[4 * KT];
        {
            const int gt = gtok >= 0 ? gtok : a.cu[first];
            const half_t *xr = a.x + (size_t)gt * H + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < 4 * KT; ++ks) bf[ks] = *(const f16x8 *)(xr + 16 * ks);
        }
        // DMA: piece (i) of a slab for this wave = tile i / 3, row block rb = i % 3 (Q, K, V rows), rows wp*8 .. +8 of it
        const unsigned loff = (unsigned)(((wp * 8 + (lane >> 3)) * H + (((lane & 7) ^ (((wp & 1) << 2) | ((lane >> 4) & 3))) * 8)) * 2);
        const int S = n_head * NBAR;                          // slabs in total
        auto dma_piece = [&](int sl, int dslot, auto i_tag) __attribute__((always_inline)) {
            constexpr int i = decltype(i_tag)::value, t = i / 3, rb = i % 3;
            const int h = sl / NBAR, j = sl - h * NBAR;
            const char *src = (const char *)a.w + ((size_t)(rb * H + h * 32) * H + (size_t)(j * GB + t) * 64) * 2;
            __builtin_amdgcn_global_load_lds(AS_GLOBAL(src + loff), AS_LDS(RING + dslot * SLAB + t * Q2_TILE + rb * 4096 + wp * 1024), 16, 0, 0);
        };
        static_for<PPS>([&](auto i) __attribute__((always_inline)) { dma_piece(0, 0, i); });
        static_for<PPS>([&](auto i) __attribute__((always_inline)) { dma_piece(S > 1 ? 1 : 0, 1, i); });

        // per-lane LDS address of the weight fragment of k-step kk of a tile: the chunk swizzle is an XOR of 2*kk + hi
        const unsigned aX0 = lds_addr(RING) + off64(l31, hi);
        unsigned aS[4];
        int rslot = 0, sl = 0;                                // ring slot and index of the slab being multiplied
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
        // x rows, slab 0 and (compiler: the ordinary loads above are waited for with vmcnt(0)) slab 1 have landed
        [[maybe_unused]] const bool tl_sel = tid == 256;
        TL_STAMP_AT(tl_sel, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        TL_STAMP_AT(tl_sel, 1);
        read_half(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});

        for (int h = 0; h < n_head; ++h) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
            static_for<2 * KT>([&](auto hh_tag) __attribute__((always_inline)) {
                constexpr int hh = decltype(hh_tag)::value, kt = hh >> 1, half = hh & 1, par = hh & 1;
                constexpr int nhh = (hh + 1) % (2 * KT), ntt = (nhh >> 1) % GB, nhalf = nhh & 1;
                constexpr bool crossing = half == 1 && (kt + 1) % GB == 0;        // the next half opens a new slab
                constexpr int hs = hh % (2 * GB);                                  // half index inside the slab
                // the pieces of slab sl+2 go behind the MFMAs of slab sl (piece i in half (2 i) / 3 of the slab), into the
                // slot slab sl-1 was read from.  (Taken before the crossing update: this half still belongs to slab sl.)
                const int sreq = min(sl + 2, S - 1), dslot = rslot == 0 ? 2 : rslot - 1;
                if constexpr (crossing) {
                    if (h < 8) TL_STAMP_AT(tl_sel, 2 + 6 * h + (hh == 2 * KT - 1 ? 2 : 0));
                    // outstanding here, oldest first: the next slab (complete), then the slab after it WITHOUT the one
                    // piece this last half is about to request
                    q2_slab_barrier<PPS - 1>(F[par]);
                    if (h < 8) TL_STAMP_AT(tl_sel, 2 + 6 * h + (hh == 2 * KT - 1 ? 3 : 1));
                    rslot = rslot == 2 ? 0 : rslot + 1;
                    ++sl;
                    set_slot();
                }
                read_half(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, ntt>{}, std::integral_constant<int, nhalf>{});
                q2_wait6(F[par]);
                static_for<6>([&](auto m_tag) __attribute__((always_inline)) {
                    constexpr int m = decltype(m_tag)::value, i = m / 3, rb = m % 3;
                    constexpr int ks = 2 * hh + i;
                    if constexpr (rb < 2) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[par][i][rb], bf[ks], acc[rb], 0, 0, 0);
                    else acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[ks], F[par][i][2], acc[2], 0, 0, 0);
                    // pieces of this half: i0 = ceil(3 hs / 2) .. (3 (hs + 1) + 1) / 2 - 1
                    constexpr int i0 = (3 * hs + 1) / 2, i1 = (3 * (hs + 1) + 1) / 2;
                    if constexpr (m == 1) dma_piece(sreq, dslot, std::integral_constant<int, i0>{});
                    if constexpr (m == 4 && i1 - i0 == 2) dma_piece(sreq, dslot, std::integral_constant<int, i0 + 1>{});
                });
            });
            // ---- publish Q_h, K_h (row-major, swizzled) and V_h^T; the last slab barrier above also told us that the
            // attention waves are done with the previous head's copies
            {
                const float *bq = BS + h * 32 + 4 * hi, *bk = BS + H + h * 32 + 4 * hi;
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const f32x4 b0 = *(const f32x4 *)(bq + 8 * gg), b1 = *(const f32x4 *)(bk + 8 * gg);
                    f16x4 oq, ok;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        oq[e] = (_Float16)(acc[0][4 * gg + e] + b0[e]);
                        ok[e] = (_Float16)(acc[1][4 * gg + e] + b1[e]);
                    }
                    *(f16x4 *)(QS + q2_off32(slot, gg) + hi * 8) = oq;
                    *(f16x4 *)(KS + q2_off32(slot, gg) + hi * 8) = ok;
                }
                const float bv = BS[2 * H + h * 32 + l31];
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    f16x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (_Float16)(acc[2][4 * gg + e] + bv);
                    *(f16x4 *)(VT + l31 * Q2_VT_LD + blk * 32 + 8 * gg + 4 * hi) = ov;
                }
            }
            if (h < 8) TL_STAMP_AT(tl_sel, 2 + 6 * h + 4);
            q2_publish_barrier(F[0]);
            if (h < 8) TL_STAMP_AT(tl_sel, 2 + 6 * h + 5);
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
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the projection waves' first barrier

        f32x16 s[4];
        float mx = 0.f;
        auto part1 = [&]() __attribute__((always_inline)) {                // S^T = K Q^T, scale, mask, row maximum
            f16x8 qf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) qf[kk] = *(const f16x8 *)(QS + q2_off32(slot, kk * 2 + hi));
            mx = -INFINITY;
            // (opaque copies: left alone the compiler hoists the 64 mask comparisons out of the head loop into SGPR pairs,
            // spills them to VGPR lanes and reads them back with two v_readlane per element)
            int kb = kbase;
            unsigned kl = klen;
            asm volatile("" : "+v"(kb), "+v"(kl));
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (!(need & (1u << kt))) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const f16x8 kf = *(const f16x8 *)(KS + q2_off32(kt * 32 + l31, kk * 2 + hi));
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], s[kt], 0, 0, 0);
                }
                if (inner & (1u << kt)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = s[kt][r] * sc;
                        s[kt][r] = v;
                        mx = fmaxf(mx, v);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // key = kt*32 + (r&3) + 8*(r>>2) + 4*hi is in [k0, k1)
                        const bool ok = (unsigned)(kt * 32 + (r & 3) + 8 * (r >> 2) + kb) < kl;
                        const float v = ok ? s[kt][r] * sc : -INFINITY;
                        s[kt][r] = v;
                        mx = fmaxf(mx, v);
                    }
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            mx = fmaxf(mx, -3.0e38f);                         // empty slots (no keys): keeps exp2(-inf - mx) = 0, no NaN
        };
        auto part2 = [&](int hh) __attribute__((always_inline)) {           // softmax, O^T = V^T P^T, normalise, store
            float psum = 0.f;
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (!(need & (1u << kt))) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(s[kt][r] - mx);
                    s[kt][r] = pv;
                    psum += pv;
                }
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    f16x8 pf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[e] = (_Float16)s[kt][8 * st + e];
                    const int key0 = kt * 32 + 16 * st + 4 * hi;              // keys key0..+3 and key0+8..+11
                    const half_t *vr = VT + l31 * Q2_VT_LD + key0;
                    const f16x4 v0 = *(const f16x4 *)vr, v1 = *(const f16x4 *)(vr + 8);
                    f16x8 vf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
                    o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o, 0, 0, 0);
                }
            }
            psum += __shfl_xor(psum, 32);
            if (gtok >= 0) {
                const float inv = 1.0f / psum;
                half_t *op = a.out + (size_t)gtok * H + hh * 32;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f16x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (_Float16)(o[4 * gq + e] * inv);
                    *(f16x4 *)(op + 8 * gq + 4 * hi) = ov;
                }
            }
        };
        // head h is attended while head h+1 is projected: NBAR - 1 slab barriers in the middle, then "done with
        // Q/K/V^T" (= the projection waves' last slab barrier of the head) and "published"
        [[maybe_unused]] const bool tl_sel = tid == 0;
        for (int h = 0; h < n_head; ++h) {
            if (h < 8) TL_STAMP_AT(tl_sel, 128 + 6 * h);
            if (h > 0) part1();
            if (h < 8) TL_STAMP_AT(tl_sel, 128 + 6 * h + 1);
            if constexpr (NBAR == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (h < 8) TL_STAMP_AT(tl_sel, 128 + 6 * h + 2);
            if (h > 0) part2(h - 1);
            if (h < 8) TL_STAMP_AT(tl_sel, 128 + 6 * h + 3);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (h < 8) TL_STAMP_AT(tl_sel, 128 + 6 * h + 4);
            asm volatile("s_barrier" ::: "memory");
            if (h < 8) TL_STAMP_AT(tl_sel, 128 + 6 * h + 5);
        }
        TL_STAMP_AT(tl_sel, 190);
        part1();
        TL_STAMP_AT(tl_sel, 191);
        part2(n_head - 1);
        TL_STAMP_AT(tl_sel, 192);
    }
}

bool qkv_attention2_supported(const GemmWeight &Wqkv, int n_head, int d_head, int max_len) {
    const int H = n_head * d_head;
    return Wqkv.type == GW_F16 && d_head == 32 && Wqkv.K == H && Wqkv.N == 3 * H && (H == 128 || H == 256 || H == 384) &&
           max_len <= Q2_WIN && max_len > 0;
}

int qkv_attention2_sentences_per_window(int max_len) { return Q2_WIN / ((max_len + 15) & ~15); }

void launch_qkv_attention2(const GemmWeight &Wqkv, const half_t *x, const float *bias, const int32_t *cu_seqlens,
                           int n_sentences, const int2 *groups, int n_groups, int max_len, int n_head, half_t *out,
                           hipStream_t stream) {
    Qkv2Args a;
    a.x = x; a.w = Wqkv.w16; a.bias = bias; a.cu = cu_seqlens; a.groups = groups; a.out = out;
    a.n_head = n_head; a.n_sent = n_sentences;
    a.spw = qkv_attention2_sentences_per_window(max_len);
    const int grid = groups ? n_groups : (n_sentences + a.spw - 1) / a.spw;
    const int KT = Wqkv.K / 64, GB = KT / 2;
    const size_t lds = (size_t)3 * GB * Q2_TILE + 2 * Q2_WIN * 64 + 32 * Q2_VT_LD * 2 + (size_t)3 * Wqkv.K * sizeof(float);
    auto go = [&](auto kernel) {
        (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), lds, stream, a);
        TL_DUMP_RAW(grid >= 256, 200);
    };
    switch (KT) {
        case 2: go(qkv_attention2_kernel<2, 1>); break;
        case 4: go(qkv_attention2_kernel<4, 2>); break;
        default: go(qkv_attention2_kernel<6, 3>); break;
    }
}

}  // namespace bert_hip
