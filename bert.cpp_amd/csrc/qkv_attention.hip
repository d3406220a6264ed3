// qkv_attention.hip — Q|K|V projection and self-attention of one sentence in ONE kernel (gfx950, d_head = 32).
//
// Replaces, for sentences of up to 128 tokens, the pair panel_store_kernel (reference bert.cpp:822-839: three
// ggml_mul_mat + bias) and attention_mfma_kernel (bert.cpp:841-856: K Q^T, scale, soft_max, V^T P, merge heads).
// The [tokens][3H] Q|K|V activation never exists in HBM (it is 3x the layer's hidden state: written once and read
// once per layer, it was the largest HBM stream of the forward pass); per layer the kernel reads the hidden
// state once and writes the attention context once.
//
// One workgroup = one sentence.  Its n <= 128 rows of the hidden state are brought into LDS once (96 KiB for
// H = 384, the MFMA B operand of every head).  Then the waves specialise:
//   waves 4,5,6  "projection" waves: wave g computes one of Q_h, K_h, V_h ([32 features] x [128 tokens], K = H)
//                for head h from its PRIVATE 32-row weight tiles, streamed by LDS-DMA through a private 3-slot
//                ring and retired with counted s_waitcnt vmcnt only — no barrier inside the head;
//   waves 0..3   "attention" waves: wave a owns query block a (32 queries) of the head projected one step
//                earlier: S^T = K Q^T, softmax over the keys, O^T = V^T P^T, normalise, store (as attention.hip);
//   wave 7       takes part in the barriers only.
// Head h+1 is projected while head h is attended; the two groups meet at two barriers per head, between which the
// projection waves publish Q_h (row-major, swizzled), K_h (same) and V_h^T in 24 KiB of LDS.  Each SIMD hosts one
// wave of either kind, so the projection MFMAs run under the softmax VALU work of the attention wave.
//
// MFMA roles: Q and K waves use the weights as the A operand (accumulator rows = features: a lane owns 4
// consecutive features of one token -> 8-byte row-major LDS writes); the V wave swaps the operands (accumulator
// rows = tokens: a lane owns 4 consecutive keys of one feature -> 8-byte writes into V^T).  The arithmetic (k
// order, f32 accumulation, f16 rounding points, exp2-based softmax) is the same as in the two kernels it
// replaces, so both paths give the same bits.
#include "tile_stream.h"

#include <type_traits>
#include <utility>

namespace bert_hip {

namespace {

constexpr int QA_TOK = 128;                      // tokens per workgroup (one sentence)
constexpr int QA_WSLOT = 4096;                   // one private weight tile: 32 rows x 64 halfs
constexpr int QA_VT_LD = QA_TOK + 4;             // halfs per V^T row (8-byte skew: conflict-free ds_read_b64)

struct QkvAttArgs {
    const half_t *x;         // [T_pad][H] hidden state (rows of a sentence are contiguous)
    const half_t *w;         // [3H (padded)][H] f16: Q rows, K rows, V rows                       (WT == GW_F16)
    const uint4 *qs;         // q4 nibble plane, tile-contiguous (kernels.h GemmWeight)             (WT != GW_F16)
    const void *sc;          // q4 scale plane
    const float *bias;       // [3H]
    const int32_t *cu;       // [n_sentences + 1]
    half_t *out;             // [T_pad][H] attention context
    int n_head;
};

// 16-byte chunk swizzle of a [rows][32 halfs] tile (64-byte rows) for conflict-free ds_read_b128
__device__ __forceinline__ int off32(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// ds_read_b128 the compiler does not track: the caller retires it with its own s_waitcnt lgkmcnt(N)
template <int OFF>
__device__ __forceinline__ f16x8 lds_read_b128(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

__device__ __forceinline__ void wait_lgkm10(f16x8 &a0, f16x8 &a1, f16x8 &a2, f16x8 &a3, f16x8 &a4, f16x8 &a5, f16x8 &a6,
                                            f16x8 &a7, f16x8 &a8, f16x8 &a9) {
    asm volatile("s_waitcnt lgkmcnt(10)"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9));
}

// Global loads the compiler does not track (its own wait for a tracked load would be vmcnt(0) and drain the
// weight tiles in flight); retired by the counted wait of the k-tile loop, see wait_vm4_bias.
__device__ __forceinline__ void gload_untracked(f32x4 &v, const float *p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p));
}
__device__ __forceinline__ void gload_untracked(float &v, const float *p) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p));
}
// q4 block of this lane (16 B of nibbles + scale) as untracked loads, and the counted wait that hands a fetched tile
// (and, on a head's last tile, the bias registers) to its consumers
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Q4Pend { u32x4 q; unsigned sc; };             // native vector types: asm operands
template <int WT>
__device__ __forceinline__ void q4_fetch_untracked(Q4Pend &r, const uint4 *qs, const void *sc, size_t bi) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.q) : "v"(qs + bi));
    if (WT == GW_Q4_0) asm volatile("global_load_ushort %0, %1, off" : "=v"(r.sc) : "v"((const unsigned short *)sc + bi));
    else asm volatile("global_load_dword %0, %1, off" : "=v"(r.sc) : "v"((const unsigned *)sc + bi));
}
__device__ __forceinline__ void wait_vm2_q4(Q4Pend &r) {
    asm volatile("s_waitcnt vmcnt(2)" : "+v"(r.q), "+v"(r.sc) : : "memory");
}
__device__ __forceinline__ void wait_vm2_q4_bias(Q4Pend &r, f32x4 &b0, f32x4 &b1, f32x4 &b2, f32x4 &b3, float &bv) {
    asm volatile("s_waitcnt vmcnt(2)" : "+v"(r.q), "+v"(r.sc), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(bv) : : "memory");
}
__device__ __forceinline__ void wait_vm4_bias(f32x4 &b0, f32x4 &b1, f32x4 &b2, f32x4 &b3, float &bv) {
    asm volatile("s_waitcnt vmcnt(4)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(bv) : : "memory");
}

}  // namespace

template <int KT, int WT>                        // H = 64 * KT = 32 * n_head; WT = weight type (q4: KT even)
__global__ __launch_bounds__(512, 2) void qkv_attention_kernel(QkvAttArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int H = 64 * KT;
    char *XP = smem;                                          // KT tiles [128 tok][64 halfs], off64 swizzle
    char *WR = smem + KT * 16384;                             // 3 waves x 3 slots x 4 KiB
    char *QS = WR + 3 * 3 * QA_WSLOT;                         // [128][32] halfs, off32 swizzle
    char *KS = QS + QA_TOK * 64;
    half_t *VT = (half_t *)(KS + QA_TOK * 64);                // [32][QA_VT_LD]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.x;
    const int tok0 = a.cu[b], n = a.cu[b + 1] - tok0;
    if (n <= 0) return;
    const int n_head = a.n_head;

    // ---- prologue: the sentence's rows -> LDS (rows >= n repeat row n-1: finite values whose keys are masked
    // and whose queries are never stored)
    {
        const half_t *xb = a.x + (size_t)tok0 * H;
#pragma unroll
        for (int i = 0; i < 2 * KT; ++i) {
            const int piece = wave * 2 * KT + i;              // 1-KiB piece: tile kt, rows p*8 .. p*8+7
            const int kt = piece >> 4, p = piece & 15;
            const int r = p * 8 + (lane >> 3), ch = (lane & 7) ^ ((r >> 1) & 7);
            const int rs = r < n ? r : n - 1;
            __builtin_amdgcn_global_load_lds(AS_GLOBAL(xb + (size_t)rs * H + kt * 64 + ch * 8), AS_LDS(XP + piece * 1024), 16, 0, 0);
        }
    }

    if (wave >= 4 && wave < 7) {
        // =============================== projection wave g: 0 = Q, 1 = K, 2 = V ===============================
        const int g = wave - 4;
        char *ring = WR + g * 3 * QA_WSLOT;
        const half_t *wg = a.w + (size_t)g * H * H;           // this wave's H rows; head h uses rows h*32 .. +32
        unsigned loffW[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 8 + (lane >> 3), ch = (lane & 7) ^ ((r >> 1) & 7);
            loffW[i] = (unsigned)(r * H + ch * 8) * 2u;
        }
        const int ntiles = n_head * KT;
        // f16: tile t goes straight into ring slot `slot` by LDS-DMA.  q4_0 / q4_1: the lane fetches its block of the
        // tile (row lane >> 1, k-block lane & 1; the 64 blocks of a wave's tile are contiguous in the weight planes) into
        // pend[t & 1] and expands it into the slot one tile later (expand), dequantised exactly like the panel kernels.
        Q4Pend pend[2] = {{{0, 0, 0, 0}, 0}, {{0, 0, 0, 0}, 0}};
        auto issue = [&](auto par_tag, int t, int slot) __attribute__((always_inline)) {    // tile t = head (t / KT), k-tile (t % KT)
            t = t < ntiles ? t : ntiles - 1;                  // past the end: re-read the last tile into a dead slot
            const int h = t / KT, kt = t - h * KT;
            if (WT == GW_F16) {
                const half_t *src = wg + (size_t)h * 32 * H + kt * 64;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_global_load_lds(AS_GLOBAL((const char *)src + loffW[i]), AS_LDS(ring + slot * QA_WSLOT + i * 1024), 16, 0, 0);
            } else {
                const int nrow = g * H + h * 32;              // first of the 32 rows, inside one 128-row tile of the planes
                const size_t bi = ((size_t)((nrow >> 7) * KT + kt) * 128 + (nrow & 127)) * 2 + lane;
                q4_fetch_untracked<WT>(pend[decltype(par_tag)::value], a.qs, a.sc, bi);
            }
        };
        auto expand = [&](auto par_tag, int slot) __attribute__((always_inline)) {
            if (WT != GW_F16) {
                const Q4Pend &pr = pend[decltype(par_tag)::value];
                QRegs r;
                r.q.x = pr.q[0]; r.q.y = pr.q[1]; r.q.z = pr.q[2]; r.q.w = pr.q[3]; r.sc = pr.sc;
                char *tile = ring + slot * QA_WSLOT;
                const int row = lane >> 1, blk = lane & 1;
                q4_expand_to_lds<WT>(r, tile, (row << 2) | (blk << 1) | 0);      // elements 0-15 of the block
                q4_expand_to_lds<WT>(r, tile, (row << 2) | (blk << 1) | 1);      // elements 16-31
                asm volatile("" ::: "memory");               // the stores stay above the hand-issued reads of the slot
            }
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        issue(P0{}, 0, 0);
        issue(P1{}, 1, 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // B0: x panel complete
        if (WT != GW_F16) {
            asm volatile("" : "+v"(pend[0].q), "+v"(pend[0].sc), "+v"(pend[1].q), "+v"(pend[1].sc));   // landed (vmcnt(0) above)
            expand(P0{}, 0);
        }

        // per-lane LDS byte addresses of the fragments (ds_read_b128 issued by hand below)
        unsigned aWl[4], aXl[4], aXh[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            aWl[kk] = lds_addr(ring) + off64(l31, kk * 2 + hi);           // + slot * 4 KiB
            aXl[kk] = lds_addr(XP) + off64(l31, kk * 2 + hi);             // + tb * 4 KiB + kt * 16 KiB (kt < 4)
            aXh[kk] = aXl[kk] + 65536;                                    // k-tiles 4, 5
        }
        const float *bias_g = a.bias + g * H;

        // SWAP = the V wave's operand order; one copy of the head loop per order (a per-MFMA branch on g would put
        // every MFMA into its own basic block).
        // The wave has its SIMD's matrix pipe to itself, so it hides its own LDS latency: the fragments of a tile
        // are read in two halves (k-steps 0,1 and 2,3: 2 weight + 8 hidden-state ds_read_b128 each); while the 8
        // MFMAs of one half run, the reads of the next half — of the next tile for the second half — are in
        // flight.  The reads are issued as asm so that the waits can be partial (lgkmcnt(10): everything but
        // the newest half), which the compiler's own wait insertion never does.
        auto project = [&](auto swap_tag) {
        constexpr bool SWAP = decltype(swap_tag)::value;
        f16x8 fw[2][2], fx[2][2][4];                          // [half][k-step in half]([token block])
        auto read_half = [&](auto kt_tag, auto half_tag, unsigned slot_off) {
            constexpr int kt = decltype(kt_tag)::value, half = decltype(half_tag)::value;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kk = half * 2 + i;
                fw[half][i] = lds_read_b128<0>(aWl[kk] + slot_off);
                if constexpr (kt < 4) {
                    fx[half][i][0] = lds_read_b128<kt * 16384>(aXl[kk]);
                    fx[half][i][1] = lds_read_b128<kt * 16384 + 4096>(aXl[kk]);
                    fx[half][i][2] = lds_read_b128<kt * 16384 + 8192>(aXl[kk]);
                    fx[half][i][3] = lds_read_b128<kt * 16384 + 12288>(aXl[kk]);
                } else {
                    fx[half][i][0] = lds_read_b128<(kt - 4) * 16384>(aXh[kk]);
                    fx[half][i][1] = lds_read_b128<(kt - 4) * 16384 + 4096>(aXh[kk]);
                    fx[half][i][2] = lds_read_b128<(kt - 4) * 16384 + 8192>(aXh[kk]);
                    fx[half][i][3] = lds_read_b128<(kt - 4) * 16384 + 12288>(aXh[kk]);
                }
            }
        };
        f32x16 acc[4];
        // all reads older than the newest 10 have landed; names the half's registers so that its MFMAs stay below
        auto wait_half = [&](auto half_tag) {
            constexpr int half = decltype(half_tag)::value;
            wait_lgkm10(fw[half][0], fw[half][1], fx[half][0][0], fx[half][0][1], fx[half][0][2], fx[half][0][3],
                        fx[half][1][0], fx[half][1][1], fx[half][1][2], fx[half][1][3]);
        };
        auto mma_half = [&](auto half_tag) {
            constexpr int half = decltype(half_tag)::value;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) {
                    if (!SWAP) acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[half][i], fx[half][i][tb], acc[tb], 0, 0, 0);
                    else acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fx[half][i][tb], fw[half][i], acc[tb], 0, 0, 0);
                }
        };
        using H0 = std::integral_constant<int, 0>;
        using H1 = std::integral_constant<int, 1>;

        int t = 0, slot = 0;
        [[maybe_unused]] int tl = 0;
        [[maybe_unused]] const bool tl_sel = tid == 256;                       // lane 0 of the Q projection wave
        TL_STAMP_AT(tl_sel, tl++);
        read_half(H0{}, H0{}, 0);                             // tile 0 landed before B0
        for (int h = 0; h <= n_head; ++h) {
            f32x4 bqk[4] = {};
            float bvv = 0.f;
            if (h < n_head) {
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tb][r] = 0.f;
                static_for<KT>([&](auto kt_tag) {
                    constexpr int kt = decltype(kt_tag)::value;
                    const unsigned so = (unsigned)slot * QA_WSLOT;
                    const unsigned so1 = (unsigned)(slot == 2 ? 0 : slot + 1) * QA_WSLOT;
                    read_half(kt_tag, H1{}, so);
                    if (kt == (KT > 1 ? 1 : 0)) {
                        // bias of this head (16 features of a lane for Q / K, one for V), older than the DMA pieces
                        // below so that the counted wait does not have to cover them
                        if (!SWAP) {
#pragma unroll
                            for (int gg = 0; gg < 4; ++gg) gload_untracked(bqk[gg], bias_g + h * 32 + 8 * gg + 4 * hi);
                        } else {
                            gload_untracked(bvv, bias_g + h * 32 + l31);
                        }
                    }
                    using PT = std::integral_constant<int, kt & 1>;          // parity of t (q4: KT is even)
                    using PN = std::integral_constant<int, (kt + 1) & 1>;
                    issue(PT{}, t + 2, slot == 0 ? 2 : slot - 1);
                    wait_half(H0{});
                    mma_half(H0{});
                    if (h < 3) TL_STAMP_AT(tl_sel, tl++);
                    // tile t+1 has landed once at most the pieces of tile t+2 are outstanding (so have the older bias
                    // loads: the last tile's wait hands them to the epilogue)
                    if (WT == GW_F16) {
                        if (kt == KT - 1) wait_vm4_bias(bqk[0], bqk[1], bqk[2], bqk[3], bvv);
                        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    } else {
                        if (kt == KT - 1) wait_vm2_q4_bias(pend[PN::value], bqk[0], bqk[1], bqk[2], bqk[3], bvv);
                        else wait_vm2_q4(pend[PN::value]);
                        expand(PN{}, slot == 2 ? 0 : slot + 1);              // tile t+1 -> its slot (read one tile ago)
                    }
                    read_half(std::integral_constant<int, (kt + 1) % KT>{}, H0{}, so1);
                    wait_half(H1{});
                    mma_half(H1{});
                    slot = slot == 2 ? 0 : slot + 1;
                    ++t;
                });
            }
            // B1: the attention waves are done with the previous head's Q / K / V^T
            if (h < 3) TL_STAMP_AT(tl_sel, tl++);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (h < 3) TL_STAMP_AT(tl_sel, tl++);
            if (h < n_head) {
                if (!SWAP) {
                    // rows = features: lane owns features 8gg + 4hi + 0..3 of token tb*32 + l31
                    char *dst = g == 0 ? QS : KS;
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) {
#pragma unroll
                        for (int tb = 0; tb < 4; ++tb) {
                            f16x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (_Float16)(acc[tb][4 * gg + e] + bqk[gg][e]);
                            *(f16x4 *)(dst + off32(tb * 32 + l31, gg) + hi * 8) = o;
                        }
                    }
                } else {
                    // rows = tokens: lane owns keys tb*32 + 8gg + 4hi + 0..3 of feature l31
#pragma unroll
                    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                        for (int gg = 0; gg < 4; ++gg) {
                            f16x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (_Float16)(acc[tb][4 * gg + e] + bvv);
                            *(f16x4 *)(VT + l31 * QA_VT_LD + tb * 32 + 8 * gg + 4 * hi) = o;
                        }
                }
            }
            // B2: Q / K / V^T of head h are published
            if (h < 3) TL_STAMP_AT(tl_sel, tl++);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (h < 3 || h == n_head) TL_STAMP_AT(tl_sel, tl++);
        }
        };
        if (g < 2) project(std::false_type{}); else project(std::true_type{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the dead prefetches must not outlive the LDS allocation
    } else if (wave < 4) {
        // =============================== attention wave: query block `wave` ===============================
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // B0
        const int qb = wave;
        const bool active = qb * 32 < n;
        const float sc = 1.44269504088896340736f / __builtin_sqrtf(32.0f);   // log2(e) / sqrt(d)
        [[maybe_unused]] int tl = 128;
        [[maybe_unused]] const bool tl_sel = tid == 0;
        for (int h = 0; h <= n_head; ++h) {
            if (h < 4) TL_STAMP_AT(tl_sel, tl++);
            if (h > 0 && active) {
                const int hh = h - 1;
                f16x8 qf[2];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) qf[kk] = *(const f16x8 *)(QS + off32(qb * 32 + l31, kk * 2 + hi));
                // ---- S^T: 4 key tiles x 16 regs; reg r of tile kt <-> key kt*32 + (r&3) + 8*(r>>2) + 4*hi
                f32x16 s[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const f16x8 kf = *(const f16x8 *)(KS + off32(kt * 32 + l31, kk * 2 + hi));
                        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], s[kt], 0, 0, 0);
                    }
                }
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const float v = key < n ? s[kt][r] : -INFINITY;
                        s[kt][r] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32)) * sc;      // = the maximum of the scaled scores (sc > 0), as attention.hip
                float psum = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], sc, -mx));
                        s[kt][r] = pv;
                        psum += pv;
                    }
                psum += __shfl_xor(psum, 32);
                // ---- O^T = V^T P^T
                f32x16 o;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        f16x8 pf;
#pragma unroll
                        for (int e = 0; e < 8; ++e) pf[e] = (_Float16)s[kt][8 * st + e];
                        const int key0 = kt * 32 + 16 * st + 4 * hi;          // keys key0..+3 and key0+8..+11
                        const half_t *vr = VT + l31 * QA_VT_LD + key0;
                        const f16x4 v0 = *(const f16x4 *)vr, v1 = *(const f16x4 *)(vr + 8);
                        f16x8 vf;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
                        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o, 0, 0, 0);
                    }
                // ---- normalise and store: lane (q, hi) owns d = 8g + 4hi + 0..3
                const int q = qb * 32 + l31;
                if (q < n) {
                    const float inv = 1.0f / psum;
                    half_t *op = a.out + (size_t)(tok0 + q) * H + hh * 32;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        f16x4 ov;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[e] = (_Float16)rounded_f32(o[4 * gq + e] * inv);
                        *(f16x4 *)(op + 8 * gq + 4 * hi) = ov;
                    }
                }
            }
            if (h < 4) TL_STAMP_AT(tl_sel, tl++);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // B1
            asm volatile("s_barrier" ::: "memory");                                 // B2
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // B0
        for (int h = 0; h <= n_head; ++h) {
            asm volatile("s_barrier" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
    }
}

bool qkv_attention_supported(const GemmWeight &Wqkv, int n_head, int d_head, int max_len) {
    const int H = n_head * d_head;
    if (!(d_head == 32 && Wqkv.K == H && Wqkv.N == 3 * H && H % 64 == 0 && H <= 384 && max_len <= QA_TOK)) return false;
    return Wqkv.type == GW_F16 || H % 128 == 0;              // q4: an even number of k-tiles per head
}

void launch_qkv_attention(const GemmWeight &Wqkv, const half_t *x, const float *bias, const int32_t *cu_seqlens,
                          int n_sentences, int n_head, half_t *out, hipStream_t stream) {
    QkvAttArgs a;
    a.x = x; a.w = Wqkv.w16; a.qs = Wqkv.qs; a.sc = Wqkv.sc; a.bias = bias; a.cu = cu_seqlens; a.out = out; a.n_head = n_head;
    const int KT = Wqkv.K / 64;
    const size_t lds = (size_t)KT * 16384 + 9 * QA_WSLOT + 2 * QA_TOK * 64 + 32 * QA_VT_LD * 2;
    static DeviceFlags configured[3][7];
    auto go = [&](auto kernel) {
        configure_once(configured[Wqkv.type][KT], [&] { (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        hipLaunchKernelGGL(kernel, dim3(n_sentences), dim3(512), lds, stream, a);
        TL_DUMP(n_sentences >= 256, 136);
    };
    if (Wqkv.type == GW_F16) {
        switch (KT) {
            case 1: go(qkv_attention_kernel<1, GW_F16>); break;
            case 2: go(qkv_attention_kernel<2, GW_F16>); break;
            case 3: go(qkv_attention_kernel<3, GW_F16>); break;
            case 4: go(qkv_attention_kernel<4, GW_F16>); break;
            case 5: go(qkv_attention_kernel<5, GW_F16>); break;
            default: go(qkv_attention_kernel<6, GW_F16>); break;
        }
    } else if (Wqkv.type == GW_Q4_0) {
        switch (KT) {
            case 2: go(qkv_attention_kernel<2, GW_Q4_0>); break;
            case 4: go(qkv_attention_kernel<4, GW_Q4_0>); break;
            default: go(qkv_attention_kernel<6, GW_Q4_0>); break;
        }
    } else {
        switch (KT) {
            case 2: go(qkv_attention_kernel<2, GW_Q4_1>); break;
            case 4: go(qkv_attention_kernel<4, GW_Q4_1>); break;
            default: go(qkv_attention_kernel<6, GW_Q4_1>); break;
        }
    }
}

}  // namespace bert_hip
