// attention.hip — scaled-dot-product attention over packed variable-length sentences (gfx950).
//
// Replaces, per (sentence, head): ggml_mul_mat(K,Q) -> ggml_scale(1/sqrt(d)) -> ggml_soft_max ->
// ggml_cont(transpose(V)) -> ggml_mul_mat(V,KQ) -> permute + ggml_cpy to [H,N]
// (reference bert.cpp:843-856).  No mask and no padding exist in the reference: each sentence
// attends over exactly its own N tokens; here that is enforced by the packed layout (keys beyond
// the sentence never enter the tile; the ragged tail of the last 32-key tile is set to -inf).
//
// One workgroup (4 waves) per (sentence, head).  K[n][d] and V^T[d][n] of the head live in LDS
// for the whole workgroup; each wave owns 32-query blocks.  Both mat-muls are computed
// "swapped" on v_mfma_f32_32x32x16_f16 so that the accumulator COLUMN of a lane is one query:
//     S^T[key][q]  = K * Q^T     (A = K rows from LDS,  B = Q rows from HBM)
//     O^T[dv][q]   = V^T * P^T   (A = V^T rows from LDS, B = P^T straight from the S^T registers)
// so the softmax max / sum over keys are in-lane reductions plus one cross-half shuffle, the
// 1/sum normalisation and the online-softmax rescale are per-lane scalars, and P never leaves
// registers: the MFMA k-slot order is arbitrary as long as A and B agree, so the C-layout key
// order of S^T (4-key runs interleaved between the two half-waves) is used as-is for P^T and
// the V^T fragment is gathered with the same permutation (two 8-byte LDS reads).
// Sequences longer than 128 keys stream over 128-key chunks with an online softmax.
#include "kernels.h"

#include <cstdlib>
#include <type_traits>

namespace bert_hip {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ATT_CHUNK = 128;      // keys per online-softmax step (4 S^T tiles of 32)
constexpr int VT_PAD = 4;           // halfs of padding per V^T row: 8-byte skew -> conflict-free ds_read_b64

template <int D>
__device__ __forceinline__ int k_off(int row, int chunk) {
    // 16-byte chunk swizzle of the K tile ([n][D] halfs) for conflict-free ds_read_b128
    if (D == 32) return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// ---- the chunk loop's issue plan (compile time) --------------------------------------------------------------------
// One wave works through a 128-key chunk as: S^T (KT * D/16 MFMAs), softmax (VALU), O^T += V^T P^T (KT * 2 * D/32 MFMAs).
// Issued in that order — rounds 1-4 — the matrix pipe idles through the softmax and the VALU through both mat-muls of the
// wave, two waves per SIMD overlap only where they happen to be out of step, and every MFMA waits for a fragment requested
// one MFMA earlier (512 x 512 tokens, d_head 64: MFMA busy 27 %, 743 us; profiles/r4_pmc.txt).  Round 5 software-pipelines
// the loop INSIDE the wave: while the softmax of chunk i runs, the wave issues the S^T MFMAs of chunk i + 1 (into a second
// set of score registers) and the P·V MFMAs of the key tile whose numerators were finished one step earlier.  The body is a
// list of GROUPS, each = at most one MFMA + one piece of VALU work, in the program order below (sched_barrier between
// groups: the machine scheduler may not regroup them); an MFMA's LDS fragment is requested PIPE_RING - 1 MFMAs ahead.
//   key tile kt = 0..3:  MFMAs  S(i+1)[kt] k-step 0, P·V(kt-1) step 0, S(i+1)[kt] k-step 1, P·V(kt-1) step 1, ...
//                        VALU   arguments(kt, keys 0-15) | exponentials | row sum + arguments(keys 16-31) | exponentials | row sum
//   tail:                MFMAs  P·V(3) ...               VALU   in-lane maximum of S(i+1), one tile per group
// The arithmetic per score is unchanged (softmax_args4 / _exp4 / _sum4 are softmax_p8's three steps, kernels.h; the row sum runs
// over the pairs in the same order): the same bits as the straight-line form and as qkv_attention2.hip.
struct PipeOp { int type, kt, idx; };          // type 1: S tile kt of the NEXT chunk, k-step idx; 2: P·V of key tile kt, step idx = st * DV + dv
struct PipeGroup { int op, piece_kt, piece; }; // op: index into ops, or -1; piece 0..4 of key tile piece_kt (piece_kt 4: maximum of next tile `piece`), or -1
constexpr int PIPE_RING = 4;
template <int D, bool NEXT>
struct PipePlan {
    static constexpr int KS = D / 16, PVN = 2 * (D / 32);
    PipeOp ops[64];
    PipeGroup g[64];
    int n_ops, n;
    constexpr PipePlan() : ops{}, g{}, n_ops(0), n(0) {
        for (int kt = 0; kt <= 4; ++kt) {
            PipeOp list[16] = {};
            int nl = 0;
            const int ns = (NEXT && kt < 4) ? KS : 0, np = kt >= 1 ? PVN : 0;
            for (int i = 0; i < (ns > np ? ns : np); ++i) {
                if (i < ns) list[nl++] = PipeOp{1, kt, i};
                if (i < np) list[nl++] = PipeOp{2, kt - 1, i};
            }
            const int npiece = kt < 4 ? 5 : (NEXT ? 4 : 0);
            const int ng = nl > npiece ? nl : npiece;
            int placed = 0;
            for (int i = 0; i < ng; ++i) {
                PipeGroup x{-1, kt, i < npiece ? i : -1};
                // the MFMAs of this key tile spread evenly over its groups
                if (placed < nl && i * nl >= placed * ng) {
                    x.op = n_ops;
                    ops[n_ops++] = list[placed++];
                }
                g[n++] = x;
            }
        }
    }
};

// NT threads: 256 (4 waves), or 512 for long sentences (their K / V^T fill most of the CU's LDS, so one workgroup is all
// a CU holds: 8 waves = two per SIMD let one wave's softmax run under the other's MFMAs).  CH = keys per online-softmax
// step.  (Sixteen waves with 64-key steps — four per SIMD within 128 registers — were measured at 512 x 512 tokens: 3 %
// SLOWER than eight with 128-key steps; the counters of that shape: MFMA busy 27 %, VALU issue ~21 %, the rest waits.
// Also measured there and dropped: two query blocks per wave pass sharing the K / V^T fragment reads (64-key steps, the
// registers allow no more: 12 % slower; 128-key steps spill), and every fragment of a step requested ahead of its MFMAs
// (3 % slower at 512 tokens; at 128 tokens the extra registers cost the third wave per SIMD: 15 % slower).  Round 4, same
// shape, replay groups of 20 launches: static priority for waves 4-7 (697 against 698-704 us: nothing), the V fragments of a
// chunk requested in front of its softmax (+0.4 %), the output rescale skipped behind a ballot while no query's maximum grows
// by more than 2^8 (+2.7 %: the branch costs more than the 32 multiplies).)
#ifndef BERT_HIP_ATT_PIPE
#define BERT_HIP_ATT_PIPE 1
#endif
// MULTI: sentences of more than one 128-key chunk can occur (the launcher's max_len > 128)
template <int D, int NT, int CH, bool MULTI = (NT > 256)>
__global__ __launch_bounds__(NT) void attention_mfma_kernel(const half_t *__restrict__ qkv,
                                                             const int32_t *__restrict__ cu_seqlens, int n_head,
                                                             half_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-contiguous logical order (see gemm.hip xcd_remap): the heads of one sentence share the
    // 128-byte lines of its Q|K|V rows, so they should hit the same XCD's L2.
    const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7;
    const int lb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int b = lb / n_head, h = lb % n_head;
    const int tok0 = cu_seqlens[b], n = cu_seqlens[b + 1] - tok0;
    if (n <= 0) return;
    const int H = n_head * D, ld = 3 * H;
    const int n_pad = (n + ATT_CHUNK - 1) / ATT_CHUNK * ATT_CHUNK;
    const int vt_ld = n_pad + VT_PAD;                 // halfs per V^T row
    char *Ks = smem;                                   // [n_pad][D] halfs, swizzled
    half_t *Vt = (half_t *)(smem + (size_t)n_pad * D * 2);   // [D][vt_ld]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;

    const int n_qblocks = (n + 31) / 32;
    // Q fragments of this wave's first query block are requested before the K/V staging loads so that
    // both HBM round trips overlap (B operand: lane (q = l31, hi) holds Q[q][kk*16 + hi*8 .. +8]).
    f16x8 qf[D / 16];
    {
        const int qrow = min(wave * 32 + l31, n - 1);
        const half_t *qp = qkv + (size_t)(tok0 + qrow) * ld + h * D + hi * 8;
#pragma unroll
        for (int kk = 0; kk < D / 16; ++kk) qf[kk] = *(const f16x8 *)(qp + kk * 16);
    }

    // ---- stage K (swizzled rows) and V^T (transposed) of this head; zero the padding.  A thread takes 16-byte chunk c of the
    // row PAIR (2 rp, 2 rp + 1): V^T then goes out as 4-byte stores (two keys of one feature), half as many as row by row.
    // EVERY load of a thread is in flight before its first LDS store (a long sentence's workgroup is alone on its CU: one
    // HBM round trip per loop iteration — the rolled form — was most of the kernel's time at 512 tokens).
    constexpr int CPR = D / 8;                         // 16-byte chunks per row
    {
        const int total = (n_pad / 2) * CPR;
        constexpr int UNR = 4;                         // row pairs in flight per thread: 16 loads of 16 bytes
        for (int base = tid; base < total; base += NT * UNR) {
            uint4 kv[UNR][2], vv[UNR][2];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int idx = base + u * NT, rp = idx / CPR, c = idx % CPR;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    kv[u][w] = uint4{0, 0, 0, 0}; vv[u][w] = uint4{0, 0, 0, 0};
                    const int row = 2 * rp + w;
                    if (idx < total && row < n) {
                        const half_t *src = qkv + (size_t)(tok0 + row) * ld + h * D + c * 8;
                        kv[u][w] = *(const uint4 *)(src + H);
                        vv[u][w] = *(const uint4 *)(src + 2 * H);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int idx = base + u * NT, rp = idx / CPR, c = idx % CPR;
                if (idx < total) {
                    *(uint4 *)(Ks + k_off<D>(2 * rp, c)) = kv[u][0];
                    *(uint4 *)(Ks + k_off<D>(2 * rp + 1, c)) = kv[u][1];
                    const f16x8 a = __builtin_bit_cast(f16x8, vv[u][0]), b = __builtin_bit_cast(f16x8, vv[u][1]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
                        *(f16x2v *)(Vt + (c * 8 + e) * vt_ld + 2 * rp) = f16x2v{a[e], b[e]};
                    }
                }
            }
        }
    }
    __syncthreads();

    const float sc = 1.44269504088896340736f / __builtin_sqrtf((float)D);   // log2(e) / sqrt(d)
    for (int qb = wave; qb < n_qblocks; qb += NT / 64) {
        if (qb != wave) {                              // later blocks (n > 128): fetch their Q fragments now
            const int qrow = min(qb * 32 + l31, n - 1);
            const half_t *qp = qkv + (size_t)(tok0 + qrow) * ld + h * D + hi * 8;
#pragma unroll
            for (int kk = 0; kk < D / 16; ++kk) qf[kk] = *(const f16x8 *)(qp + kk * 16);
        }

        f32x16 o[D / 32];
#pragma unroll
        for (int dv = 0; dv < D / 32; ++dv)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dv][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;

        // fragment addresses of the chunk at kc = 0: the swizzle of a K row depends on the row's low bits, i.e. on l31 only,
        // and a V^T row's keys are consecutive, so inside the chunk loop every read is base + compile-time offset and a
        // chunk step is one addition per base (left to the compiler this was ~100 address instructions per chunk)
        typedef const __attribute__((address_space(3))) char *lds_bytes;       // (typed LDS pointers: generic ones become flat loads)
        typedef const __attribute__((address_space(3))) half_t *lds_halfs;
        lds_bytes kbase[D / 16];
#pragma unroll
        for (int kk = 0; kk < D / 16; ++kk) kbase[kk] = (lds_bytes)Ks + k_off<D>(l31, kk * 2 + hi);
        lds_halfs vbase[D / 32];
#pragma unroll
        for (int dv = 0; dv < D / 32; ++dv) vbase[dv] = (lds_halfs)Vt + (dv * 32 + l31) * vt_ld + 4 * hi;
        constexpr int K_ROW = D * 2;                   // bytes per K row

#if BERT_HIP_ATT_PIPE
        constexpr int KT = CH / 32;                    // key tiles per step
        static_assert(KT == 4, "the issue plan is written for 128-key chunks");
        constexpr int DV = D / 32, KS = D / 16;
        constexpr bool PIPE = MULTI;                   // sentences of more than one chunk
        const int n_steps = (n + CH - 1) / CH * CH;      // (whole steps of padding are skipped: their keys are masked out anyway)
        typedef const __attribute__((address_space(3))) f16x8 *lds_f16x8;
        typedef const __attribute__((address_space(3))) f16x4 *lds_f16x4;

        // ---- prologue: S^T of the first chunk; reg r of tile kt <-> key kc + kt*32 + (r&3) + 8*(r>>2) + 4*hi
        f32x16 s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const f16x8 kf = *(lds_f16x8)(kbase[kk] + kt * 32 * K_ROW);
                // (the first k-step starts from the constant 0: no zeroing moves)
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], kk == 0 ? (f32x16)0.f : s[kt], 0, 0, 0);
            }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) kbase[kk] += CH * K_ROW;        // kbase: the NEXT chunk's K rows
        // the ragged tail of a chunk (only a sentence's last chunk can have one) -> -inf; in-lane maximum of the raw scores
        auto mask_tail = [&](f32x16 (&t)[KT], int kc) __attribute__((always_inline)) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kc + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    t[kt][r] = key < n ? t[kt][r] : -INFINITY;
                }
        };
        auto tile_max = [&](const f32x16 &t, float mx) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(t[r], t[r + 1]), mx);   // v_max3_f32
            return mx;
        };
        // (the volatile asm keeps the rare path a real branch: if-converted, its 64 compares and selects run in every chunk)
        if (CH > n) { asm volatile("; ragged first chunk" ::: "memory"); mask_tail(s, 0); }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) mx = tile_max(s[kt], mx);

        // ---- one chunk: softmax of s, O^T += V^T P^T; NEXT: S^T of the following chunk computed underneath (-> s, mx)
        // (s: the chunk's scores; sn receives the following chunk's — the loop below alternates two register sets, a copy
        // sn -> s at the end of every chunk would be 32 v_mov_b64 behind an MFMA-result wait)
        auto chunk = [&](auto next_tag, int kc, f32x16 (&s)[KT], f32x16 (&sn)[KT]) __attribute__((always_inline)) {
            constexpr bool NEXT = decltype(next_tag)::value;
            static constexpr PipePlan<D, NEXT> plan{};
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            // the scale is positive: max(s) * sc is the maximum of the scaled scores, bit for bit
            const float m_new = fmaxf(m_run, mx * sc);      // finite: every step has >= 1 real key
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);        // 0 on the first chunk
#pragma unroll
            for (int dv = 0; dv < DV; ++dv)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dv][r] *= alpha;
            float psum = 0.f, mxn = -INFINITY;
            sm_arg_t aa[KT][2];
            sm_exp_t pp[KT][2];
            f16x8 pk[KT][2];                               // the B fragments of the P·V steps
            f16x8 fr[PIPE_RING];
            auto fragment = [&](const PipeOp op) __attribute__((always_inline)) -> f16x8 {
                if (op.type == 1) return *(lds_f16x8)(kbase[op.idx] + op.kt * 32 * K_ROW);
                // keys key0..+3 and key0+8..+11 with key0 = kc + kt*32 + 16*st + 4*hi
                const lds_halfs vr = vbase[op.idx % DV] + op.kt * 32 + 16 * (op.idx / DV);
                const f16x4 v0 = *(lds_f16x4)vr, v1 = *(lds_f16x4)(vr + 8);
                return f16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            };
#pragma unroll
            for (int j = 0; j < PIPE_RING - 1; ++j)
                if (j < plan.n_ops) fr[j % PIPE_RING] = fragment(plan.ops[j]);
#pragma unroll
            for (int gi = 0; gi < plan.n; ++gi) {
                const PipeGroup G = plan.g[gi];
                if (G.op >= 0) {
                    const PipeOp op = plan.ops[G.op];
                    if (G.op + PIPE_RING - 1 < plan.n_ops) fr[(G.op + PIPE_RING - 1) % PIPE_RING] = fragment(plan.ops[G.op + PIPE_RING - 1]);
                    const f16x8 af = fr[G.op % PIPE_RING];
                    if (op.type == 1)
                        sn[op.kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, qf[op.idx], op.idx == 0 ? (f32x16)0.f : sn[op.kt], 0, 0, 0);
                    else
                        o[op.idx % DV] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, pk[op.kt][op.idx / DV], o[op.idx % DV], 0, 0, 0);
                }
                if (G.piece >= 0) {
                    const int kt = G.piece_kt, st = G.piece >= 2;
                    // (the empty volatile asm statements pin a piece's results to its group: pure operations carry no order
                    // against sched_barrier before the machine scheduler sees them — left alone the row-sum dot products and
                    // the maxima of the next chunk sink to the end of the block, out from under the MFMAs)
                    if (kt == 4) {
                        mxn = tile_max(sn[G.piece], mxn);
                        asm volatile("" : "+v"(mxn));
                    } else {
                        if (G.piece == 2 || G.piece == 4) {
                            softmax_sum4(pp[kt][G.piece == 4], psum);
                            pk[kt][G.piece == 4] = softmax_pack(pp[kt][G.piece == 4]);
                            asm volatile("" : "+v"(psum), "+v"(pk[kt][G.piece == 4]));
                        }
                        if (G.piece == 0 || G.piece == 2) {
                            aa[kt][st] = softmax_args4(s[kt][8 * st], s[kt][8 * st + 1], s[kt][8 * st + 2], s[kt][8 * st + 3], s[kt][8 * st + 4],
                                                       s[kt][8 * st + 5], s[kt][8 * st + 6], s[kt][8 * st + 7], sc, m_new);
                            asm volatile("" : "+v"(aa[kt][st]));
                        }
                        if (G.piece == 1 || G.piece == 3) {
                            pp[kt][G.piece == 3] = softmax_exp4(aa[kt][G.piece == 3]);
                            asm volatile("" : "+v"(pp[kt][G.piece == 3]));
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            psum += __shfl_xor(psum, 32);
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dv = 0; dv < DV; ++dv) vbase[dv] += CH;
            if constexpr (NEXT) {
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) kbase[kk] += CH * K_ROW;
                mx = mxn;
                if (kc + 2 * CH > n) {                   // the following chunk is the sentence's ragged last one
                    asm volatile("; ragged last chunk" ::: "memory");
                    mask_tail(sn, kc + CH);
                    mx = -INFINITY;
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) mx = tile_max(sn[kt], mx);
                }
            }
        };
        f32x16 s2[KT];
        for (int kc = 0;; kc += 2 * CH) {
            if (!(PIPE && kc + CH < n_steps)) { chunk(std::false_type{}, kc, s, s2); break; }
            chunk(std::true_type{}, kc, s, s2);
            if (!(kc + 2 * CH < n_steps)) { chunk(std::false_type{}, kc + CH, s2, s); break; }
            chunk(std::true_type{}, kc + CH, s2, s);
        }
#else      // the straight-line chunk loop of rounds 1-4 (tuning builds: -DBERT_HIP_ATT_PIPE=0)
        constexpr int KT = CH / 32;                    // key tiles per step
        const int n_steps = (n + CH - 1) / CH * CH;      // (whole steps of padding are skipped: their keys are masked out anyway)
        for (int kc = 0; kc < n_steps; kc += CH) {
            // ---- S^T chunk: KT key tiles x 16 regs; reg r of tile kt <-> key kc + kt*32 + (r&3) + 8*(r>>2) + 4*hi
            f32x16 s[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int kk = 0; kk < D / 16; ++kk) {
                    const f16x8 kf = *(const __attribute__((address_space(3))) f16x8 *)(kbase[kk] + kt * 32 * K_ROW);
                    // (the first k-step starts from the constant 0: no zeroing moves)
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], kk == 0 ? (f32x16)0.f : s[kt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int kk = 0; kk < D / 16; ++kk) kbase[kk] += CH * K_ROW;
            // ---- mask the ragged tail (only the sentence's last chunk can have one), chunk max
            if (kc + CH > n) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kc + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        s[kt][r] = key < n ? s[kt][r] : -INFINITY;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(s[kt][r], s[kt][r + 1]), mx);   // v_max3_f32
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            // the scale is positive: max(s) * sc is the maximum of the scaled scores, bit for bit
            const float m_new = fmaxf(m_run, mx * sc);      // finite: every step has >= 1 real key
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);        // 0 on the first chunk
            // numerators (softmax_p8, kernels.h; the same function in qkv_attention2.hip: equal bits across the kernels)
            float psum = 0.f;
            f16x8 pfr[KT][2];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int st = 0; st < 2; ++st)
                    pfr[kt][st] = softmax_p8(s[kt][8 * st], s[kt][8 * st + 1], s[kt][8 * st + 2], s[kt][8 * st + 3], s[kt][8 * st + 4],
                                             s[kt][8 * st + 5], s[kt][8 * st + 6], s[kt][8 * st + 7], sc, m_new, psum);
            psum += __shfl_xor(psum, 32);
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dv = 0; dv < D / 32; ++dv)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dv][r] *= alpha;
            // ---- O^T += V^T * P^T
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const f16x8 pf = pfr[kt][st];
                    // keys key0..+3 and key0+8..+11 with key0 = kc + kt*32 + 16*st + 4*hi
#pragma unroll
                    for (int dv = 0; dv < D / 32; ++dv) {
                        const lds_halfs vr = vbase[dv] + kt * 32 + 16 * st;
                        const f16x4 v0 = *(const __attribute__((address_space(3))) f16x4 *)vr, v1 = *(const __attribute__((address_space(3))) f16x4 *)(vr + 8);
                        f16x8 vf;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
                        o[dv] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[dv], 0, 0, 0);
                    }
                }
#pragma unroll
            for (int dv = 0; dv < D / 32; ++dv) vbase[dv] += CH;
        }
#endif
        // ---- normalise and store: lane (q, hi) owns dv = dvt*32 + 8g + 4hi + 0..3
        const int q = qb * 32 + l31;
        if (q < n) {
            const float inv = 1.0f / l_run;
            half_t *op = out + (size_t)(tok0 + q) * H + h * D;
#pragma unroll
            for (int dv = 0; dv < D / 32; ++dv)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (_Float16)rounded_f32(o[dv][4 * g + e] * inv);
                    *(f16x4 *)(op + dv * 32 + 8 * g + 4 * hi) = ov;
                }
        }
    }
}

template <int D>
static void launch_att(const half_t *qkv, const int32_t *cu, int B, int n_head, int max_len, half_t *out,
                       hipStream_t s) {
    const int n_pad = (max_len + ATT_CHUNK - 1) / ATT_CHUNK * ATT_CHUNK;
    const size_t lds = (size_t)n_pad * D * 2 + (size_t)D * (n_pad + VT_PAD) * 2;
    // per device: the opt-in is a per-device attribute; the devices of a context launch from threads of their own
    static DeviceFlags configured[3];
    const bool wide = n_pad > 128;
    // (tuning: BERT_HIP_ATT_WAVES=4 runs long sentences with one wave per SIMD instead of two)
    static const bool four = [] { const char *e = getenv("BERT_HIP_ATT_WAVES"); return e && atoi(e) == 4; }();
    if (lds > 64 * 1024)
        configure_once(configured[wide ? (four ? 2 : 1) : 0], [&] {                 // (once, for the largest LDS the kernel can be launched with)
            if (wide && four) (void)hipFuncSetAttribute((const void *)attention_mfma_kernel<D, 256, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            else if (wide) (void)hipFuncSetAttribute((const void *)attention_mfma_kernel<D, 512, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            else (void)hipFuncSetAttribute((const void *)attention_mfma_kernel<D, 256, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
    if (wide && four) BERT_LAUNCH((attention_mfma_kernel<D, 256, 128, true>), dim3(B * n_head), dim3(256), lds, s, qkv, cu, n_head, out);
    else if (wide) BERT_LAUNCH((attention_mfma_kernel<D, 512, 128>), dim3(B * n_head), dim3(512), lds, s, qkv, cu, n_head, out);
    else BERT_LAUNCH((attention_mfma_kernel<D, 256, 128>), dim3(B * n_head), dim3(256), lds, s, qkv, cu, n_head, out);
}

bool launch_attention_mfma(const half_t *qkv, const int32_t *cu_seqlens, int n_sentences, int n_head, int d_head,
                           int max_len, half_t *out, hipStream_t stream) {
    const int n_pad = (max_len + ATT_CHUNK - 1) / ATT_CHUNK * ATT_CHUNK;
    const size_t lds = (size_t)n_pad * d_head * 2 + (size_t)d_head * (n_pad + VT_PAD) * 2;
    if (lds > 160 * 1024) return false;
    if (d_head == 32) launch_att<32>(qkv, cu_seqlens, n_sentences, n_head, max_len, out, stream);
    else if (d_head == 64) launch_att<64>(qkv, cu_seqlens, n_sentences, n_head, max_len, out, stream);
    else return false;
    return true;
}

// ------------------------------------------------------------------------------------------------
// generic fallback: one wave per (sentence, head, query); any d_head, any length
// ------------------------------------------------------------------------------------------------
__global__ void attention_naive_kernel(const half_t *qkv, const int32_t *cu_seqlens, int n_head, int d, half_t *out) {
    extern __shared__ float sh[];          // [4 waves][max_len] scores
    const int b = blockIdx.y, h = blockIdx.z;
    const int tok0 = cu_seqlens[b], n = cu_seqlens[b + 1] - tok0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + wave;
    if (q >= n) return;
    const int H = n_head * d, ld = 3 * H;
    float *s = sh + (size_t)wave * gridDim.x * 4;   // max_len rounded up to 4 per wave
    const half_t *qp = qkv + (size_t)(tok0 + q) * ld + h * d;
    const float scale = 1.0f / sqrtf((float)d);
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) {
        const half_t *kp = qkv + (size_t)(tok0 + j) * ld + H + h * d;
        float a = 0.f;
        for (int e = 0; e < d; ++e) a += (float)kp[e] * (float)qp[e];
        a *= scale;
        s[j] = a;
        mx = fmaxf(mx, a);
    }
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) { const float p = __expf(s[j] - mx); s[j] = p; sum += p; }
    for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < d; e += 64) {
        float a = 0.f;
        for (int j = 0; j < n; ++j) a += (float)qkv[(size_t)(tok0 + j) * ld + 2 * H + h * d + e] * s[j];
        out[(size_t)(tok0 + q) * H + h * d + e] = (_Float16)(a * inv);
    }
}

void launch_attention_naive(const half_t *qkv, const int32_t *cu_seqlens, int n_sentences, int n_head, int d_head,
                            int max_len, half_t *out, hipStream_t stream) {
    const int qblocks = (max_len + 3) / 4;
    dim3 grid(qblocks, n_sentences, n_head);
    const size_t lds = (size_t)4 * qblocks * 4 * sizeof(float);
    BERT_LAUNCH(attention_naive_kernel, grid, dim3(256), lds, stream, qkv, cu_seqlens, n_head, d_head, out);
}

}  // namespace bert_hip
