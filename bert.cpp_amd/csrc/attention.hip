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
#include "tile_stream.h"      // (the timeline stamps of the tuning builds)

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace bert_hip {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// In-kernel phase clock of the tuning builds (-DBERT_HIP_TIMELINE): wave 0 and wave 4 of a workgroup (the two waves of one SIMD)
// add up, over all their items and chunks, the shader cycles between phase boundaries: staging (an item's start to behind its
// barrier), S^T (issue to RESULTS: the mark names the accumulators, so the matrix pipe has delivered them), softmax (to the
// last P fragment and the rescaled output accumulators), P·V (to the output accumulators), store.  A mark is two empty
// volatile asm statements around s_memtime that name the values the next phase starts from: nothing of the next phase can
// be scheduled in front of it, nothing of the previous one behind it.  launch_att prints the sums of a few workgroups.
#ifdef BERT_HIP_TIMELINE
#define ATT_TL(...) __VA_ARGS__
__device__ __forceinline__ long long att_clock() {
    long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
#else
#define ATT_TL(...)
#endif
#ifndef ATT_S_AHEAD
#define ATT_S_AHEAD 1                 // long-sentence form: K fragments of S^T requested this many MFMAs ahead (0: the compiler's order)
#endif
#ifndef ATT_S_AHEAD_SHORT
#define ATT_S_AHEAD_SHORT 0           // the same for the short form (sentences up to 128 tokens)
#endif
constexpr int ATT_CHUNK = 128;      // keys per online-softmax step (4 S^T tiles of 32)
constexpr int VT_PAD = 4;           // halfs of padding per V^T row: 8-byte skew -> conflict-free ds_read_b64

template <int D>
__device__ __forceinline__ int k_off(int row, int chunk) {
    // 16-byte chunk swizzle of the K tile ([n][D] halfs) for conflict-free ds_read_b128
    if (D == 32) return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// ---- measured and removed in round 5 (commit 37a896e holds the source; numbers in profiles/r5_experiments.txt) ----------------
// A chunk loop SOFTWARE-PIPELINED inside the wave: while the softmax of chunk i runs, the wave issues the S^T MFMAs of chunk
// i + 1 into a second set of score registers and the P·V MFMAs of the key tile whose numerators were finished one step
// earlier — groups of one MFMA + one piece of VALU work in a fixed program order (sched_barrier between groups, results
// pinned by empty volatile asm), every LDS fragment requested three MFMAs ahead, the score sets alternating (no copies).
// Same bits (395 tests).  512 x 512 tokens, d_head 64, per 12 launches: 9.79 ms against 8.68 ms for the straight loop below
// with eight waves, 10.46 with four (one per SIMD).  Why (tools/ubench/valu_cost.hip, profiles/r5_valu_cost.txt): beside MFMAs
// a v_exp_f32 still costs a wave 7.9 cycles (two waves per SIMD) and the loop needs 64 per chunk and lane; the phases of two
// waves that are out of step already overlap; what the straight loop loses is elsewhere (phase clock).
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
// MULTI: sentences of more than one 128-key chunk can occur (the launcher's max_len > 128).  Such a workgroup fills most of its CU's
// LDS and is alone on it, so nothing hides its staging — global loads of 128 KiB, the transposing LDS stores, the barrier: 19 k of
// a workgroup's 55 k cycles at 512 tokens (timeline of round 5, profiles/r5_attention_timeline.txt).  The MULTI form is
// therefore PERSISTENT: a workgroup walks the (sentence, head) items v = blockIdx.x, + gridDim.x, ... and requests the NEXT
// item's K / V rows into registers (64 per thread) before it computes the current one; at the next turn they only have to be
// written to LDS.  n_items: the number of items (the grid of the non-persistent form).
template <int D, int NT, int CH, bool MULTI = (NT > 256)>
__global__ __launch_bounds__(NT) void attention_mfma_kernel(const half_t *__restrict__ qkv,
                                                             const int32_t *__restrict__ cu_seqlens, int n_head,
                                                             half_t *__restrict__ out, int n_items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool PERSIST = MULTI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int H = n_head * D, ld = 3 * H;
    // XCD-contiguous logical order (see gemm.hip xcd_remap): the heads of one sentence share the
    // 128-byte lines of its Q|K|V rows, so they should hit the same XCD's L2.  (Persistent: gridDim.x is a multiple of 8, a
    // workgroup's items stay on its XCD.)
    struct Item { int h, tok0, n; };
    auto item_of = [&](int v) -> Item {
        const int q8 = n_items >> 3, r8 = n_items & 7, xcd = v & 7;
        const int lb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3);
        const int b = lb / n_head, t0 = cu_seqlens[b];
        return Item{lb % n_head, t0, cu_seqlens[b + 1] - t0};
    };
    // timeline (tuning builds): wave 0 stamps 0.., wave 4 — the other wave of its SIMD — 64..: 0 start, 1 staging done, 2 behind the
    // barrier, then per (query block pass, chunk) four stamps: chunk top, S^T issued, softmax done, P·V issued; last: rows stored
    ATT_TL(long long tl_sum[6] = {0, 0, 0, 0, 0, 0}; long long tl_t = att_clock(); const long long tl_t00 = tl_t; int tl_chunks = 0;)
#define ATT_MARK(k) ATT_TL({ const long long tl_now = att_clock(); tl_sum[k] += tl_now - tl_t; tl_t = tl_now; })

    // ---- K (swizzled rows) and V^T (transposed) of a head in LDS, the padding zeroed.  A thread takes 16-byte chunk c of the
    // row PAIR (2 rp, 2 rp + 1): V^T then goes out as 4-byte stores (two keys of one feature), half as many as row by row.
    // EVERY load of a thread is in flight before its first LDS store (one HBM round trip per loop iteration — the rolled
    // form — was most of the kernel's time at 512 tokens).
    constexpr int CPR = D / 8;                         // 16-byte chunks per row
    constexpr int UNR = MULTI ? 4 : 2;                 // row pairs in flight per thread: 16 (8) loads of 16 bytes
    uint4 kv[UNR][2], vv[UNR][2];
    auto request = [&](const Item &it, int base) __attribute__((always_inline)) {
        const int total = ((it.n + ATT_CHUNK - 1) / ATT_CHUNK * ATT_CHUNK / 2) * CPR;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = base + u * NT, rp = idx / CPR, c = idx % CPR;
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                kv[u][w] = uint4{0, 0, 0, 0}; vv[u][w] = uint4{0, 0, 0, 0};
                const int row = 2 * rp + w;
                if (idx < total && row < it.n) {
                    const half_t *src = qkv + (size_t)(it.tok0 + row) * ld + it.h * D + c * 8;
                    kv[u][w] = *(const uint4 *)(src + H);
                    vv[u][w] = *(const uint4 *)(src + 2 * H);
                }
            }
        }
    };
    auto deposit = [&](char *Ks, half_t *Vt, int vt_ld, int total, int base) __attribute__((always_inline)) {
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
    };

    int v = blockIdx.x;
    Item cur = item_of(v);
    if (PERSIST && cur.n > 0) request(cur, tid);
    for (bool first = true;; first = false) {
    const int h = cur.h, tok0 = cur.tok0, n = cur.n;
    Item nxt{0, 0, 0};
    if (n > 0) {
    ATT_MARK(5)                                        // (between items: the walk, the next item's requests)
    const int n_pad = (n + ATT_CHUNK - 1) / ATT_CHUNK * ATT_CHUNK;
    const int vt_ld = n_pad + VT_PAD;                 // halfs per V^T row
    char *Ks = smem;                                   // [n_pad][D] halfs, swizzled
    half_t *Vt = (half_t *)(smem + (size_t)n_pad * D * 2);   // [D][vt_ld]
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
    {
        const int total = (n_pad / 2) * CPR;
        if (!PERSIST) request(cur, tid);
        if (PERSIST && !first) __syncthreads();            // the previous item's fragment reads are done
        deposit(Ks, Vt, vt_ld, total, tid);
        for (int base = tid + NT * UNR; base < total; base += NT * UNR) {      // (heads beyond 32 K elements: not prefetched)
            request(cur, base);
            deposit(Ks, Vt, vt_ld, total, base);
        }
    }
    __syncthreads();
    ATT_MARK(0)
    if constexpr (MULTI) {
        // vmcnt retires in issue order: a wait for Q fragments BEHIND the requests below waits for the whole next item's rows.  Naming
        // the fragments here puts the compiler's wait here — they were requested before the staging and have landed.
#pragma unroll
        for (int kk = 0; kk < D / 16; ++kk) asm volatile("" : "+v"(qf[kk]));
    }
    if (PERSIST && v + (int)gridDim.x < n_items) {
        nxt = item_of(v + (int)gridDim.x);
        if (nxt.n > 0) request(nxt, tid);              // in flight under this item's mat-muls
    }

    const float sc = 1.44269504088896340736f / __builtin_sqrtf((float)D);   // log2(e) / sqrt(d)
    // One pass = one query block of this wave.  The next query block's Q fragments are requested into the SAME registers behind the
    // chunk loop of the pass before (they are dead there) and named at the END of that pass: the round trip runs under the pass's
    // normalise-and-store, and NO load sits inside the chunk loop.  (Until round 6 they were requested behind the last chunk's S^T
    // MFMAs, a chunk earlier: the compiler's wait for them then sits at the top of EVERY chunk, and in the first chunk of an item
    // it also waits for the next item's K / V rows requested just before — vmcnt retires in issue order: 10 k of an item's 62 k
    // cycles in the phase clock, profiles/r6_experiments.txt §4.  Requested a whole pass ahead into registers of their own the
    // 8-wave form spilled 43 registers: 11.4 against 8.4 ms per 12 launches at 512 tokens.)
    for (int qb = wave; qb < n_qblocks; qb += NT / 64) {
        f32x16 o[D / 32];
        if constexpr (MULTI) {                         // (the short form's single chunk starts its P·V MFMAs from the constant 0)
#pragma unroll
            for (int dv = 0; dv < D / 32; ++dv)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dv][r] = 0.f;
        }
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

        constexpr int KT = CH / 32;                    // key tiles per step
        const int n_steps = (n + CH - 1) / CH * CH;      // (whole steps of padding are skipped: their keys are masked out anyway)
        for (int kc = 0; kc < n_steps; kc += CH) {
            // ---- S^T chunk: KT key tiles x 16 regs; reg r of tile kt <-> key kc + kt*32 + (r&3) + 8*(r>>2) + 4*hi
            f32x16 s[KT];
            // (key tile outside, k-step inside: an accumulator's MFMAs back to back.  The other nest — four accumulators in turn, no
            // MFMA behind its predecessor's result — is 1.5x SLOWER in this phase: 3.6 k against 2.35 k cycles per chunk in the phase
            // clock of round 5, attention at 512 tokens 11.5 against 8.4 ms per 12 launches.)
            // (Measured and dropped in round 5: every fragment of this phase and of P·V requested TWO MFMAs ahead through a ring of three
            // register sets, the order pinned by sched_barrier.  In isolation that is the better form — 39 against 66 cycles per MFMA with
            // one wave on the SIMD, 56 against 68 per wave with two, tools/ubench/mfma_chain.hip — in this kernel it is 7 % slower:
            // 9.04 against 8.42-8.47 ms per 12 launches at 512 tokens on one box, level at 128 tokens.)
            constexpr int S_AHEAD = MULTI ? ATT_S_AHEAD : ATT_S_AHEAD_SHORT;
            if constexpr (S_AHEAD > 0) {
                // K fragments requested S_AHEAD MFMAs ahead, the order pinned (one read, one MFMA, alternating): left alone the compiler
                // puts every fragment into ONE register set — read, wait out the LDS round trip, MFMA, sixteen times per chunk
                constexpr int NF = KT * (D / 16);
                f16x8 kfr[NF];
                auto rd = [&](int f) __attribute__((always_inline)) { kfr[f] = *(const __attribute__((address_space(3))) f16x8 *)(kbase[f % (D / 16)] + (f / (D / 16)) * 32 * K_ROW); };
#pragma unroll
                for (int f = 0; f < S_AHEAD; ++f) rd(f);
                __builtin_amdgcn_sched_group_barrier(0x100, S_AHEAD, 0);
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    if (f + S_AHEAD < NF) rd(f + S_AHEAD);
                    const int kt = f / (D / 16), kk = f % (D / 16);
                    // (the first k-step starts from the constant 0: no zeroing moves)
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfr[f], qf[kk], kk == 0 ? (f32x16)0.f : s[kt], 0, 0, 0);
                    if (f + S_AHEAD < NF) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            } else {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int kk = 0; kk < D / 16; ++kk) {
                    const f16x8 kf = *(const __attribute__((address_space(3))) f16x8 *)(kbase[kk] + kt * 32 * K_ROW);
                    // (the first k-step starts from the constant 0: no zeroing moves)
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], kk == 0 ? (f32x16)0.f : s[kt], 0, 0, 0);
                }
            }
            }
#pragma unroll
            for (int kk = 0; kk < D / 16; ++kk) kbase[kk] += CH * K_ROW;
            ATT_TL(asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3])); ATT_MARK(1) asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3])); ++tl_chunks;)
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
            mx = xor32_max(mx);
            // the scale is positive: max(s) * sc is the maximum of the scaled scores, bit for bit
            const float m_new = fmaxf(m_run, mx * sc);      // finite: every step has >= 1 real key
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);        // 0 on the first chunk
            // numerators (softmax_p8, kernels.h; the same function in qkv_attention2.hip: equal bits across the kernels)
            float psum = 0.f;
            if constexpr (!MULTI) {
                // the short-sentence form is ONE chunk: all numerators first (the scores' 64 registers die into 32 of packed P), then
                // P·V into accumulators that start from the constant 0 (= 0 x alpha) instead of registers zeroed before S^T and kept
                // alive through it — 92 / 114 registers (d_head 32 / 64) instead of 110 / 148: four workgroups per CU where the d_head 64 form had three
                // (same arithmetic in the same order: the same bits)
                f16x8 pfr[KT][2];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int st = 0; st < 2; ++st)
                        pfr[kt][st] = softmax_p8(s[kt][8 * st], s[kt][8 * st + 1], s[kt][8 * st + 2], s[kt][8 * st + 3], s[kt][8 * st + 4],
                                                 s[kt][8 * st + 5], s[kt][8 * st + 6], s[kt][8 * st + 7], sc, m_new, psum);
                psum = xor32_sum(psum);
                l_run = l_run * alpha + psum;
                m_run = m_new;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int dv = 0; dv < D / 32; ++dv) {
                            const lds_halfs vr = vbase[dv] + kt * 32 + 16 * st;
                            const f16x4 v0 = *(const __attribute__((address_space(3))) f16x4 *)vr, v1 = *(const __attribute__((address_space(3))) f16x4 *)(vr + 8);
                            f16x8 vf;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
                            o[dv] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pfr[kt][st], kt == 0 && st == 0 ? (f32x16)0.f : o[dv], 0, 0, 0);
                        }
            } else {
            f16x8 pfr[KT][2];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int st = 0; st < 2; ++st)
                    pfr[kt][st] = softmax_p8(s[kt][8 * st], s[kt][8 * st + 1], s[kt][8 * st + 2], s[kt][8 * st + 3], s[kt][8 * st + 4],
                                             s[kt][8 * st + 5], s[kt][8 * st + 6], s[kt][8 * st + 7], sc, m_new, psum);
            psum = xor32_sum(psum);
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dv = 0; dv < D / 32; ++dv)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dv][r] *= alpha;
            ATT_TL(asm volatile("" : "+v"(pfr[0][0]), "+v"(pfr[1][1]), "+v"(pfr[2][0]), "+v"(pfr[3][1]), "+v"(o[0]), "+v"(l_run)); ATT_MARK(2)
                   asm volatile("" : "+v"(pfr[0][0]), "+v"(pfr[1][1]), "+v"(pfr[2][0]), "+v"(pfr[3][1]), "+v"(o[0]), "+v"(l_run));)
            // ---- O^T += V^T * P^T; keys key0..+3 and key0+8..+11 with key0 = kc + kt*32 + 16*st + 4*hi
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const f16x8 pf = pfr[kt][st];
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
            }
#pragma unroll
            for (int dv = 0; dv < D / 32; ++dv) vbase[dv] += CH;
            ATT_TL(asm volatile("" : "+v"(o[0]), "+v"(o[D / 32 - 1])); ATT_MARK(3) asm volatile("" : "+v"(o[0]), "+v"(o[D / 32 - 1]));)
        }
        // ---- normalise and store: lane (q, hi) owns dv = dvt*32 + 8g + 4hi + 0..3
        auto normalise_and_store = [&]() __attribute__((always_inline)) {
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
        };
        if (MULTI && qb + NT / 64 < n_qblocks) {
            // the next pass's fragments: requested in FRONT of this pass's stores, named BEHIND them (the compiler's wait goes where
            // they are named: at the top of the pass it would, in an item's first pass, wait for the next item's rows), request
            // and naming in ONE branch (in two, the compiler sees a path on which the request is never waited for and keeps a wait
            // at the top of every chunk)
            const int qrow = min((qb + NT / 64) * 32 + l31, n - 1);
            const half_t *qp = qkv + (size_t)(tok0 + qrow) * ld + h * D + hi * 8;
#pragma unroll
            for (int kk = 0; kk < D / 16; ++kk) qf[kk] = *(const f16x8 *)(qp + kk * 16);
            normalise_and_store();
#pragma unroll
            for (int kk = 0; kk < D / 16; ++kk) asm volatile("" : "+v"(qf[kk]));
        } else {
            normalise_and_store();
        }
        ATT_MARK(4)
    }
    }      // n > 0
    else if (PERSIST && v + (int)gridDim.x < n_items) {
        // (an empty sentence: nothing to compute, but the walk goes on and the next item's rows must be under way)
        nxt = item_of(v + (int)gridDim.x);
        __syncthreads();                               // (kv / vv hold nothing of value for an empty item; keep the waves together)
        if (nxt.n > 0) request(nxt, tid);
    }
    v += (int)gridDim.x;
    if (!PERSIST || v >= n_items) break;
    cur = nxt;
    }      // items
    ATT_TL(if ((tid & 255) == 0) {
        unsigned long long *tl = g_timeline + (blockIdx.x & 1023) * 256 + (tid >> 8) * 16;
        for (int k = 0; k < 6; ++k) tl[k] = (unsigned long long)tl_sum[k];
        tl[6] = (unsigned long long)(att_clock() - tl_t00); tl[7] = (unsigned long long)tl_chunks;
    })
#undef ATT_MARK
}

template <int D>
static void launch_att(const half_t *qkv, const int32_t *cu, int B, int n_head, int max_len, half_t *out,
                       hipStream_t s) {
    const int n_pad = (max_len + ATT_CHUNK - 1) / ATT_CHUNK * ATT_CHUNK;
    const size_t lds = (size_t)n_pad * D * 2 + (size_t)D * (n_pad + VT_PAD) * 2;
    // per device: the opt-in is a per-device attribute; the devices of a context launch from threads of their own
    static DeviceFlags configured[2];
    const bool wide = n_pad > 128;
    if (lds > 64 * 1024)
        configure_once(configured[wide], [&] {                 // (once, for the largest LDS the kernel can be launched with)
            if (wide) (void)hipFuncSetAttribute((const void *)attention_mfma_kernel<D, 512, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            else (void)hipFuncSetAttribute((const void *)attention_mfma_kernel<D, 256, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
    const int items = B * n_head;
    // the persistent form: one workgroup per CU (a multiple of 8: a workgroup's items stay on its XCD)
    static int n_cu[MAX_HIP_DEVICES] = {};
    const int dev = current_device_slot();
    if (!n_cu[dev]) {
        hipDeviceProp_t prop;
        n_cu[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8 ? prop.multiProcessorCount / 8 * 8 : 256;
    }
    const int pgrid = items >= 8 ? std::min(n_cu[dev], items / 8 * 8) : items;
    if (wide) BERT_LAUNCH((attention_mfma_kernel<D, 512, 128>), dim3(pgrid), dim3(512), lds, s, qkv, cu, n_head, out, items);
    else BERT_LAUNCH((attention_mfma_kernel<D, 256, 128>), dim3(items), dim3(256), lds, s, qkv, cu, n_head, out, items);
#ifdef BERT_HIP_TIMELINE
    {
        static int shots = 0;
        if (wide && shots++ == 20) {
            (void)hipDeviceSynchronize();
            static unsigned long long h[1024 * 256];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_timeline), sizeof(h));
            fprintf(stderr, "attention phase clock, shader cycles summed per wave over its items (workgroup: wave 0 | wave 4): staging S softmax PV store between total chunks\n");
            for (int b : {0, 1, 2, 100, 200, 255})
                for (int w = 0; w < 2; ++w) {
                    const unsigned long long *t = h + b * 256 + w * 16;
                    fprintf(stderr, "attphase wg %3d wave %d: %8llu %8llu %8llu %8llu %8llu %8llu | %9llu  %llu\n", b, 4 * w, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
                }
        }
    }
#endif
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
