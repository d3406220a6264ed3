// attention_core.h — one (sentence, head) of the attention on v_mfma_f32_32x32x16_f16: the body of attention_mfma_kernel
// (attention.hip, where the design is described) as a device function, shared with the one-launch form of the latency
// route (sentence_kernel.hip).  Every thread of the workgroup calls it (uniform arguments); smem: n_pad * D * 2 +
// D * (n_pad + VT_PAD) * 2 bytes.
#pragma once
#include "kernels.h"

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
template <int D, int NT, int CH>
__device__ __forceinline__ void attention_head(const half_t *__restrict__ qkv, const int32_t *__restrict__ cu_seqlens, int n_head,
                                               half_t *__restrict__ out, char *smem, int b, int h) {
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
            // the scale is positive: max(s) * sc is the maximum of the scaled scores, bit for bit; the exponent below is one
            // fma per score (the same arithmetic as qkv_attention2.hip: equal bits across the kernels)
            const float m_new = fmaxf(m_run, mx * sc);      // finite: every step has >= 1 real key
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);        // 0 on the first chunk
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], sc, -m_new));
                    s[kt][r] = pv;
                    psum += pv;
                }
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
                    f16x8 pf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[e] = (_Float16)s[kt][8 * st + e];
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

}  // namespace bert_hip
