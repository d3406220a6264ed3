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
    if (wide) BERT_LAUNCH((attention_mfma_kernel<D, 512, 128>), dim3(B * n_head), dim3(512), lds, s, qkv, cu, n_head, out);
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
