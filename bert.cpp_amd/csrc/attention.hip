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
#include "attention_core.h"

namespace bert_hip {

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
    attention_head<D, NT, CH>(qkv, cu_seqlens, n_head, out, smem, b, h);
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
