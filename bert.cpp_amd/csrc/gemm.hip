// gemm.hip — weight mat-muls of the encoder layer for gfx950 (CDNA4):
//   C[t][n] = epilogue( sum_k A[t][k] * W[n][k] + bias[n] )        A: f16 activations, W: f16 / q4_0 / q4_1
//
// Replaces ggml_mul_mat(weight, cur) + ggml_add(ggml_repeat(bias)) [+ ggml_gelu | + residual add]
// at reference bert.cpp:822-839 (Q/K/V, fused here into one [3H,H] weight), :859-865 (attention
// output + residual), :878-882 (FFN up + GELU), :885-891 (FFN down + residual).
//
// Design (one 64-wide wavefront = one 64x64 output sub-tile):
//   * 128 tokens x 128 features per workgroup of 4 waves, reduction tile 64 = two q4 blocks.
//   * v_mfma_f32_32x32x16_f16 with the WEIGHT tile as the A operand and the ACTIVATION tile as the
//     B operand, so a lane's accumulator column is one token and its registers are 4-feature
//     runs: the epilogue packs 4 f16 per 8-byte store and bias / residual loads are 4-wide.
//   * activation tiles (and f16 weight tiles) go HBM -> LDS by global_load_lds_dwordx4 (no VGPR
//     round trip), double buffered, one barrier per reduction tile.  The LDS image is lane-linear
//     per wave instruction, so the bank swizzle is applied to the SOURCE address and again on the
//     fragment read (chunk ^ ((row >> 1) & 7)), which makes every ds_read_b128 conflict-free.
//   * q4_0 / q4_1 weights stay quantised in HBM (16 B of nibbles + one f16 scale [+ f16 min] per
//     32 weights, tile-contiguous so a tile is one coalesced 4 KiB read); each thread dequantises
//     one block per reduction tile in registers (v_perm_b32 builds 1024+q half pairs, packed f16
//     math applies (q-8)*d or q*d+m) and writes four swizzled 16-B chunks into the LDS tile.
#include "kernels.h"

#include <cstdlib>

namespace bert_hip {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define AS_GLOBAL(p) ((const __attribute__((address_space(1))) void *)(p))
#define AS_LDS(p) ((__attribute__((address_space(3))) void *)(p))

struct GemmArgs {
    const half_t *A;        // [M_pad][K]
    const half_t *w16;      // [N_pad][K]
    const uint4 *qs;
    const void *sc;
    const float *bias;      // [N]
    const half_t *resid;    // [M_pad][N] or null
    half_t *C;              // [M_pad][N]
    int N, K, n_tiles_n;
};

constexpr int TILE_BYTES = 128 * 128;        // 128 rows x 64 halfs
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // activation tile + weight tile

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// One wave instruction moves 8 rows x 128 B; a wave moves 32 rows, the workgroup the 128-row tile.
__device__ __forceinline__ void dma_tile(const half_t *src, int ld, char *tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = wave * 4 + i;
        const int r = g * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        __builtin_amdgcn_global_load_lds(AS_GLOBAL(src + (size_t)r * ld + c * 8), AS_LDS(tile + g * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ float gelu_tanh(float x) {
    // 0.5 x (1 + tanh(u)) == x / (1 + exp(-2u)),  u = sqrt(2/pi) x (1 + 0.044715 x^2)
    // exp(-2u) = exp2(x * (c1 + c2 x^2)); one v_exp_f32 + one v_rcp_f32 per element.
    const float c1 = -2.0f * 0.79788456080286535588f * 1.44269504088896340736f;
    const float c2 = c1 * 0.044715f;
    const float t = x * __builtin_fmaf(x * x, c2, c1);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}

// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, each with a private
// L2).  Remap so that every XCD walks a CONTIGUOUS range of logical tiles: all feature tiles of one
// token tile then run back to back on one XCD and the activation tile is fetched from HBM once
// instead of once per XCD.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// 4 bytes each holding a nibble value 0..15  ->  two f16x2 = (1024+n0, 1024+n1), (1024+n2, 1024+n3)
__device__ __forceinline__ void nib4_to_half(unsigned n4, f16x2 &p01, f16x2 &p23) {
    const unsigned a = __builtin_amdgcn_perm(0x64646464u, n4, 0x04010400u);
    const unsigned b = __builtin_amdgcn_perm(0x64646464u, n4, 0x04030402u);
    p01 = __builtin_bit_cast(f16x2, a);
    p23 = __builtin_bit_cast(f16x2, b);
}

template <int WT>
__device__ __forceinline__ void dequant_block_to_lds(const uint4 &q, unsigned scbits, char *tile, int row, int kb) {
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
    f16x2 d2, m2;
    if (WT == GW_Q4_0) {
        const _Float16 d = __builtin_bit_cast(_Float16, (unsigned short)(scbits & 0xffffu));
        d2 = (f16x2){d, d};
        m2 = (f16x2){(_Float16)0, (_Float16)0};
    } else {
        const f16x2 dm = __builtin_bit_cast(f16x2, scbits);
        d2 = (f16x2){dm[0], dm[0]};
        m2 = (f16x2){dm[1], dm[1]};
    }
    const f16x2 off = WT == GW_Q4_0 ? (f16x2){(_Float16)1032.0f, (_Float16)1032.0f}
                                     : (f16x2){(_Float16)1024.0f, (_Float16)1024.0f};
    // chunk j holds block elements 8j..8j+7: j=0,1 low nibbles of bytes 0-7 / 8-15; j=2,3 high nibbles
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f16x2 h[4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned word = w[(j & 1) * 2 + u];
            const unsigned n4 = (j < 2 ? word : (word >> 4)) & 0x0f0f0f0fu;
            nib4_to_half(n4, h[2 * u], h[2 * u + 1]);
        }
        uint4 out;
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f16x2 v = h[e] - off;                       // exact small integer in f16
            if (WT == GW_Q4_0) v = v * d2;              // (q - 8) * d
            else v = v * d2 + m2;                       // q * d + m
            o[e] = __builtin_bit_cast(unsigned, v);
        }
        out.x = o[0]; out.y = o[1]; out.z = o[2]; out.w = o[3];
        *(uint4 *)(tile + lds_off(row, kb * 4 + j)) = out;
    }
}

template <int WT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = lb % p.n_tiles_n, mt = lb / p.n_tiles_n;
    const int m0 = mt * GEMM_BM, n0 = nt * GEMM_BN;
    const int K = p.K, nk = K / GEMM_BK;
    const int wf = wave & 1, wt = wave >> 1;          // feature half / token half of the tile
    const int l31 = lane & 31, hi = lane >> 5;

    const half_t *Abase = p.A + (size_t)m0 * K;
    const half_t *Wbase = p.w16 + (size_t)n0 * K;
    const size_t qbase = (size_t)nt * nk * 256;       // blocks of this feature tile, [kt][row][2]

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 qn = {0, 0, 0, 0};
    unsigned sn = 0;

    // ---- prologue: tile 0 -> stage 0
    dma_tile(Abase, K, smem, wave, lane);
    if (WT == GW_F16) {
        dma_tile(Wbase, K, smem + TILE_BYTES, wave, lane);
    } else {
        qn = p.qs[qbase + tid];
        sn = WT == GW_Q4_0 ? (unsigned)((const unsigned short *)p.sc)[qbase + tid] : ((const unsigned *)p.sc)[qbase + tid];
        dequant_block_to_lds<WT>(qn, sn, smem + TILE_BYTES, tid >> 1, tid & 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        char *cur = smem + (kt & 1) * STAGE_BYTES;
        char *nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
        const bool more = kt + 1 < nk;
        if (more) {   // issue the next tile's HBM traffic before touching the matrix cores
            dma_tile(Abase + (kt + 1) * GEMM_BK, K, nxt, wave, lane);
            if (WT == GW_F16) {
                dma_tile(Wbase + (kt + 1) * GEMM_BK, K, nxt + TILE_BYTES, wave, lane);
            } else {
                const size_t bi = qbase + (size_t)(kt + 1) * 256 + tid;
                qn = p.qs[bi];
                sn = WT == GW_Q4_0 ? (unsigned)((const unsigned short *)p.sc)[bi] : ((const unsigned *)p.sc)[bi];
            }
        }
        const char *At = cur, *Wt = cur + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int c = kk * 2 + hi;
            f16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *(const f16x8 *)(Wt + lds_off(wf * 64 + i * 32 + l31, c));
                b[i] = *(const f16x8 *)(At + lds_off(wt * 64 + i * 32 + l31, c));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (WT != GW_F16 && more) dequant_block_to_lds<WT>(qn, sn, nxt + TILE_BYTES, tid >> 1, tid & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue.  Accumulator layout: lane owns token column l31 and 4-feature register runs.
    // Stage the 128x128 f32 tile in LDS (16-B chunk index XOR (token & 31): conflict-free both ways),
    // then every wave-instruction writes two full 256-B output rows: 32 lanes x 4 features each.
    // (The last reduction tile ended with a barrier, so the staging buffers are free.)
    float *Cs = (float *)smem;                 // [128 tokens][32 chunks of 4 floats]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tok = wt * 64 + j * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int chunk = wf * 16 + i * 8 + g * 2 + hi;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                *(f32x4 *)(Cs + tok * 128 + ((chunk ^ (tok & 31)) << 2)) = v;
            }
        }
    __syncthreads();
    const int chunk = tid & 31;
    const int f0 = n0 + chunk * 4;
    if (f0 < p.N) {
        const f32x4 bv = *(const f32x4 *)(p.bias + f0);
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int tok = s * 8 + (tid >> 5);
            f32x4 v = *(const f32x4 *)(Cs + tok * 128 + ((chunk ^ (tok & 31)) << 2));
            const size_t off = ((size_t)m0 + tok) * p.N + f0;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bv[e];
            if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
            }
            if (EPI == EPI_BIAS_RESID) {
                const f16x4 rv = *(const f16x4 *)(p.resid + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
            }
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
            *(f16x4 *)(p.C + off) = o;
        }
    }
}

template <int WT>
static void launch_wt(const GemmArgs &a, int grid, int epilogue, hipStream_t s) {
    switch (epilogue) {
        case EPI_BIAS: BERT_LAUNCH((gemm_mfma_kernel<WT, EPI_BIAS>), dim3(grid), dim3(256), 0, s, a); break;
        case EPI_BIAS_GELU: BERT_LAUNCH((gemm_mfma_kernel<WT, EPI_BIAS_GELU>), dim3(grid), dim3(256), 0, s, a); break;
        default: BERT_LAUNCH((gemm_mfma_kernel<WT, EPI_BIAS_RESID>), dim3(grid), dim3(256), 0, s, a); break;
    }
}

void launch_gemm_mfma(const GemmWeight &W, const half_t *A, const float *bias, const half_t *resid, half_t *C,
                      int M_pad, int epilogue, hipStream_t stream) {
    GemmArgs a;
    a.A = A; a.w16 = W.w16; a.qs = W.qs; a.sc = W.sc; a.bias = bias; a.resid = resid; a.C = C;
    a.N = W.N; a.K = W.K; a.n_tiles_n = W.N_pad / GEMM_BN;
    const int grid = a.n_tiles_n * (M_pad / GEMM_BM);
    if (W.type == GW_F16) launch_wt<GW_F16>(a, grid, epilogue, stream);
    else if (W.type == GW_Q4_0) launch_wt<GW_Q4_0>(a, grid, epilogue, stream);
    else launch_wt<GW_Q4_1>(a, grid, epilogue, stream);
}

// ------------------------------------------------------------------------------------------------
// generic fallback: one thread per output element, f16 weights [N][K] (dequantised copy)
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ void gemm_naive_kernel(const half_t *A, const half_t *W, const float *bias, const half_t *resid, half_t *C,
                                  int M, int N, int K) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int t = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (n >= N || t >= M) return;
    const half_t *a = A + (size_t)t * K, *w = W + (size_t)n * K;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)a[k] * (float)w[k];
    s += bias[n];
    if (EPI == EPI_BIAS_GELU) s = gelu_tanh(s);
    if (EPI == EPI_BIAS_RESID) s += (float)resid[(size_t)t * N + n];
    C[(size_t)t * N + n] = (_Float16)s;
}

void launch_gemm_naive(const GemmWeight &W, const half_t *A, const float *bias, const half_t *resid, half_t *C,
                       int M, int epilogue, hipStream_t stream) {
    dim3 grid((W.N + 63) / 64, (M + 3) / 4), block(256);
    switch (epilogue) {
        case EPI_BIAS: BERT_LAUNCH((gemm_naive_kernel<EPI_BIAS>), grid, block, 0, stream, A, W.naive16, bias, resid, C, M, W.N, W.K); break;
        case EPI_BIAS_GELU: BERT_LAUNCH((gemm_naive_kernel<EPI_BIAS_GELU>), grid, block, 0, stream, A, W.naive16, bias, resid, C, M, W.N, W.K); break;
        default: BERT_LAUNCH((gemm_naive_kernel<EPI_BIAS_RESID>), grid, block, 0, stream, A, W.naive16, bias, resid, C, M, W.N, W.K); break;
    }
}

}  // namespace bert_hip
