// gemm256.hip — the large-tile weight mat-mul of the H > 384 models (bert-base / mpnet dimensions, BASELINE configs 3, 4):
//   C[t][n] = epilogue( sum_k A[t][k] * W[n][k] + bias[n] )        A: f16 activations, W: f16 (q4 matrices are expanded at load)
// Same operation and epilogues as gemm.hip (reference bert.cpp:822-839, :859-865, :878-882, :885-891); what changes is the
// shape of the work, chosen for the two things a power-limited MI355X pays for besides the MFMAs themselves — LDS
// fragment reads and L2 -> LDS tile traffic:
//   * 256 tokens x 256 features per workgroup of 8 waves (two per SIMD: while one wave waits for a tile or a fragment
//     the other one has the matrix pipe), a wave owns 128 features x 64 tokens = 4 x 2 accumulator blocks, so a k-step
//     is 6 fragment reads for 8 MFMAs (gemm.hip: 4 for 4) and a 64-deep reduction tile moves 64 KiB through the LDS-DMA
//     for 256 MFMAs (gemm.hip: 32 KiB for 64);
//   * v_mfma_f32_32x32x16_f16 with the WEIGHT tile as the A operand and the ACTIVATION tile as the B operand: a lane's
//     accumulator column is one token, its registers are 4-feature runs (bias / GELU / residual per lane, f32);
//   * tiles go HBM/L2 -> LDS by global_load_lds_dwordx4, double buffered (2 x 64 KiB), one barrier per reduction tile,
//     the 16-byte chunk swizzle on the SOURCE address (the DMA writes lane-linearly) and again on the fragment reads;
//   * epilogue: the f32 tile goes through LDS in two passes of 128 feature columns (the tile buffers are free by then),
//     so every global access of the output and of the residual is a full 512-byte row segment; bias, GELU and the
//     residual are applied in f32 and the result is rounded once, exactly as gemm.hip does.
// Workgroups are remapped so that each XCD walks a contiguous range of logical tiles (all feature tiles of a token tile
// back to back): the activation tile then comes from HBM once, not once per XCD.
#include "kernels.h"

#include <type_traits>

namespace bert_hip {

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define G2_GLOBAL(p) ((const __attribute__((address_space(1))) void *)(p))
#define G2_LDS(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 64;
constexpr int G2_TILE = 256 * 128;               // 256 rows x 64 halfs
constexpr int G2_STAGE = 2 * G2_TILE;            // activation tile + weight tile

struct Gemm256Args {
    const half_t *A;        // [M_pad][K], M_pad % 256 == 0
    const half_t *w16;      // [N_pad][K], N % 256 == 0
    const float *bias;      // [N]
    const half_t *resid;    // [M_pad][N] or null
    half_t *C;              // [M_pad][N]
    int N, K, n_tiles_n;
};

__device__ __forceinline__ int g2_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// 256 rows x 128 B: 32 pieces of 1 KiB (8 rows each), 4 per wave
__device__ __forceinline__ void g2_dma_tile(const half_t *src, int ld, char *tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = wave * 4 + i;
        const int r = g * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        __builtin_amdgcn_global_load_lds(G2_GLOBAL(src + (size_t)r * ld + c * 8), G2_LDS(tile + g * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ int g2_xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(Gemm256Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lb = g2_xcd_remap(blockIdx.x, gridDim.x);
    const int nt = lb % p.n_tiles_n, mt = lb / p.n_tiles_n;
    const int m0 = mt * G2_BM, n0 = nt * G2_BN;
    const int K = p.K, nk = K / G2_BK;
    const int wf = wave & 1, wq = wave >> 1;         // feature half (128) / token quarter (64) of the tile
    const int l31 = lane & 31, hi = lane >> 5;

    const half_t *Abase = p.A + (size_t)m0 * K;
    const half_t *Wbase = p.w16 + (size_t)n0 * K;

    f32x16 acc[4][2];                                 // [feature block][token block]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    g2_dma_tile(Abase, K, smem, wave, lane);
    g2_dma_tile(Wbase, K, smem + G2_TILE, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        char *cur = smem + (kt & 1) * G2_STAGE;
        char *nxt = smem + ((kt + 1) & 1) * G2_STAGE;
        if (kt + 1 < nk) {                            // the next tile's traffic first: it lands under this tile's MFMAs
            g2_dma_tile(Abase + (kt + 1) * G2_BK, K, nxt, wave, lane);
            g2_dma_tile(Wbase + (kt + 1) * G2_BK, K, nxt + G2_TILE, wave, lane);
        }
        const char *At = cur, *Wt = cur + G2_TILE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int c = kk * 2 + hi;
            f16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *(const f16x8 *)(Wt + g2_off(wf * 128 + i * 32 + l31, c));
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *(const f16x8 *)(At + g2_off(wq * 64 + j * 32 + l31, c));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: two passes of 128 feature columns through LDS as f32 [256 tokens][32 chunks of 4 floats], the chunk
    // index XORed with (token & 31): conflict-free for the accumulator-layout writes and for the row-wise reads
    float *Cs = (float *)smem;
    const int chunk = tid & 31, trow = tid >> 5;
    // the residual rows of a pass are requested BEFORE its accumulators go through LDS (pass 1: before pass 0's rows are
    // finished): sixteen dependent load -> add -> store rounds cost a workgroup 17 us of HBM latency (measured: fixed cost
    // 25.6 us per workgroup against 8.5 us for the plain epilogue), sixteen loads in flight under other work cost nothing
    f16x4 rv[2][16];
    auto load_resid = [&](auto pass_tag) __attribute__((always_inline)) {
        constexpr int pass = decltype(pass_tag)::value;
        if (EPI == EPI_BIAS_RESID) {
#pragma unroll
            for (int s = 0; s < 16; ++s) rv[pass][s] = *(const f16x4 *)(p.resid + ((size_t)m0 + s * 16 + trow) * p.N + n0 + pass * 128 + chunk * 4);
        }
    };
    auto stage = [&](int pass) __attribute__((always_inline)) {
        if (wf == pass) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int tok = wq * 64 + j * 32 + l31;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ch = i * 8 + g * 2 + hi;                    // features 4 * ch .. + 3 of this pass
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                        *(f32x4 *)(Cs + tok * 128 + ((ch ^ (tok & 31)) << 2)) = v;
                    }
                }
        }
    };
    auto finish = [&](auto pass_tag) __attribute__((always_inline)) {
        constexpr int pass = decltype(pass_tag)::value;
        const int f0 = n0 + pass * 128 + chunk * 4;
        const f32x4 bv = *(const f32x4 *)(p.bias + f0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int tok = s * 16 + trow;
            f32x4 v = *(const f32x4 *)(Cs + tok * 128 + ((chunk ^ (tok & 31)) << 2));
            const size_t off = ((size_t)m0 + tok) * p.N + f0;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bv[e];
            if (EPI == EPI_BIAS_RESID) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rv[pass][s][e];
            }
            f16x4 o;
            if (EPI == EPI_BIAS_GELU) {
                // packed f16, as layer_tail.hip evaluates it (the reference reads the GELU from an f16 table)
                const f16x2_t g0 = gelu_pk16(v[0], v[1]), g1 = gelu_pk16(v[2], v[3]);
                o[0] = g0[0]; o[1] = g0[1]; o[2] = g1[0]; o[3] = g1[1];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
            }
            *(f16x4 *)(p.C + off) = o;
        }
    };
    if constexpr (EPI != EPI_BIAS_RESID) {
        // no residual: bias (and GELU) in the accumulator layout, then the whole 256 x 256 tile goes through LDS ONCE as f16
        // ([256 tokens][32 chunks of 8 features], chunk index XORed with token & 31), and every global store is 16 bytes of a
        // full 512-byte row segment.  Same arithmetic (f32 bias add, one rounding), half the LDS traffic and barriers of the
        // two-pass f32 form below.
        char *Ch = smem;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 bq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[g] = *(const f32x4 *)(p.bias + n0 + wf * 128 + i * 32 + 8 * g + 4 * hi);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int tok = wq * 64 + j * 32 + l31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + bq[g][e];
                    f16x4 o;
                    if (EPI == EPI_BIAS_GELU) {
                        const f16x2_t g0 = gelu_pk16(v[0], v[1]), g1 = gelu_pk16(v[2], v[3]);
                        o[0] = g0[0]; o[1] = g0[1]; o[2] = g1[0]; o[3] = g1[1];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
                    }
                    const int c = wf * 16 + i * 4 + g;                        // 16-byte chunk (8 features) of the row
                    *(f16x4 *)(Ch + tok * 512 + ((c ^ (tok & 31)) << 4) + hi * 8) = o;
                }
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int tok = s * 16 + trow;
            const uint4 v = *(const uint4 *)(Ch + tok * 512 + ((chunk ^ (tok & 31)) << 4));
            *(uint4 *)(p.C + ((size_t)m0 + tok) * p.N + n0 + chunk * 8) = v;
        }
        return;
    }
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    load_resid(P0{});
    stage(0);
    __syncthreads();
    load_resid(P1{});
    finish(P0{});
    __syncthreads();
    stage(1);
    __syncthreads();
    finish(P1{});
}

bool gemm256_supported(const GemmWeight &W, int M_pad) {
    return W.type == GW_F16 && W.w16 && W.N % G2_BN == 0 && W.K % G2_BK == 0 && M_pad % G2_BM == 0 && M_pad > 0;
}

void launch_gemm256(const GemmWeight &W, const half_t *A, const float *bias, const half_t *resid, half_t *C, int M_pad,
                    int epilogue, hipStream_t stream) {
    Gemm256Args a;
    a.A = A; a.w16 = W.w16; a.bias = bias; a.resid = resid; a.C = C;
    a.N = W.N; a.K = W.K; a.n_tiles_n = W.N / G2_BN;
    const int grid = a.n_tiles_n * (M_pad / G2_BM);
    const size_t lds = 2 * G2_STAGE;                  // 128 KiB
    static bool configured[3][MAX_HIP_DEVICES] = {};
    auto go = [&](auto kernel, int e) {
        if (first_launch_on_device(configured[e]))
            (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), lds, stream, a);
    };
    switch (epilogue) {
        case EPI_BIAS: go(gemm256_kernel<EPI_BIAS>, 0); break;
        case EPI_BIAS_GELU: go(gemm256_kernel<EPI_BIAS_GELU>, 1); break;
        default: go(gemm256_kernel<EPI_BIAS_RESID>, 2); break;
    }
}

}  // namespace bert_hip
