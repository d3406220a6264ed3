// gemm2x.hip — the weight mat-mul of the H > 384 models with TWO independent workgroups per CU (round 4):
//   C[t][n] = epilogue( sum_k A[t][k] * W[n][k] + bias[n] (+ resid[t][n]) )        A: f16 activations, W: f16 image
// Same operation, operand layouts, accumulator layout, epilogue arithmetic and bits as gemm256.hip (reference bert.cpp:822-839,
// :859-865, :878-882, :885-891).  What changes is who shares what:
//   gemm256: ONE workgroup of 8 waves per CU, 256 x 256 tile, one barrier domain.  The two waves of a SIMD are always in the
//   same place — both in the main loop, both at the barrier, both in the epilogue: 5.7 (bias) / 10.6 (GELU) / 14-25 us
//   (residual) per output tile with the matrix pipe idle, beside a 17.4 us main loop at K = 768.
//   gemm2x: TWO workgroups of 4 waves per CU (one wave per SIMD each), 256 tokens x 128 features per workgroup, each with its
//   own stream of reduction tiles, its own barriers and its own output tiles.  The partner of a wave on its SIMD belongs to the
//   OTHER workgroup: while one stores a finished tile, waits at a barrier or for a tile to land, the other has the matrix pipe.
//   The second workgroup of a CU starts half an output tile late (`skew`), so the epilogues of the two fall into each other's
//   main loops.
// The price: a 256 x 128 tile moves (256 + 128) rows per 128 x 256 x k MACs through the LDS-DMA, 1.5 x gemm256's bytes per
// MFMA; and two workgroups must share the CU's 160 KiB: reduction tiles are 32 deep (64-byte rows), THREE stages of 24 KiB
// (requests run two stages = 1024 MFMA-cycles ahead, as in gemm256; the wait at a stage's barrier is vmcnt(6): the six pieces
// of the stage after it stay in flight) + 2 KiB of staging per wave = 80 KiB per workgroup.
//   * 64-byte rows: the 16-byte chunk c of row r sits at position c ^ ((r >> 2) & 3) (swizzle on the DMA's source address and
//     again on the fragment reads: conflict-free for ds_read_b128's lane groups);
//   * a wave owns 64 tokens x 128 features = 4 x 2 accumulator blocks (as in gemm256: 6 fragment reads per 8 MFMAs), the four
//     waves are the four token quarters of the tile;
//   * the stage index is a compile-time constant (the stream is unrolled by three, K / 32 must be a multiple of 3): fragment
//     addresses never change, the stage is an immediate offset of the ds_read;
//   * persistent: workgroup j of an XCD walks output tiles j, j + 64, ... of the XCD's range (feature tiles of a token tile back
//     to back), the stream of reduction tiles runs across output tiles, a finished tile's epilogue runs behind the next tile's
//     first barrier; accumulators start from bias (+ residual), requested inside the previous epilogue (gemm256.hip).
#include "tile_stream.h"

#include <type_traits>

namespace bert_hip {

namespace {

constexpr int X_BM = 256, X_BN = 128, X_BK = 32;
constexpr int X_A_TILE = X_BM * X_BK * 2;          // 16 KiB: 256 token rows x 64 bytes
constexpr int X_W_TILE = X_BN * X_BK * 2;          //  8 KiB: 128 feature rows x 64 bytes
constexpr int X_STAGE = X_A_TILE + X_W_TILE;       // 24 KiB
constexpr int X_NSTAGE = 3;
constexpr int X_STG = 2048;                        // staging per wave: [16 tokens][64 features] f16
constexpr int X_LDS = X_NSTAGE * X_STAGE + 4 * X_STG;   // 80 KiB: two workgroups fill a CU's LDS

struct Gemm2xArgs {
    const half_t *A;        // [M_pad][K], M_pad % 256 == 0
    const half_t *w16;      // [N_pad][K], N % 128 == 0
    const float *bias;      // [N]
    const half_t *resid;    // [M_pad][N] or null
    half_t *C;              // [M_pad][N]
    int N, K, n_tiles_n, n_tiles;
    int n_groups;           // feature-tile groups (gemm256.hip: XCD pairs split the feature tiles of a wide matrix)
    int skew;               // s_sleep units (64 cycles) the second workgroup of every CU waits before it starts
};

template <int OFF>
__device__ __forceinline__ f16x8 x_read_b128(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
struct XFrag {
    f16x8 a[4], b[2];                                  // weight rows (4 x 32 features), activation rows (2 x 32 tokens) of one k-step
};
__device__ __forceinline__ void x_wait6(XFrag &f) {
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]) : : "memory");
}
// stage barrier: this wave's pieces of the stage have landed (everything but the six pieces of the stage after it), its reads
// of the previous stage have returned (the fragments of that stage's last k-step are named: their MFMAs run behind the barrier)
__device__ __forceinline__ void x_stage_barrier(XFrag &f) {
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier"
                 : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]) : : "memory");
}
__device__ __forceinline__ void x_drain_barrier(XFrag &f) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier"
                 : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]) : : "memory");
}

}  // namespace

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm2x_kernel(Gemm2xArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);        // the wave = token quarter (64) of the tile
    const int K = p.K, nk = K / X_BK;

    // ---- this workgroup's output tiles (gemm256.hip's walk with 128-feature tiles and 2 x 32 workgroups per XCD)
    const int xcd = blockIdx.x & 7, S = gridDim.x >> 3, j_in_xcd = (int)(blockIdx.x >> 3);
    const int G = p.n_groups, cnt_n = p.n_tiles_n / G, n_begin = (xcd % G) * cnt_n;
    int t_begin, t_end;
    if (G == 1) {
        const int q8 = p.n_tiles >> 3, r8 = p.n_tiles & 7;
        t_begin = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        t_end = t_begin + q8 + (xcd < r8 ? 1 : 0);
    } else {
        const int R = 8 / G, r = xcd / G, tm = p.n_tiles / p.n_tiles_n, q = tm / R, rem = tm % R;
        const int m_begin = r < rem ? r * (q + 1) : rem * (q + 1) + (r - rem) * q;
        t_begin = m_begin * cnt_n;
        t_end = t_begin + (q + (r < rem ? 1 : 0)) * cnt_n;
    }
    int tile = t_begin + j_in_xcd;
    if (tile >= t_end) return;
    // the workgroups of the second half of an XCD's list are (in dispatch order) the second ones on their CUs: half a tile late
    if (2 * j_in_xcd >= S)
        for (int i = 0; i < p.skew; i += 64) __builtin_amdgcn_s_sleep(64);

    // ---- LDS-DMA: a stage is 16 activation pieces + 8 weight pieces of 1 KiB (16 rows of 64 bytes); a wave issues activation
    // pieces 4 wq .. 4 wq + 3 and weight pieces 2 wq, 2 wq + 1.  Lane l of a piece: row l >> 2, position l & 3 <- source chunk
    // (l & 3) ^ ((row >> 2) & 3) = (l & 3) ^ ((l >> 4) & 3) (pieces start at multiples of 16 rows).  Source offsets in elements:
    const unsigned dsrc = (unsigned)((lane >> 2) * K + ((((lane & 3) ^ ((lane >> 4) & 3))) << 3));
    auto dma_piece = [&](const half_t *a_src, const half_t *w_src, char *stage_base, auto i_tag) __attribute__((always_inline)) {
        constexpr int i = decltype(i_tag)::value;       // 0..3: activation pieces, 4, 5: weight pieces
        if constexpr (i < 4)
            __builtin_amdgcn_global_load_lds(AS_GLOBAL(a_src + (size_t)((wq * 4 + i) * 16) * K + dsrc), AS_LDS(stage_base + (wq * 4 + i) * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds(AS_GLOBAL(w_src + (size_t)((wq * 2 + (i - 4)) * 16) * K + dsrc), AS_LDS(stage_base + X_A_TILE + (wq * 2 + (i - 4)) * 1024), 16, 0, 0);
    };

    // ---- fragment addresses of the two k-steps of a stage (stage 0; stage s is the immediate offset s * X_STAGE)
    const int l31 = lane & 31, hi = lane >> 5;
    unsigned aW[2], aA[2];
    {
        const int s = (l31 >> 2) & 3;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned swz = (unsigned)(((kk * 2 + hi) ^ s) << 4);
            aA[kk] = (unsigned)(size_t)smem + (unsigned)((wq * 64 + l31) * 64) + swz;      // (LDS addresses are 32-bit)
            aW[kk] = (unsigned)(size_t)smem + (unsigned)(X_A_TILE + l31 * 64) + swz;
        }
    }
    auto read_frag = [&](XFrag &f, auto st_tag, auto kk_tag) __attribute__((always_inline)) {
        constexpr int kk = decltype(kk_tag)::value, o = decltype(st_tag)::value * X_STAGE;
        f.a[0] = x_read_b128<o>(aW[kk]); f.a[1] = x_read_b128<o + 2048>(aW[kk]);
        f.a[2] = x_read_b128<o + 4096>(aW[kk]); f.a[3] = x_read_b128<o + 6144>(aW[kk]);
        f.b[0] = x_read_b128<o>(aA[kk]); f.b[1] = x_read_b128<o + 2048>(aA[kk]);
    };

    f32x16 acc[4][2];                                 // [feature block][token block]
    auto mfma_step_with = [&](const XFrag &f, auto fill) __attribute__((always_inline)) {
        static_for<8>([&](auto m_tag) __attribute__((always_inline)) {
            constexpr int m = decltype(m_tag)::value, i = m >> 1, j = m & 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
            fill(m_tag);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto mfma_step = [&](const XFrag &f) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
    };

    // ---- accumulators start from bias (+ residual), requested between the phases of the previous epilogue (gemm256.hip)
    f16x4 rv[EPI == EPI_BIAS_RESID ? 4 : 1][2][4];
    auto init_loads = [&](int im0, int in0) __attribute__((always_inline)) {
        int l31 = lane & 31, hi = lane >> 5;
        asm volatile("" : "+v"(l31), "+v"(hi));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *(const f32x4 *)(p.bias + in0 + i * 32 + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][0][4 * g + e] = b[e];
            }
        if (EPI == EPI_BIAS_RESID) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const half_t *rrow = p.resid + ((size_t)im0 + wq * 64 + j * 32 + l31) * p.N + in0 + 4 * hi;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) rv[i][j][g] = *(const f16x4 *)(rrow + i * 32 + 8 * g);
            }
        }
    };
    auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float b = acc[i][0][4 * g + e];
                    if (EPI == EPI_BIAS_RESID) {
                        acc[i][1][4 * g + e] = b + (float)rv[i][1][g][e];
                        acc[i][0][4 * g + e] = b + (float)rv[i][0][g][e];
                    } else {
                        acc[i][1][4 * g + e] = b;
                    }
                }
    };

    // ---- epilogue of the output tile at (em0, en0): gemm256.hip's two phases; the staging area is 2 KiB per wave, so a round
    // (ip, j) = [32 tokens][64 features] goes through it in two halves of 16 tokens (the lanes of the other half sit out the
    // writes): every global store is still 16 bytes of a full 128-byte row segment, sixteen store instructions per tile and wave.
    char *const stg = smem + X_NSTAGE * X_STAGE + wq * X_STG;
    auto epilogue = [&](int em0, int en0, bool next, int nm0, int nn0) __attribute__((always_inline)) {
        int l31 = lane & 31, hi = lane >> 5, lane_e = lane;
        asm volatile("" : "+v"(l31), "+v"(hi), "+v"(lane_e));
        typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
        u32x16 o16[4];
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x16 &a = acc[2 * ip + ii][j];
                        f16x4 h;
                        if (EPI == EPI_BIAS_GELU) {
                            const f16x2_t g0 = gelu_pk16(a[4 * g], a[4 * g + 1]), g1 = gelu_pk16(a[4 * g + 2], a[4 * g + 3]);
                            h[0] = g0[0]; h[1] = g0[1]; h[2] = g1[0]; h[3] = g1[1];
                            __builtin_amdgcn_sched_barrier(0);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) h[e] = (_Float16)a[4 * g + e];
                        }
                        const uint2 hb = __builtin_bit_cast(uint2, h);
                        o16[ip * 2 + j][(ii * 4 + g) * 2] = hb.x;
                        o16[ip * 2 + j][(ii * 4 + g) * 2 + 1] = hb.y;
                    }
        __builtin_amdgcn_sched_barrier(0);
        {
            int im0 = next ? nm0 : em0, in0 = next ? nn0 : en0;
            asm volatile("" : "+s"(im0), "+s"(in0) : "v"(o16[0][15]), "v"(o16[1][15]), "v"(o16[2][15]), "v"(o16[3][15]), "v"(o16[0][0]), "v"(o16[1][0]), "v"(o16[2][0]), "v"(o16[3][0]) : "memory");
            init_loads(im0, in0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if ((l31 >> 4) == half) {
                        const int r16 = l31 & 15;
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const uint2 hb = {o16[ip * 2 + j][(ii * 4 + g) * 2], o16[ip * 2 + j][(ii * 4 + g) * 2 + 1]};
                                *(uint2 *)(stg + r16 * 128 + (((ii * 4 + g) ^ (r16 & 7)) << 4) + hi * 8) = hb;
                            }
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    half_t *crow = p.C + ((size_t)em0 + wq * 64 + j * 32 + half * 16) * p.N + en0 + ip * 64;
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int row = it * 8 + (lane_e >> 3), ch = lane_e & 7;
                        const uint4 v = *(const uint4 *)(stg + row * 128 + ((ch ^ (row & 7)) << 4));
                        *(uint4 *)(crow + (size_t)row * p.N + ch * 8) = v;
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
    };

    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    int m0 = (tile / cnt_n) * X_BM, n0 = (n_begin + tile % cnt_n) * X_BN;
    {   // stages 0 and 1 of the first output tile
        const half_t *a = p.A + (size_t)m0 * K, *w = p.w16 + (size_t)n0 * K;
        static_for<6>([&](auto i) __attribute__((always_inline)) { dma_piece(a, w, smem, i); });
        static_for<6>([&](auto i) __attribute__((always_inline)) { dma_piece(a + X_BK, w + X_BK, smem + X_STAGE, i); });
    }
    init_loads(m0, n0);                                // (the first tile's initial values: one exposed round trip per launch)
    init_acc();
    XFrag f0, f1;
    bool have_prev = false;
    int pm0 = 0, pn0 = 0;

    for (;;) {
        const int next = tile + S;
        const bool more = next < t_end;
        // (after the last output tile the stream requests this tile's first stages once more: cheaper than a branch around every request)
        const int nm0 = more ? (next / cnt_n) * X_BM : m0, nn0 = more ? (n_begin + next % cnt_n) * X_BN : n0;
        const half_t *ta = p.A + (size_t)m0 * K, *tw = p.w16 + (size_t)n0 * K;
        const half_t *na = p.A + (size_t)nm0 * K, *nw = p.w16 + (size_t)nn0 * K;
        // one stage (32 of k): behind its barrier the deferred last k-step of the stage before (or, FIRST: of the previous output
        // tile, then that tile's epilogue), then its own first k-step; the six pieces of the stage two ahead go out behind MFMAs
        // 1, 4, 7 of both k-steps (into the buffer of the stage before, whose reads have returned)
        auto stage = [&](auto st_tag, auto first_tag, int kt) __attribute__((always_inline)) {
            constexpr int st = decltype(st_tag)::value;
            constexpr bool first = decltype(first_tag)::value;
            x_stage_barrier(f1);
            const int q = kt + 2;                      // the stage requested now
            const half_t *ra = q < nk ? ta + q * X_BK : na + (q - nk) * X_BK;
            const half_t *rw = q < nk ? tw + q * X_BK : nw + (q - nk) * X_BK;
            char *rstage = smem + ((st + 2) % X_NSTAGE) * X_STAGE;
            auto fill = [&](auto base_tag, auto m_tag) __attribute__((always_inline)) {
                constexpr int m = decltype(m_tag)::value, base = decltype(base_tag)::value;
                if constexpr (m == 1) dma_piece(ra, rw, rstage, std::integral_constant<int, base>{});
                if constexpr (m == 4) dma_piece(ra, rw, rstage, std::integral_constant<int, base + 1>{});
                if constexpr (m == 7) dma_piece(ra, rw, rstage, std::integral_constant<int, base + 2>{});
            };
            if constexpr (first) {
                // (the first three pieces in one go: behind them come the previous tile's last k-step and its epilogue)
                dma_piece(ra, rw, rstage, I0{}); dma_piece(ra, rw, rstage, I1{}); dma_piece(ra, rw, rstage, I2{});
                if (have_prev) {
                    mfma_step(f1);
                    epilogue(pm0, pn0, true, m0, n0);
                    init_acc();
                }
                read_frag(f0, st_tag, I0{});
            } else {
                read_frag(f0, st_tag, I0{});
                mfma_step_with(f1, [&](auto m) __attribute__((always_inline)) { fill(I0{}, m); });
            }
            read_frag(f1, st_tag, I1{}); x_wait6(f0);
            mfma_step_with(f0, [&](auto m) __attribute__((always_inline)) { fill(I3{}, m); });
        };
        stage(I0{}, std::true_type{}, 0);
        stage(I1{}, std::false_type{}, 1);
        stage(I2{}, std::false_type{}, 2);
        for (int kt = 3; kt < nk; kt += 3) {
            stage(I0{}, std::false_type{}, kt);
            stage(I1{}, std::false_type{}, kt + 1);
            stage(I2{}, std::false_type{}, kt + 2);
        }
        have_prev = true; pm0 = m0; pn0 = n0;
        if (!more) break;
        tile = next; m0 = nm0; n0 = nn0;
    }
    // the requests issued behind the last output tile must not outlive the workgroup
    x_drain_barrier(f1);
    mfma_step(f1);
    epilogue(pm0, pn0, false, 0, 0);
}

bool gemm2x_supported(const GemmWeight &W, int M_pad) {
    return W.type == GW_F16 && W.w16 && W.N % X_BN == 0 && W.K % (3 * X_BK) == 0 && M_pad % X_BM == 0 && M_pad > 0;
}

void launch_gemm2x(const GemmWeight &W, const half_t *A, const float *bias, const half_t *resid, half_t *C, int M_pad,
                   int epilogue, hipStream_t stream, int skew_override) {
    Gemm2xArgs a;
    a.A = A; a.w16 = W.w16; a.bias = bias; a.resid = resid; a.C = C;
    a.N = W.N; a.K = W.K; a.n_tiles_n = W.N / X_BN;
    a.n_tiles = a.n_tiles_n * (M_pad / X_BM);
    a.n_groups = 1;
    if ((size_t)W.N * W.K * 2 > (size_t)3 << 20 && W.N >= 4 * W.K && a.n_tiles_n % 2 == 0 && M_pad / X_BM >= 64) a.n_groups = 2;
    // half of an output tile's main loop (K / 32 stages of 16 MFMAs = 512 cycles each), in s_sleep units of 64 cycles
    a.skew = skew_override >= 0 ? skew_override : (W.K / X_BK) * 512 / 2 / 64 * 64;
    static int n_cu[MAX_HIP_DEVICES] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < MAX_HIP_DEVICES && !n_cu[dev]) {
        hipDeviceProp_t prop;
        n_cu[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8 ? prop.multiProcessorCount / 8 * 8 : 256;
    }
    const int cus = dev >= 0 && dev < MAX_HIP_DEVICES ? n_cu[dev] : 256;
    const int grid = std::min(2 * cus, (a.n_tiles + 7) / 8 * 8);       // two persistent workgroups per CU
    static DeviceFlags configured[3];
    auto go = [&](auto kernel, int e) {
        configure_once(configured[e], [&] { (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS); });
        BERT_LAUNCH(kernel, dim3(grid), dim3(256), X_LDS, stream, a);
    };
    switch (epilogue) {
        case EPI_BIAS: go(gemm2x_kernel<EPI_BIAS>, 0); break;
        case EPI_BIAS_GELU: go(gemm2x_kernel<EPI_BIAS_GELU>, 1); break;
        default: go(gemm2x_kernel<EPI_BIAS_RESID>, 2); break;
    }
}

}  // namespace bert_hip
