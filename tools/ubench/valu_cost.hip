// valu_cost.hip — what the softmax's candidate instructions cost a wave (gfx950), alone and beside MFMAs: per round a wave issues
// 64 instances of ONE instruction (eight independent destination registers in turn) and 0 or 8 v_mfma_f32_32x32x16_f16 (one
// per eight instances, independent accumulators), one or two waves per SIMD, every CU busy.  Prints shader cycles per
// instruction instance: (round(kind) - round(nothing)) / 64.   usage: valu_cost
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ONE(K, STR) if constexpr (KIND == K) asm volatile(STR : "+v"(d[i & 7]) : "v"(a), "v"(b), "v"(c))
template <int KIND, int NM>
__global__ __launch_bounds__(512) void kernel(const f16x8 *src, const float *fsrc, int rounds, long long *cycles, float *sink) {
    const int lane = threadIdx.x & 63, gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    f16x8 fa[2], fb[2];
    for (int i = 0; i < 2; ++i) { fa[i] = src[(gwave * 4 + i) & 4095]; fb[i] = src[(gwave * 4 + 2 + i) & 4095]; }
    f32x16 acc[8] = {};
    float d[8];
    for (int i = 0; i < 8; ++i) d[i] = fsrc[(lane + i) & 255] * 0.001f;
    float a = fsrc[lane & 255] * 0.5f, b = fsrc[(lane + 17) & 255] * 0.25f, c = fsrc[(lane + 31) & 255];
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (NM && (i & 7) == 0) acc[(i >> 3) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i & 1], fb[(i >> 3) & 1], acc[(i >> 3) & 7], 0, 0, 0);
            ONE(1, "v_fma_f32 %0, %1, %2, %0");
            ONE(2, "v_fma_mixlo_f16 %0, %1, %2, -%3");
            ONE(3, "v_fma_mixhi_f16 %0, %1, %2, -%3");
            ONE(4, "v_exp_f32_e32 %0, %1");
            ONE(5, "v_exp_f16_e32 %0, %1");
            ONE(6, "v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1");
            ONE(7, "v_dot2c_f32_f16_e32 %0, %1, %2");
            ONE(8, "v_cvt_pk_f16_f32 %0, %1, %2");
            ONE(9, "v_add_f32_e32 %0, %1, %0");
            ONE(10, "v_max3_f32 %0, %1, %2, %0");
            ONE(11, "v_pk_mul_f16 %0, %1, %2");
            ONE(12, "v_pk_add_f16 %0, %1, %0");
            ONE(13, "v_cvt_f32_f16_e32 %0, %1");
            ONE(14, "v_pk_fma_f16 %0, %1, %2, %0");
            ONE(15, "v_mul_f32_e32 %0, %1, %2");
            ONE(16, "v_dot2_f32_f16 %0, %1, %2, %0");
            ONE(17, "v_exp_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1");
            ONE(18, "v_pack_b32_f16 %0, %1, %2");
            ONE(19, "v_perm_b32 %0, %1, %2, %3");
            ONE(20, "v_mov_b32_e32 %0, %1");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) { s += d[i]; for (int k = 0; k < 16; ++k) s += acc[i][k]; }
    if (s == 1234.5f) sink[gwave] = s;
    if (lane == 0) cycles[gwave] = t1 - t0;
}

static const char *NAMES[] = {"(nothing)", "v_fma_f32", "v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_exp_f32", "v_exp_f16", "v_exp_f16_sdwa hi<-hi preserve",
                              "v_dot2c_f32_f16", "v_cvt_pk_f16_f32", "v_add_f32", "v_max3_f32", "v_pk_mul_f16", "v_pk_add_f16", "v_cvt_f32_f16",
                              "v_pk_fma_f16", "v_mul_f32", "v_dot2_f32_f16 (vop3p)", "v_exp_f16_sdwa dword<-hi", "v_pack_b32_f16", "v_perm_b32", "v_mov_b32"};

template <int KIND, int NM>
static double run(int threads, const f16x8 *src, const float *fsrc, long long *cyc, float *sink) {
    const int rounds = 2000, grid = 256;
    kernel<KIND, NM><<<grid, threads>>>(src, fsrc, rounds, cyc, sink);
    kernel<KIND, NM><<<grid, threads>>>(src, fsrc, rounds, cyc, sink);
    hipDeviceSynchronize();
    const int nw = grid * threads / 64;
    std::vector<long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    return sum / nw / rounds;             // cycles per round
}

template <int KIND>
static void row(const f16x8 *src, const float *fsrc, long long *cyc, float *sink, const double *base) {
    const double r[4] = {run<KIND, 0>(256, src, fsrc, cyc, sink), run<KIND, 0>(512, src, fsrc, cyc, sink), run<KIND, 8>(256, src, fsrc, cyc, sink),
                         run<KIND, 8>(512, src, fsrc, cyc, sink)};
    printf("%-34s", NAMES[KIND]);
    for (int i = 0; i < 4; ++i) printf("  %7.1f (%5.2f)", r[i], (r[i] - base[i]) / 64.0);
    printf("\n");
}

int main() {
    f16x8 *src; float *fsrc; long long *cyc; float *sink;
    hipMalloc(&src, 4096 * sizeof(f16x8)); hipMalloc(&fsrc, 256 * 4); hipMalloc(&cyc, 4096 * sizeof(long long)); hipMalloc(&sink, 65536);
    std::vector<_Float16> h(4096 * 8);
    std::vector<float> f(256);
    srand(1);
    for (auto &x : h) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.125f);
    for (auto &x : f) x = rand() / (float)RAND_MAX - 0.5f;
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(fsrc, f.data(), f.size() * 4, hipMemcpyHostToDevice);
    printf("cycles per round of 64 instances [+ 8 MFMAs] (and per instance above the empty round)\n");
    printf("%-34s  %15s  %15s  %15s  %15s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "1 w + 8 MFMA", "2 w + 8 MFMA");
    double base[4] = {0, 0, 0, 0};
    base[0] = run<0, 0>(256, src, fsrc, cyc, sink); base[1] = run<0, 0>(512, src, fsrc, cyc, sink);
    base[2] = run<0, 8>(256, src, fsrc, cyc, sink); base[3] = run<0, 8>(512, src, fsrc, cyc, sink);
    printf("%-34s  %7.1f          %7.1f          %7.1f          %7.1f\n", NAMES[0], base[0], base[1], base[2], base[3]);
    row<1>(src, fsrc, cyc, sink, base); row<9>(src, fsrc, cyc, sink, base); row<15>(src, fsrc, cyc, sink, base); row<20>(src, fsrc, cyc, sink, base);
    row<10>(src, fsrc, cyc, sink, base); row<2>(src, fsrc, cyc, sink, base); row<3>(src, fsrc, cyc, sink, base); row<8>(src, fsrc, cyc, sink, base);
    row<4>(src, fsrc, cyc, sink, base); row<5>(src, fsrc, cyc, sink, base); row<6>(src, fsrc, cyc, sink, base); row<17>(src, fsrc, cyc, sink, base);
    row<7>(src, fsrc, cyc, sink, base); row<16>(src, fsrc, cyc, sink, base); row<11>(src, fsrc, cyc, sink, base); row<12>(src, fsrc, cyc, sink, base);
    row<14>(src, fsrc, cyc, sink, base); row<13>(src, fsrc, cyc, sink, base); row<18>(src, fsrc, cyc, sink, base); row<19>(src, fsrc, cyc, sink, base);
    return 0;
}
