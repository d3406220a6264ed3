// mfma_chain.hip — 16 v_mfma_f32_32x32x16_f16 per round into four accumulators, the A fragment of each read from LDS (ds_read_b128,
// requested one MFMA ahead, like the attention kernels' S^T phase), in three orders:
//   A  c0 c0 c0 c0 c1 c1 c1 c1 c2 ... (an accumulator's four k-steps back to back: attention.hip's S^T nest)
//   B  c0 c1 c0 c1 c0 c1 c0 c1 c2 c3 c2 c3 ...  (two accumulators alternating: what P.V does)
//   C  c0 c1 c2 c3 c0 c1 c2 c3 ...    (round robin)
// and D = order A without the LDS reads (fragments in registers).  One and two waves per SIMD, every CU busy.  Shader cycles per round.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ORDER, int DIST>
__global__ __launch_bounds__(512) void kernel(const f16x8 *src, int rounds, long long *cycles, float *sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((f16x8 *)lds)[i] = src[i & 4095];
    f16x8 q[4];
    for (int i = 0; i < 4; ++i) q[i] = src[(gwave * 4 + i) & 4095];
    f32x16 acc[4] = {};
    __syncthreads();
    // (attention.hip's K tile: row l31, 16-byte chunk (kk * 2 + hi) ^ ((row >> 1) & 7): conflict-free ds_read_b128)
    const int l31 = lane & 31;
    const __attribute__((address_space(3))) char *base = (const __attribute__((address_space(3))) char *)lds + l31 * 128 + ((((lane >> 5)) ^ ((l31 >> 1) & 7)) << 4);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        // fragment of MFMA i requested DIST MFMAs ahead (a ring of DIST + 1 register sets)
        f16x8 ring[DIST + 1];
#pragma unroll
        for (int j = 0; j < DIST; ++j) ring[j] = *(const __attribute__((address_space(3))) f16x8 *)(base + (j * 4096 & 0xffff) + ((r & 1) << 5));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kt = ORDER == 1 ? (i >> 3) * 2 + (i & 1) : ORDER == 2 ? (i & 3) : (i >> 2);
            const int kk = ORDER == 1 ? (i >> 1) & 3 : ORDER == 2 ? (i >> 2) : (i & 3);
            if (ORDER != 3 && i + DIST < 16) ring[(i + DIST) % (DIST + 1)] = *(const __attribute__((address_space(3))) f16x8 *)(base + ((i + DIST) * 4096 & 0xffff) + ((r & 1) << 5));
            acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[ORDER == 3 ? 0 : i % (DIST + 1)], q[kk], acc[kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
    if (s == 1234.5f) sink[gwave] = s;
    if (lane == 0) cycles[gwave] = t1 - t0;
}

template <int ORDER, int DIST>
static double run(int threads, const f16x8 *src, long long *cyc, float *sink) {
    const int rounds = 2000, grid = 256;
    hipFuncSetAttribute((const void *)kernel<ORDER, DIST>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    kernel<ORDER, DIST><<<grid, threads, 65536>>>(src, rounds, cyc, sink);
    kernel<ORDER, DIST><<<grid, threads, 65536>>>(src, rounds, cyc, sink);
    hipDeviceSynchronize();
    const int nw = grid * threads / 64;
    std::vector<long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    return sum / nw / rounds;
}

int main() {
    f16x8 *src; long long *cyc; float *sink;
    hipMalloc(&src, 4096 * sizeof(f16x8)); hipMalloc(&cyc, 4096 * sizeof(long long)); hipMalloc(&sink, 65536);
    std::vector<_Float16> h(4096 * 8);
    srand(1);
    for (auto &x : h) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.125f);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("shader cycles per round of 16 MFMAs (per MFMA)      1 wave/SIMD        2 waves/SIMD\n");
    const char *names[] = {"A  an accumulator's 4 k-steps back to back, fragment 1 ahead", "B  two accumulators alternating, 1 ahead", "C  four accumulators round robin, 1 ahead",
                           "D  order A, fragments in registers", "A  fragment 2 MFMAs ahead", "A  fragment 3 MFMAs ahead", "A  fragment 4 MFMAs ahead"};
    double r[7][2] = {{run<0, 1>(256, src, cyc, sink), run<0, 1>(512, src, cyc, sink)}, {run<1, 1>(256, src, cyc, sink), run<1, 1>(512, src, cyc, sink)},
                      {run<2, 1>(256, src, cyc, sink), run<2, 1>(512, src, cyc, sink)}, {run<3, 1>(256, src, cyc, sink), run<3, 1>(512, src, cyc, sink)},
                      {run<0, 2>(256, src, cyc, sink), run<0, 2>(512, src, cyc, sink)}, {run<0, 3>(256, src, cyc, sink), run<0, 3>(512, src, cyc, sink)},
                      {run<0, 4>(256, src, cyc, sink), run<0, 4>(512, src, cyc, sink)}};
    for (int i = 0; i < 7; ++i) printf("%-62s %7.1f (%5.1f)   %7.1f (%5.1f)\n", names[i], r[i][0], r[i][0] / 16, r[i][1], r[i][1] / 16);
    return 0;
}
