// mfma_power.hip — what the matrix cores of an MI355X sustain under the board's power limit: v_mfma_f32_32x32x16_f16 back to back
// from registers (no LDS, no memory), 256 CUs, one or two waves per SIMD, operands zero / small integers / random f16.
// Prints TFLOP/s per fill over a ~0.4 s run (long enough for the power management to settle).  usage: mfma_power [seconds]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_loop(const f16x8 *src, float *sink, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) & 65535]; b[i] = src[(tid * 8 + 4 + i) & 65535]; }
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[tid] = s;
}

int main(int argc, char **argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 0.4;
    f16x8 *d; float *sink;
    hipMalloc(&d, 65536 * sizeof(f16x8)); hipMalloc(&sink, 1 << 22);
    for (int fill = 0; fill < 3; ++fill)
        for (int threads : {256, 512}) {
            std::vector<_Float16> h(65536 * 8);
            srand(1);
            for (auto &v : h) v = fill == 0 ? (_Float16)0.f : fill == 1 ? (_Float16)(float)(rand() % 3 - 1) : (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.125f);
            hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
            const int iters = 20000;
            mfma_loop<<<256, threads>>>(d, sink, 100);
            hipDeviceSynchronize();
            int launches = 0;
            auto t0 = std::chrono::steady_clock::now();
            double dt = 0;
            while (dt < secs) {
                mfma_loop<<<256, threads>>>(d, sink, iters);
                hipDeviceSynchronize();
                ++launches;
                dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            const double flops = (double)launches * 256 * (threads / 64) * iters * 16.0 * 32768.0;
            printf("fill %s, %d waves/SIMD: %.0f TFLOP/s (%.2f of 2500)\n", fill == 0 ? "zeros" : fill == 1 ? "{-1,0,1}" : "random f16",
                   threads / 256, flops / dt / 1e12, flops / dt / 2.5e15);
            fflush(stdout);
        }
    return 0;
}
