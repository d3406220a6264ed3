// store_beside_mfma.hip — what a global store costs a wave that is multiplying: one wave per SIMD runs 64 MFMAs
// (v_mfma_f32_32x32x16_f16, 16 independent accumulators) per round and issues 0 / 1 / 2 / 4 stores of 16 bytes per lane
// between them, evenly spaced.  Prints shader cycles per round with and without the stores, for a line that stays in L2 and
// for a streaming footprint.  usage: store_beside_mfma
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_AS(p) ((const __attribute__((address_space(1))) void *)(p))
#define LDS_AS(p) ((__attribute__((address_space(3))) void *)(p))
// NLOAD: LDS-DMA pieces (1 KiB, global_load_lds_dwordx4) per round besides the stores, from an L2-resident 16 MiB region
// PHASED: 0 = loads and stores evenly interleaved over the round; p > 0: loads behind MFMAs 0..NLOAD-1, stores behind MFMAs p..p+15
#ifndef PHASED
#define PHASED 0
#endif
#ifndef NLOAD
#define NLOAD 0
#endif
template <int NSTORE>
__global__ __launch_bounds__(256) void kernel(char *buf, const f16x8 *src, int rounds, size_t round_stride, int wrap, size_t pitch, long long *cycles, float *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gwave = blockIdx.x * 4 + wave;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(gwave * 8 + i) & 4095]; b[i] = src[(gwave * 8 + 4 + i) & 4095]; }
    f32x16 acc[16] = {};
    char *base = buf + (size_t)gwave * 4096 * (wrap + 1) + (size_t)(lane >> 3) * 512 + (lane & 7) * 16;   // 8 rows x 128 B
    // GEMM-like output walk (pitch > 0): the workgroup owns 256 x 256 f16 tiles of a [M][pitch / 2] matrix, tile t of the
    // workgroup = tile index blockIdx.x + 256 t, n fastest; a wave owns 128 rows x 128 columns; round r of a tile stores
    // block (j = (r >> 1) & 3, ip = r & 1): 4 stores of 8 rows x 128 B
    const int wf = wave & 1, wt = wave >> 1;
    u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
    extern __shared__ char lds[];
    const char *lsrc = buf + ((size_t)1 << 30) + (size_t)(gwave & 255) * 65536 + lane * 16;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        char *pr = base + (size_t)(r & wrap) * round_stride;
        size_t row_step = 1024;
        if (pitch) {
            // (no division here: the walk's arithmetic would sit outside the MFMAs and be what is measured)
            const int j = (r >> 1) & 3, ip = r & 1;
            pr = buf + ((size_t)(blockIdx.x + 256 * ((r >> 3) & 1)) * 256 + wt * 128 + j * 32 + (lane >> 3)) * pitch + wf * 256 + ip * 128 + (lane & 7) * 16;
            row_step = 8 * pitch;
        }
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            acc[m & 15] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[(m >> 2) & 3], acc[m & 15], 0, 0, 0);
            // PHASED: the loads behind the first NLOAD MFMAs, the stores inside the last 16 — apart in time
            if (PHASED ? (NLOAD && m < NLOAD) : (NLOAD && m % (64 / (NLOAD ? NLOAD : 1)) == 1))
                __builtin_amdgcn_global_load_lds(GLOBAL_AS(lsrc + (size_t)((r * 16 + m) & 63) * 1024), LDS_AS(lds + wave * 16384 + (m & 15) * 1024), 16, 0, 0);
            if (PHASED ? (NSTORE && m >= PHASED && (m - PHASED) % (16 / (NSTORE ? NSTORE : 1)) == 1 && m < PHASED + 16) : (NSTORE && m % (64 / (NSTORE ? NSTORE : 1)) == 3)) *(u32x4 *)(pr + (PHASED ? ((m - PHASED) / (16 / (NSTORE ? NSTORE : 1))) : (m / (64 / (NSTORE ? NSTORE : 1)))) * row_step) = v;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
    if (s == 1234.5f) sink[gwave] = s;
    if (lane == 0) cycles[gwave] = t1 - t0;
}

int main() {
    char *buf; f16x8 *src; long long *cyc; float *sink;
    const size_t bytes = (size_t)2 << 30;
    hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
    hipMalloc(&src, 4096 * sizeof(f16x8)); hipMalloc(&cyc, 1024 * sizeof(long long)); hipMalloc(&sink, 4096);
    std::vector<_Float16> h(4096 * 8);
    srand(1);
    for (auto &x : h) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.125f);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    printf("LDS-DMA pieces per round beside the stores: %d, phased %d\n", NLOAD, PHASED);
    const int rounds = 400;
    for (size_t pitch : {(size_t)4608})
    for (int wrap : {0, 63}) {
        if (pitch && !wrap) continue;
        double base_cyc = 0;
        for (int ns : {0, 1, 2, 4, 8}) {
            for (int rep = 0; rep < 3; ++rep) {
                switch (ns) {
                    case 0: hipLaunchKernelGGL(kernel<0>, dim3(256), dim3(256), 65536, 0, buf, src, rounds, (size_t)8192, wrap, pitch, cyc, sink); break;
                    case 1: hipLaunchKernelGGL(kernel<1>, dim3(256), dim3(256), 65536, 0, buf, src, rounds, (size_t)8192, wrap, pitch, cyc, sink); break;
                    case 2: hipLaunchKernelGGL(kernel<2>, dim3(256), dim3(256), 65536, 0, buf, src, rounds, (size_t)8192, wrap, pitch, cyc, sink); break;
                    case 4: hipLaunchKernelGGL(kernel<4>, dim3(256), dim3(256), 65536, 0, buf, src, rounds, (size_t)8192, wrap, pitch, cyc, sink); break;
                    default: hipLaunchKernelGGL(kernel<8>, dim3(256), dim3(256), 65536, 0, buf, src, rounds, (size_t)8192, wrap, pitch, cyc, sink); break;
                }
                hipDeviceSynchronize();
            }
            std::vector<long long> c(1024);
            hipMemcpy(c.data(), cyc, c.size() * sizeof(long long), hipMemcpyDeviceToHost);
            double avg = 0;
            for (auto x : c) avg += x;
            avg /= c.size() * rounds;
            if (ns == 0) base_cyc = avg;
            if (pitch) printf("[M][%zu] f16 output, 256 x 256 tiles: ", pitch / 2);
            printf("%s: %d stores per 64 MFMAs: %.0f cycles per round (%+.0f, %.0f per store)\n", wrap ? "streaming (64 x 8 KiB per wave)" : "L2-resident",
                   ns, avg, avg - base_cyc, ns ? (avg - base_cyc) / ns : 0.0);
        }
    }
    return 0;
}
