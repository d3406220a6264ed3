// store_issue.hip — what a CU's vector-memory path costs per 16-byte-per-lane instruction when every wave of the chip issues
// them at once (the shape of a GEMM epilogue / a tile prologue): global stores and LDS-DMA loads, per wave 64 instructions
// of 1 KiB, in three address shapes: one contiguous KiB, 8 rows x 128 B (row pitch 1536 B: a [T][768] f16 matrix), 16 rows x 64 B.
// Prints shader cycles per instruction per CU (wall of the slowest wave / instructions per CU) and the GB/s of the whole chip.
// usage: store_issue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define GLOBAL_AS(p) ((const __attribute__((address_space(1))) void *)(p))
#define LDS_AS(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int N_INSTR = 64;
// every wave rewrites WRAP + 1 instruction footprints: 0 .. small = the lines stay in L2 (issue cost), 63 = streaming (HBM)
#ifndef WRAP
#define WRAP 1
#endif

// shape 0: lane l -> 16 B at l * 16 of a 1 KiB block; 1: row l / 8 (pitch), 16 B chunk l % 8; 2: row l / 4, chunk l % 4
__device__ __forceinline__ size_t lane_offset(int shape, int lane, size_t pitch) {
    if (shape == 0) return (size_t)lane * 16;
    if (shape == 1) return (size_t)(lane >> 3) * pitch + (lane & 7) * 16;
    return (size_t)(lane >> 2) * pitch + (lane & 3) * 16;
}

template <bool STORE>
__global__ __launch_bounds__(512) void issue_kernel(char *buf, int shape, size_t pitch, size_t wave_stride, size_t instr_stride,
                                                    long long *cycles) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gwave = blockIdx.x * (blockDim.x >> 6) + wave;
    char *base = buf + (size_t)gwave * wave_stride + lane_offset(shape, lane, pitch);
    uint4 v = make_uint4(lane, wave, gwave, 7);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
    for (int i = 0; i < N_INSTR; ++i) {
        if (STORE) *(uint4 *)(base + (size_t)(i & WRAP) * instr_stride) = v;
        else __builtin_amdgcn_global_load_lds(GLOBAL_AS(base + (size_t)(i & WRAP) * instr_stride), LDS_AS(lds + wave * 8192 + (i & 7) * 1024), 16, 0, 0);
    }
    const long long t_issue = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { cycles[gwave * 2] = t_issue - t0; cycles[gwave * 2 + 1] = t1 - t0; }
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    char *buf; long long *cyc;
    hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
    hipMalloc(&cyc, 4096 * 2 * sizeof(long long));
    hipFuncSetAttribute((const void *)issue_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int store = 1; store >= 0; --store)
        for (int threads : {512, 256})
            for (int shape = 0; shape < 3; ++shape) {
                const int waves = 256 * threads / 64;
                const size_t pitch = 1536;
                // shape 0: instruction i at +1 KiB; shapes 1, 2: the next 8 (16) rows of the wave's 64-feature column block
                const size_t instr_stride = shape == 0 ? 1024 : shape == 1 ? 8 * pitch : 16 * pitch;
                const size_t wave_stride = shape == 0 ? (WRAP + 1) * 1024 : (size_t)(WRAP + 1) * (shape == 1 ? 8 : 16) * pitch;
                if (wave_stride * waves > bytes) { printf("buffer too small\n"); return 1; }
                float best = 1e9f; double issue = 0, total = 0;
                for (int rep = 0; rep < 5; ++rep) {
                    hipEventRecord(e0);
                    if (store) hipLaunchKernelGGL(issue_kernel<true>, dim3(256), dim3(threads), 65536, 0, buf, shape, pitch, wave_stride, instr_stride, cyc);
                    else hipLaunchKernelGGL(issue_kernel<false>, dim3(256), dim3(threads), 65536, 0, buf, shape, pitch, wave_stride, instr_stride, cyc);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) {
                        best = ms;
                        std::vector<long long> h(waves * 2);
                        hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
                        issue = total = 0;
                        for (int w = 0; w < waves; ++w) { issue += h[2 * w]; total += h[2 * w + 1]; }
                        issue /= waves; total /= waves;
                    }
                }
                const int per_cu = N_INSTR * threads / 64;
                printf("%s %d waves/CU shape %d (%s): issue %.0f cyc/wave = %.1f cyc per instruction per CU; acked %.0f cyc = %.1f per instruction per CU, %.1f B/cyc/CU; kernel %.1f us\n",
                       store ? "store" : "lds-dma", threads / 64, shape, shape == 0 ? "1 KiB contiguous" : shape == 1 ? "8 rows x 128 B" : "16 rows x 64 B",
                       issue, issue / per_cu, total, total / per_cu, per_cu * 1024.0 / total, best * 1e3);
            }
    return 0;
}
