// micro-benchmark: one "interval" (one k-tile of 64 between two barriers) of the tile-stream kernels for different
// wave decompositions of the same work per CU: 64 MFMA 32x32x16 per interval, 3-slot ring with counted vmcnt.
//   W8 : 8 waves (2/SIMD), wave tile 32 feat x 64 tok: 12 ds_read_b128 + 8 MFMA, 4 DMA pieces per wave
//   W4 : 4 waves (1/SIMD), wave tile 64 feat x 64 tok: 16 ds_read_b128 + 16 MFMA, 8 DMA pieces per wave
// ORDER 0: reads then MFMAs;  1: software pipelined per k-step (reads of step kk+1 issued before the MFMAs of kk)
// DMAKB: KiB moved by LDS-DMA per interval (32 = activation + weight tile, 16 = weight tile only, 0 = none)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define AS_GLOBAL(p) ((const __attribute__((address_space(1))) void *)(p))
#define AS_LDS(p) ((__attribute__((address_space(3))) void *)(p))

template <int P> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(P) : "memory"); }

template <int WAVES, int ORDER, int DMAKB>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) void k(const char *src, float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int FM = WAVES == 8 ? 1 : 2;          // 32-feature blocks per wave
    constexpr int P = DMAKB / WAVES;                // 1-KiB DMA pieces per wave per tile
    f32x16 acc[FM][2];
    for (int f = 0; f < FM; ++f) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[f][j][r] = 0.f;
    const char *gsrc = src + (size_t)blockIdx.x * 32768 + lane * 16;
    const int wq = WAVES == 8 ? (wave & 3) : (wave & 1), wt = WAVES == 8 ? (wave >> 2) : (wave >> 1);
    // fragment addresses: conflict-free pattern (row = lane & 31, 16-B chunk = 2*kk + (lane >> 5), XOR swizzle)
    auto off64 = [](int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); };
    int aW[FM][4], aY[2][4];
    for (int kk = 0; kk < 4; ++kk) {
        for (int f = 0; f < FM; ++f) aW[f][kk] = 16384 + off64((wq * FM + f) * 32 + (lane & 31), kk * 2 + (lane >> 5));
        for (int j = 0; j < 2; ++j) aY[j][kk] = off64(wt * 64 + j * 32 + (lane & 31), kk * 2 + (lane >> 5));
    }
    auto issue = [&](int slot) {
#pragma unroll
        for (int i = 0; i < P; ++i)
            __builtin_amdgcn_global_load_lds(AS_GLOBAL(gsrc + (wave * P + i) * 1024), AS_LDS(smem + slot * 32768 + (wave * P + i) * 1024), 16, 0, 0);
    };
    issue(0);
    issue(1);
    int slot = 0;
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        wait_vm<P>();
        if (ORDER < 2) {
            issue(slot >= 1 ? slot - 1 : 2);         // slot + 2 mod 3
            __builtin_amdgcn_sched_barrier(0);
        }
        const char *s = smem + slot * 32768;
        if (ORDER == 0) {
            f16x8 w[FM][4], y[2][4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int f = 0; f < FM; ++f) w[f][kk] = *(const f16x8 *)(s + aW[f][kk]);
#pragma unroll
                for (int j = 0; j < 2; ++j) y[j][kk] = *(const f16x8 *)(s + aY[j][kk]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int f = 0; f < FM; ++f)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[f][kk], y[j][kk], acc[f][j], 0, 0, 0);
        } else if (ORDER == 2) {
            // reads first, then the DMA pieces of tile t+2 spread between the MFMAs (one piece per MFMA group)
            f16x8 w[FM][4], y[2][4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int f = 0; f < FM; ++f) w[f][kk] = *(const f16x8 *)(s + aW[f][kk]);
#pragma unroll
                for (int j = 0; j < 2; ++j) y[j][kk] = *(const f16x8 *)(s + aY[j][kk]);
            }
            issue(slot >= 1 ? slot - 1 : 2);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int f = 0; f < FM; ++f)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[f][kk], y[j][kk], acc[f][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * (FM + 2), 0);
            if constexpr (P > 0) {
                constexpr int MPG = (FM * 8) / (P > 0 ? P : 1);
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, MPG, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
            }
        } else if (ORDER == 4 || ORDER == 5 || ORDER == 6) {
            // reads ordered by k-step, no hard fence before the MFMAs: the MFMAs of step 0 start when only its
            // fragments have landed (partial lgkmcnt); DMA pieces between the MFMAs
            f16x8 w[FM][4], y[2][4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int f = 0; f < FM; ++f) w[f][kk] = *(const f16x8 *)(s + aW[f][kk]);
#pragma unroll
                for (int j = 0; j < 2; ++j) y[j][kk] = *(const f16x8 *)(s + aY[j][kk]);
            }
            issue(slot >= 1 ? slot - 1 : 2);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int f = 0; f < FM; ++f)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[f][kk], y[j][kk], acc[f][j], 0, 0, 0);
            constexpr int RPK = FM + 2, MPK = FM * 2;          // reads / MFMAs per k-step
            if constexpr (ORDER == 4) {
                __builtin_amdgcn_sched_group_barrier(0x100, 4 * RPK, 0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    __builtin_amdgcn_sched_group_barrier(0x008, MPK, 0);
                    if constexpr (P >= 4) __builtin_amdgcn_sched_group_barrier(0x010, P / 4, 0);
                }
            } else if constexpr (ORDER == 5) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * RPK, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MPK, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, RPK, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MPK, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, RPK, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MPK, 0);
                if constexpr (P >= 2) __builtin_amdgcn_sched_group_barrier(0x010, P / 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MPK, 0);
                if constexpr (P >= 2) __builtin_amdgcn_sched_group_barrier(0x010, P / 2, 0);
            } else {
                // DMA first in the MFMA stream (lands sooner), reads progressive
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * RPK, 0);
                if constexpr (P >= 2) __builtin_amdgcn_sched_group_barrier(0x010, P / 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MPK, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, RPK, 0);
                if constexpr (P >= 2) __builtin_amdgcn_sched_group_barrier(0x010, P / 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MPK, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, RPK, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * MPK, 0);
            }
        } else if (ORDER == 3) {
            // pipelined per k-step AND the DMA pieces spread between the MFMAs
            f16x8 w[2][FM], y[2][2];
            issue(slot >= 1 ? slot - 1 : 2);
#pragma unroll
            for (int f = 0; f < FM; ++f) w[0][f] = *(const f16x8 *)(s + aW[f][0]);
#pragma unroll
            for (int j = 0; j < 2; ++j) y[0][j] = *(const f16x8 *)(s + aY[j][0]);
            __builtin_amdgcn_sched_group_barrier(0x100, FM + 2, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk < 3) {
#pragma unroll
                    for (int f = 0; f < FM; ++f) w[nxt][f] = *(const f16x8 *)(s + aW[f][kk + 1]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) y[nxt][j] = *(const f16x8 *)(s + aY[j][kk + 1]);
                }
#pragma unroll
                for (int f = 0; f < FM; ++f)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][f], y[cur][j], acc[f][j], 0, 0, 0);
                if (kk < 3) __builtin_amdgcn_sched_group_barrier(0x100, FM + 2, 0);
                if constexpr (P >= 4) {
                    constexpr int MPG = (FM * 2) / (P >= 4 ? P / 4 : 1);
#pragma unroll
                    for (int i = 0; i < P / 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, MPG, 0);
                        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, FM * 2, 0);
                    if (P > 0 && kk < P) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
            }
        } else {
            f16x8 w[2][FM], y[2][2];
#pragma unroll
            for (int f = 0; f < FM; ++f) w[0][f] = *(const f16x8 *)(s + aW[f][0]);
#pragma unroll
            for (int j = 0; j < 2; ++j) y[0][j] = *(const f16x8 *)(s + aY[j][0]);
            __builtin_amdgcn_sched_group_barrier(0x100, FM + 2, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk < 3) {
#pragma unroll
                    for (int f = 0; f < FM; ++f) w[nxt][f] = *(const f16x8 *)(s + aW[f][kk + 1]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) y[nxt][j] = *(const f16x8 *)(s + aY[j][kk + 1]);
                }
#pragma unroll
                for (int f = 0; f < FM; ++f)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][f], y[cur][j], acc[f][j], 0, 0, 0);
                // keep this order: the reads of step kk+1 in front of the MFMAs of step kk
                if (kk < 3) __builtin_amdgcn_sched_group_barrier(0x100, FM + 2, 0);    // DS reads
                __builtin_amdgcn_sched_group_barrier(0x008, FM * 2, 0);                // MFMA
            }
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
    long t1 = __builtin_readcyclecounter();
    float sum = 0;
    for (int f = 0; f < FM; ++f) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[f][j][r];
    if (sum == 12345.678f || (tid == 0 && blockIdx.x == 0)) out[0] = (float)(t1 - t0) / iters + (sum == 1.f ? 1 : 0);
}

// Cross-barrier software pipeline (8 waves): the MFMAs of k-steps 2,3 of tile t-1 run right after barrier t from
// fragments that are already in registers, covering the LDS latency of tile t's reads; k-steps 0,1 of tile t follow.
template <int DMAKB, int VAR>
__global__ __launch_bounds__(512, 2) void kx(const char *src, float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int P = DMAKB / 8;
    f32x16 acc[2];
    for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const char *gsrc = src + (size_t)blockIdx.x * 32768 + lane * 16;
    const int wq = wave & 3, wt = wave >> 2;
    auto off64 = [](int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); };
    int aW[4], aY[2][4];
    for (int kk = 0; kk < 4; ++kk) {
        aW[kk] = 16384 + off64(wq * 32 + (lane & 31), kk * 2 + (lane >> 5));
        for (int j = 0; j < 2; ++j) aY[j][kk] = off64(wt * 64 + j * 32 + (lane & 31), kk * 2 + (lane >> 5));
    }
    auto issue = [&](int slot) {
#pragma unroll
        for (int i = 0; i < P; ++i)
            __builtin_amdgcn_global_load_lds(AS_GLOBAL(gsrc + (wave * P + i) * 1024), AS_LDS(smem + slot * 32768 + (wave * P + i) * 1024), 16, 0, 0);
    };
    issue(0);
    issue(1);
    int slot = 0;
    f16x8 gw[2][2], gy[2][2][2];       // [buffer][kk-2][j]: carried fragments of k-steps 2,3
    for (int b = 0; b < 2; ++b) for (int k = 0; k < 2; ++k) { gw[b][k] = (f16x8)(_Float16)0; for (int j = 0; j < 2; ++j) gy[b][k][j] = (f16x8)(_Float16)0; }
    auto interval = [&](auto CUR) {
        constexpr int cur = decltype(CUR)::value, prv = cur ^ 1;
        wait_vm<P>();
        const char *s = smem + slot * 32768;
        f16x8 w0 = *(const f16x8 *)(s + aW[0]), y00 = *(const f16x8 *)(s + aY[0][0]), y01 = *(const f16x8 *)(s + aY[1][0]);
        f16x8 w1 = *(const f16x8 *)(s + aW[1]), y10 = *(const f16x8 *)(s + aY[0][1]), y11 = *(const f16x8 *)(s + aY[1][1]);
        gw[cur][0] = *(const f16x8 *)(s + aW[2]); gy[cur][0][0] = *(const f16x8 *)(s + aY[0][2]); gy[cur][0][1] = *(const f16x8 *)(s + aY[1][2]);
        gw[cur][1] = *(const f16x8 *)(s + aW[3]); gy[cur][1][0] = *(const f16x8 *)(s + aY[0][3]); gy[cur][1][1] = *(const f16x8 *)(s + aY[1][3]);
        issue(slot >= 1 ? slot - 1 : 2);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gw[prv][0], gy[prv][0][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gw[prv][0], gy[prv][0][1], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gw[prv][1], gy[prv][1][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gw[prv][1], gy[prv][1][1], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, y00, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, y01, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, y10, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, y11, acc[1], 0, 0, 0);
        if constexpr (VAR == 0) {
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if constexpr (P >= 4) __builtin_amdgcn_sched_group_barrier(0x010, P / 4, 0);
            }
        } else {
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if constexpr (P >= 2) __builtin_amdgcn_sched_group_barrier(0x010, P / 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if constexpr (P >= 2) __builtin_amdgcn_sched_group_barrier(0x010, P / 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        slot = slot == 2 ? 0 : slot + 1;
    };
    for (int it = 0; it < iters; it += 2) {
        interval(std::integral_constant<int, 0>{});
        interval(std::integral_constant<int, 1>{});
    }
    float sum = 0;
    for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
    for (int b = 0; b < 2; ++b) for (int k = 0; k < 2; ++k) sum += (float)gw[b][k][0] + (float)gy[b][k][0][0] + (float)gy[b][k][1][0];
    if (sum == 12345.678f) out[0] = sum;
}

template <int DMAKB, int VAR> void runx(const char *src, float *out, const char *name) {
    auto kern = kx<DMAKB, VAR>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 4000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kern<<<256, 512, 3 * 32768 + 32768>>>(src, out, 10);
    hipEventRecord(a);
    kern<<<256, 512, 3 * 32768 + 32768>>>(src, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s %8.1f ns/interval  -> %6.0f TFLOP/s at 256 CUs\n", name, ms * 1e6 / iters,
           256.0 * 64 * 2 * 32 * 32 * 16 / (ms * 1e6 / iters) / 1e3);
}

template <int WAVES, int ORDER, int DMAKB> void run(const char *src, float *out, const char *name) {
    auto kern = k<WAVES, ORDER, DMAKB>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 4000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kern<<<256, WAVES * 64, 3 * 32768 + 32768>>>(src, out, 10);
    hipEventRecord(a);
    kern<<<256, WAVES * 64, 3 * 32768 + 32768>>>(src, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s %8.1f ns/interval  -> %6.0f TFLOP/s at 256 CUs\n", name, ms * 1e6 / iters,
           256.0 * 64 * 2 * 32 * 32 * 16 / (ms * 1e6 / iters) / 1e3);
}
int main() {
    char *src; float *out;
    hipMalloc(&src, 256 * 32768 + 65536); hipMemset(src, 0, 256 * 32768 + 65536); hipMalloc(&out, 64);
    run<8, 0, 32>(src, out, "W8 reads->mfma, 32 KiB DMA (today)");
    run<8, 0, 16>(src, out, "W8 reads->mfma, 16 KiB DMA");
    run<8, 0, 0>(src, out, "W8 reads->mfma, no DMA");
    run<8, 1, 32>(src, out, "W8 pipelined, 32 KiB DMA");
    run<4, 0, 32>(src, out, "W4 reads->mfma, 32 KiB DMA");
    run<4, 1, 32>(src, out, "W4 pipelined, 32 KiB DMA");
    run<4, 1, 16>(src, out, "W4 pipelined, 16 KiB DMA");
    run<4, 1, 0>(src, out, "W4 pipelined, no DMA");
    run<4, 0, 0>(src, out, "W4 reads->mfma, no DMA");
    run<8, 2, 32>(src, out, "W8 reads, mfma+DMA interleaved, 32 KiB");
    run<8, 2, 16>(src, out, "W8 reads, mfma+DMA interleaved, 16 KiB");
    run<4, 2, 32>(src, out, "W4 reads, mfma+DMA interleaved, 32 KiB");
    run<4, 3, 32>(src, out, "W4 pipelined + DMA interleaved, 32 KiB");
    run<4, 3, 16>(src, out, "W4 pipelined + DMA interleaved, 16 KiB");
    run<8, 3, 32>(src, out, "W8 pipelined + DMA interleaved, 32 KiB");
    run<8, 4, 32>(src, out, "W8 order4 (reads by k, MFMA|DMA), 32 KiB");
    run<8, 5, 32>(src, out, "W8 order5 (reads split, DMA late), 32 KiB");
    run<8, 6, 32>(src, out, "W8 order6 (reads split, DMA early), 32 KiB");
    runx<32, 0>(src, out, "W8 cross-barrier pipeline v0, 32 KiB");
    runx<32, 1>(src, out, "W8 cross-barrier pipeline v1, 32 KiB");
    runx<16, 0>(src, out, "W8 cross-barrier pipeline v0, 16 KiB");
    runx<16, 1>(src, out, "W8 cross-barrier pipeline v1, 16 KiB");
    runx<0, 1>(src, out, "W8 cross-barrier pipeline v1, no DMA");
    run<4, 4, 32>(src, out, "W4 order4, 32 KiB");
    run<4, 5, 32>(src, out, "W4 order5, 32 KiB");
    run<4, 6, 32>(src, out, "W4 order6, 32 KiB");
    run<8, 4, 16>(src, out, "W8 order4, 16 KiB");
    run<8, 5, 16>(src, out, "W8 order5, 16 KiB");
    return 0;
}
