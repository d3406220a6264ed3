// wave_sum.hip — the butterfly sum of the row kernels (v += shfl_xor(v, 32), 16, ... 1: six ds_bpermute round trips) against the same
// pairs added through v_permlane32_swap (xor 32), ds_swizzle (xor 16 / 8 / 4) and DPP quad_perm (xor 2 / 1): the sums must agree in every bit.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ float wave_sum_fast(float v) {
    // xor 32: the halves swapped (v_permlane32_swap), 16 / 8 / 4: ds_swizzle bit-mask mode, 2 / 1: DPP quad_perm
    {
        // (by hand: this compiler's __builtin_amdgcn_permlane32_swap hands back its first result twice; two registers, or the instruction
        // copies the low half up and loses the high one; s_nop: the wait states between a VALU write and a lane-crossing read are ours inside an asm)
        float a = v, b = v;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
        v = a + b;
    }
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (16 << 10) | 0x1F));
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (8 << 10) | 0x1F));
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (4 << 10) | 0x1F));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    return v;
}
__device__ __forceinline__ float wave_sum_ref(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__global__ void k(const float *in, float *a, float *b) {
    const float v = in[threadIdx.x + blockIdx.x * 64];
    a[threadIdx.x + blockIdx.x * 64] = wave_sum_fast(v);
    b[threadIdx.x + blockIdx.x * 64] = wave_sum_ref(v);
}
int main() {
    const int N = 64 * 1000;
    float *in, *a, *b; hipMalloc(&in, N * 4); hipMalloc(&a, N * 4); hipMalloc(&b, N * 4);
    float *h = new float[N], *ha = new float[N], *hb = new float[N];
    srand(1); for (int i = 0; i < N; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * (1 + (i % 7) * 1000.f);
    hipMemcpy(in, h, N * 4, hipMemcpyHostToDevice);
    k<<<1000, 64>>>(in, a, b);
    hipMemcpy(ha, a, N * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, N * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < N; ++i) bad += memcmp(&ha[i], &hb[i], 4) != 0;
    printf("wave_sum fast vs shfl_xor butterfly: %d of %d lanes differ\n", bad, N);
    return bad != 0;
}
