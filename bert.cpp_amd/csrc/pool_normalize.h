// pool_normalize.h — mean over the tokens of a sentence, then y / ||y||_2 (reference bert.cpp:904-913): the body shared by
// pool_normalize_kernel (misc_kernels.hip: a workgroup of 256 threads per sentence) and the epilogue of model_kernel.hip
// (a workgroup pools its window's sentences itself: threads 0..255 of its 512 work, all of them meet at the barriers).
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace bert_hip {

typedef _Float16 pool_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pool_f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float pool_wave_sum(float v) { return wave_sum_f32(v); }

// Sentence b = rows tok0 .. tok0 + n - 1 of x.  Wave w (of four) sums tokens w, w+4, ... over 16-byte (H % 8 == 0) or 4-byte row
// reads, the four partial rows are combined through LDS: part = [4][H] floats + 4.  Every thread of the workgroup calls this
// (uniform arguments); threads with working == false only keep the barriers company.
__device__ __forceinline__ void pool_normalize_sentence(const half_t *x, int tok0, int n, int b, int H, int max_len, int *status,
                                                        float *out, float *part, int tid, bool working) {
    const int wave = tid >> 6, lane = tid & 63;
    if (n <= 0 || n > max_len) {
        // the batch does not keep the caller's promise (bert_hip_eval_packed_device: max_len): the kernels upstream were
        // chosen and sized for max_len, so this sentence's result is not trustworthy -> NaN row, status word
        if (working) {
            for (int e = tid; e < H; e += 256) out[(size_t)b * H + e] = __builtin_nanf("");
            if (tid == 0 && status) atomicOr(status, 1);
        }
        return;
    }
    const float invn = 1.0f / (float)n;
    if (working) {
        if (H % 8 == 0) {
            // 16-byte runs per lane (same per-element summation order as the pair loop below)
            for (int c = lane; c < H / 8; c += 64) {
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
                for (int t = wave; t < n; t += 4) {
                    const pool_f16x8 v = *(const pool_f16x8 *)(x + (size_t)(tok0 + t) * H + 8 * c);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += (float)v[i] * invn;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) part[wave * H + 8 * c + i] = acc[i];
            }
        } else {
            for (int e = 2 * lane; e < H; e += 128) {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
                for (int t = wave; t < n; t += 4) {
                    const pool_f16x2 v = *(const pool_f16x2 *)(x + (size_t)(tok0 + t) * H + e);
                    a0 += (float)v[0] * invn; a1 += (float)v[1] * invn;
                }
                part[wave * H + e] = a0; part[wave * H + e + 1] = a1;
            }
        }
    }
    __syncthreads();
    float sq = 0.f;
    if (working) {
        for (int e = tid; e < H; e += 256) {
            const float a = (part[e] + part[H + e]) + (part[2 * H + e] + part[3 * H + e]);
            part[e] = a;
            sq += a * a;
        }
        sq = pool_wave_sum(sq);
    }
    __syncthreads();
    float *red = part + 4 * H;
    if (working && lane == 0) red[wave] = sq;
    __syncthreads();
    if (working) {
        const float scale = 1.0f / sqrtf((red[0] + red[1]) + (red[2] + red[3]));
        for (int e = tid; e < H; e += 256) out[(size_t)b * H + e] = part[e] * scale;
    }
}

}  // namespace bert_hip
