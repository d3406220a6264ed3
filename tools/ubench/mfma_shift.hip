// mfma_shift.hip — is the P·V accumulation of the attention kernels invariant under a shift of a sentence's keys by 8 window slots?
// v_mfma_f32_32x32x16_f16 sums 16 products per instruction; the attention kernels feed window key (16 st + 4 hi + e) [e < 4] and
// (16 st + 8 + 4 hi + e - 4) [e >= 4] into k-slot 8 hi + e.  A sentence of 32 keys at window slot 0 is two instructions; at slot
// 8 it is three, with exact zeros (p = 0) in the slots of its neighbours.  If the results differ in any bit, 8-slot window
// granularity cannot keep "a sentence's bits do not depend on where it sits in a window".      usage: mfma_shift
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// vt [32 dv][48 window keys], p [32 q][48 window keys] (f16), out [32 dv][32 q] f32 : O^T = V^T P^T over `steps` 16-key steps
__global__ void pv(const _Float16 *vt, const _Float16 *p, int steps, float *out) {
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    f32x16 o = {};
    for (int st = 0; st < steps; ++st) {
        f16x8 vf, pf;
        for (int e = 0; e < 8; ++e) {
            const int key = 16 * st + 4 * hi + (e < 4 ? e : e + 4);
            vf[e] = vt[l31 * 48 + key];
            pf[e] = p[l31 * 48 + key];
        }
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = o[r];
}

int main() {
    _Float16 *dv, *dp; float *dout;
    hipMalloc(&dv, 32 * 48 * 2); hipMalloc(&dp, 32 * 48 * 2); hipMalloc(&dout, 32 * 32 * 4);
    srand(3);
    long long diff = 0, total = 0; double maxrel = 0;
    for (int trial = 0; trial < 200; ++trial) {
        std::vector<_Float16> v(32 * 32), p(32 * 32);
        for (auto &x : v) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
        for (auto &x : p) x = (_Float16)(rand() / (float)RAND_MAX);          // softmax numerators: (0, 1]
        std::vector<float> o[2];
        for (int shift = 0; shift < 2; ++shift) {
            std::vector<_Float16> hv(32 * 48, (_Float16)0.f), hp(32 * 48, (_Float16)0.f);
            for (int r = 0; r < 32; ++r)
                for (int k = 0; k < 32; ++k) { hv[r * 48 + 8 * shift + k] = v[r * 32 + k]; hp[r * 48 + 8 * shift + k] = p[r * 32 + k]; }
            // (the neighbours' V values are not zero in the kernel, only their probabilities: fill the other slots of V with noise)
            for (int r = 0; r < 32; ++r)
                for (int k = 0; k < 48; ++k) if (k < 8 * shift || k >= 8 * shift + 32) hv[r * 48 + k] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
            hipMemcpy(dv, hv.data(), hv.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
            pv<<<1, 64>>>(dv, dp, shift ? 3 : 2, dout);
            o[shift].resize(32 * 32);
            hipMemcpy(o[shift].data(), dout, 32 * 32 * 4, hipMemcpyDeviceToHost);
        }
        for (int i = 0; i < 32 * 32; ++i) {
            ++total;
            if (o[0][i] != o[1][i]) { ++diff; const double rel = fabs((double)o[0][i] - o[1][i]) / (fabs((double)o[0][i]) + 1e-30); if (rel > maxrel) maxrel = rel; }
        }
    }
    printf("P.V results of a 32-key sentence at window slot 0 against slot 8: %lld of %lld differ in at least one bit (largest relative difference %.3g)\n", diff, total, maxrel);
    return 0;
}
