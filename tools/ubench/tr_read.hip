// tr_read.hip — what does ds_read_b64_tr_b16 return?  LDS holds the 16-bit value i at element i; every lane supplies its own
// 8-byte-aligned address and the four 16-bit results of every lane are printed as element indices.  Two address patterns:
//   A: lane l reads at element 4 l (the 64 lanes cover 256 consecutive elements)
//   B: the pattern attention.hip uses for a V tile stored as [16-column sub-tile][key][16 columns] (32-byte rows): see below
// and the cycles of 64 back-to-back reads per wave (8 waves) for pattern B with a sub-tile stride of `stride` bytes.
//                                                                                   usage: tr_read [stride_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned short u16;

__global__ void probe(const int *addr_elems, u16 *out) {
    __shared__ u16 lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (u16)i;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) u16 *)lds + 2u * (unsigned)addr_elems[threadIdx.x];
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = (u16)(r.x & 0xffff); out[threadIdx.x * 4 + 1] = (u16)(r.x >> 16);
    out[threadIdx.x * 4 + 2] = (u16)(r.y & 0xffff); out[threadIdx.x * 4 + 3] = (u16)(r.y >> 16);
}

// timing: every wave issues `n` reads of pattern B (or the plain ds_read_b64 of the same addresses) back to back
template <bool TR>
__global__ __launch_bounds__(512) void rate(int stride_bytes, int n, long long *cycles, unsigned *sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    // group g: sub-tile (g & 1), keys 4 (g >> 1) .. + 3; lane i of the group: key row i >> 2, columns 4 (i & 3) ..
    const unsigned a = base + (unsigned)((g & 1) * stride_bytes + (4 * (g >> 1) + (i >> 2)) * 32 + (i & 3) * 8);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    unsigned acc = 0;
    for (int k = 0; k < n; ++k) {
        uint2 r;
        const unsigned ak = a + (unsigned)((k & 31) * 256);
        if (TR) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(ak) : "memory");
        else asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(ak) : "memory");
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        acc ^= r.x ^ r.y;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 0x12345) sink[0] = acc;
}

int main(int argc, char **argv) {
    int *da; u16 *dout;
    hipMalloc(&da, 64 * 4); hipMalloc(&dout, 64 * 4 * 2);
    for (int pat = 0; pat < 2; ++pat) {
        int h[64];
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) h[l] = 4 * l;
            else { const int g = l >> 4, i = l & 15; h[l] = (g & 1) * 2048 + (4 * (g >> 1) + (i >> 2)) * 16 + (i & 3) * 4; }
        }
        hipMemcpy(da, h, sizeof(h), hipMemcpyHostToDevice);
        probe<<<1, 64>>>(da, dout);
        u16 o[256];
        hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
        printf("pattern %c (lane: address element -> the four results as element indices)\n", 'A' + pat);
        for (int l = 0; l < 64; ++l) printf("  lane %2d @%4d -> %4d %4d %4d %4d%s", l, h[l], o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3], (l & 3) == 3 ? "\n" : "");
    }
    long long *dc; unsigned *ds;
    hipMalloc(&dc, 8); hipMalloc(&ds, 4);
    for (int stride : {argc > 1 ? atoi(argv[1]) : 16384, 16384 + 128, 16384 + 256, 16384 + 64}) {
        for (int tr = 0; tr < 2; ++tr) {
            long long c = 0;
            for (int rep = 0; rep < 3; ++rep) {
                if (tr) rate<true><<<1, 512, 65536>>>(stride, 4096, dc, ds); else rate<false><<<1, 512, 65536>>>(stride, 4096, dc, ds);
                hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
            }
            printf("sub-tile stride %6d bytes, %s: %.1f cycles per read and wave (8 waves, 4096 reads each)\n", stride, tr ? "ds_read_b64_tr_b16" : "ds_read_b64       ", c / 4096.0);
        }
    }
    return 0;
}
