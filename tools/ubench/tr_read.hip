// tr_read.hip — what does ds_read_b64_tr_b16 return?  LDS holds the 16-bit value i at element i; every lane supplies its own
// 8-byte-aligned address and the four 16-bit results of every lane are printed as element indices.  Two address patterns:
//   A: lane l reads at element 4 l (the 64 lanes cover 256 consecutive elements)
//   B: the pattern attention.hip uses for a V tile stored as [16-column sub-tile][key][16 columns] (32-byte rows): see below
// Result (MI355X): inside each group of 16 lanes, lane i receives as element j the element (i % 4) of what lane 4 j + i / 4 of the group
// addressed — i.e. column i of the [4 rows][16 columns] block the group's 16 x 8 bytes cover when lane s addresses row s / 4, columns
// 4 (s % 4) ..; there is no lane offset inside the instruction.  (attention.hip's V staging built on it: profiles/r6_experiments.txt §4.)
//                                                                                   usage: tr_read
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

int main() {
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
    return 0;
}
