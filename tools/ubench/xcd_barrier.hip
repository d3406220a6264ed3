// xcd_barrier.hip — a barrier among the workgroups of ONE XCD (32 CUs share one L2) against the device-wide barrier of
// grid_barrier.hip and against a kernel boundary: would the latency route (33 launches per sentence) pay as one persistent
// launch confined to an XCD?  256 workgroups start (160 KiB of LDS each: one per CU); those that find themselves on XCD 0
// (HW_REG_XCC_ID) form the team, the others leave.  Barrier = every member stores its epoch into its slot of ONE 128-byte
// line (after vmcnt(0): its data is in L2), one wave polls the line with device-scope loads (L1 bypassed, L2 hit).
// usage: xcd_barrier [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned load_l2(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool team_barrier(unsigned *flags, int rank, int team, unsigned epoch) {
    __shared__ int ok_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(flags + rank, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = false;
        for (int spins = 0; spins < (1 << 20); ++spins) {
            const unsigned f = threadIdx.x < team ? load_l2(flags + threadIdx.x) : epoch;
            if (__all(f >= epoch)) { ok = true; break; }
        }
        if (threadIdx.x == 0) ok_s = ok;
    }
    __syncthreads();
    return ok_s;
}

__global__ __launch_bounds__(256) void persistent(float *buf, unsigned *flags, unsigned *team_counter, int iters, int payload, int *bad, int *team_seen) {
    extern __shared__ char lds[];
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;
    if (xcc != 0) return;
    __shared__ int rank_s;
    if (threadIdx.x == 0) rank_s = (int)atomicAdd(team_counter, 1u);
    __syncthreads();
    const int rank = rank_s, team = 32;
    if (rank >= team) { if (threadIdx.x == 0) atomicAdd(bad, 1000); return; }
    if (threadIdx.x == 0) atomicAdd(team_seen, 1);
    const int partner = (rank + 5) % team;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (payload) buf[(size_t)((it & 1) * team + rank) * 256 + threadIdx.x] = (float)(it + rank);
        if (!team_barrier(flags, rank, team, (unsigned)it + 1)) { if (threadIdx.x == 0) atomicAdd(bad, 1); return; }
        if (payload) {
            const float v = __builtin_bit_cast(float, load_l2((const unsigned *)buf + (size_t)((it & 1) * team + partner) * 256 + threadIdx.x));
            if (v != (float)(it + partner)) atomicAdd(bad, 1);
            acc += v;
        }
    }
    if (acc == 12345.678f) buf[0] = acc;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    float *buf; unsigned *flags, *team_counter; int *bad, *seen;
    (void)hipMalloc(&buf, (size_t)2 * 32 * 256 * 4); (void)hipMalloc(&flags, 128); (void)hipMalloc(&team_counter, 4); (void)hipMalloc(&bad, 4); (void)hipMalloc(&seen, 4);
    (void)hipFuncSetAttribute((const void *)persistent, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int payload = 0; payload < 2; ++payload) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(flags, 0, 128); (void)hipMemset(team_counter, 0, 4); (void)hipMemset(bad, 0, 4); (void)hipMemset(seen, 0, 4);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            persistent<<<256, 256, 150 * 1024>>>(buf, flags, team_counter, rep ? iters : 10, payload, bad, seen);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            int nbad, nseen; (void)hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&nseen, seen, 4, hipMemcpyDeviceToHost);
            if (rep) printf("one launch, the %d workgroups of XCD 0, flag-line barrier%s: %8.2f us per step (errors: %d)\n", nseen, payload ? " + 1 KiB hand-over" : " only", ms * 1e3 / iters, nbad);
        }
    }
    return 0;
}
