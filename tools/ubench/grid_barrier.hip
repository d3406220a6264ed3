// grid_barrier.hip — what a device-wide barrier inside one launch costs on an MI355X, against a kernel boundary: G workgroups of 256
// threads each write 1 KiB, meet at a barrier (agent-scope release, atomic arrival counter, bounded spin, agent-scope acquire) and
// read the 1 KiB a workgroup of ANOTHER XCD wrote; the same hand-over as a chain of launches.  Decides whether the latency route
// (skinny.hip: 33 launches per call) should become one persistent launch.  usage: grid_barrier [G]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1 << 22)) { ok = false; break; }      // (never hang the GPU: give up after ~a second)
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}

__global__ __launch_bounds__(256) void persistent(float *buf, unsigned *counter, int iters, int payload, int *bad) {
    const int G = gridDim.x, w = blockIdx.x, partner = (w + 3) % G;           // (+3: another XCD)
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (payload) buf[(size_t)((it & 1) * G + w) * 256 + threadIdx.x] = (float)(it + w);
        if (!grid_barrier(counter, (unsigned)(it + 1) * G)) { if (threadIdx.x == 0) atomicAdd(bad, 1); return; }
        if (payload) {
            const float v = buf[(size_t)((it & 1) * G + partner) * 256 + threadIdx.x];
            if (v != (float)(it + partner)) atomicAdd(bad, 1);
            acc += v;
        }
    }
    if (acc == 12345.678f) buf[0] = acc;
}

__global__ __launch_bounds__(256) void step_kernel(float *buf, int it, int *bad) {
    const int G = gridDim.x, w = blockIdx.x, partner = (w + 3) % G;
    if (it > 0) {
        const float v = buf[(size_t)(((it - 1) & 1) * G + partner) * 256 + threadIdx.x];
        if (v != (float)(it - 1 + partner)) atomicAdd(bad, 1);
    }
    buf[(size_t)((it & 1) * G + w) * 256 + threadIdx.x] = (float)(it + w);
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 192, iters = 2000;
    float *buf; unsigned *counter; int *bad;
    (void)hipMalloc(&buf, (size_t)2 * G * 256 * 4); hipMalloc(&counter, 4); hipMalloc(&bad, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int payload = 0; payload < 2; ++payload) {
        hipMemset(counter, 0, 4); hipMemset(bad, 0, 4);
        persistent<<<G, 256>>>(buf, counter, 10, payload, bad);
        hipDeviceSynchronize();
        hipMemset(counter, 0, 4);
        hipEventRecord(a);
        persistent<<<G, 256>>>(buf, counter, iters, payload, bad);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        int nbad; hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost);
        printf("one launch, %d workgroups, barrier%s: %8.2f us per step (errors: %d)\n", G, payload ? " + 1 KiB hand-over across XCDs" : " only", ms * 1e3 / iters, nbad);
    }
    hipMemset(bad, 0, 4);
    for (int it = 0; it < 10; ++it) step_kernel<<<G, 256>>>(buf, it, bad);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int it = 0; it < iters; ++it) step_kernel<<<G, 256>>>(buf, it, bad);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    int nbad; hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost);
    printf("a launch per step, %d workgroups, the same hand-over:      %8.2f us per step (errors: %d)\n", G, ms * 1e3 / iters, nbad);
    return 0;
}
