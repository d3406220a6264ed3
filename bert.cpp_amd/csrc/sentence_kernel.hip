// sentence_kernel.hip — the latency route in ONE launch: all encoder layers of a call of at most 128 tokens (the reference's
// real callers: bert_encode per request, reference bert.cpp:943-950, examples/server.cpp:98-114) on the 32 CUs of ONE XCD.
//
// skinny.hip runs the route as 5 launches per layer.  Every launch starts cold (weight block from the memory-side cache: one
// round trip; token fragments: L2 was written back and invalidated at the boundary) and a boundary costs 2.9 us
// (tools/ubench/grid_barrier.hip) — 33 x ~7 us.  A device-wide barrier inside one launch is no cheaper (13.6 us for 192
// workgroups: the arrivals cross the fabric one by one).  But the workgroups of ONE XCD share an L2: a barrier among its 32
// CUs — every member stores its epoch into its slot of one 128-byte line, one wave polls the line with device-scope loads that
// hit L2 — costs 1.05 us, 1.28 with a hand-over (tools/ubench/xcd_barrier.hip), the activations never leave that L2, and the
// next phase's first weight block is requested BEFORE the barrier (weights do not depend on it).  32 CUs are enough: a
// 128-token call is 5.6 GFLOP.
//
// 256 workgroups start (more than half a CU's LDS each: one per CU); those that find themselves on XCD 0 (HW_REG_XCC_ID)
// take a rank from a counter and form the team, the others leave at once.  Phases per layer, a team barrier behind each:
//   QKV projection (a workgroup per 32-feature tile, a wave per 32-token block sharing the tile's weight rows in LDS)
//   -> attention (a workgroup per sentence and head: attention_core.h) -> out-projection -> LayerNorm 1 (a workgroup per token
//   block) -> up-projection + GELU -> down-projection -> LayerNorm 2
// SAME BITS as skinny.hip's launches and so as the batch route: the mat-mul waves are skinny_tile.h's, the attention is
// attention_core.h's, the LayerNorms tile_stream.h's layernorm_runs_of — the source those kernels are made of.  (The
// LayerNorms are phases of their own here: fused into the following projection they are recomputed per feature tile, which
// 192 workgroups absorb and 32 do not.  LayerNorm 1 leaves its rows twice: in feature order — the residual of the
// down-projection — and in the up-projection's fragment order, w16p's.)
#include "attention_core.h"
#include "skinny_tile.h"

#include <type_traits>

#ifndef SK1_FENCE
#define SK1_FENCE 0
#endif
#ifndef SK1_POLL_SLEEP
#define SK1_POLL_SLEEP 0
#endif

namespace bert_hip {

namespace {

constexpr int SK1_TEAM = 32;                   // CUs of an XCD
constexpr int SK1_Z1 = 96 * 1024, SK1_Z2 = 120 * 1024, SK1_LDS = 148 * 1024;   // LDS zones: [0, 96 K) down-projection block /
                                               // attention / LayerNorm rows; two 24 KiB blocks for the K = H mat-muls
constexpr int SK1_MAX_LAYERS = 16;

struct SentenceLayer {
    const half_t *Wqkv, *Wo, *W1, *W2;        // w16, w16, w16p, w16p
    const float *bqkv, *bo, *g1, *be1, *b1, *b2, *g2, *be2;
};

struct SentenceArgs {
    SentenceLayer L[SK1_MAX_LAYERS];
    int n_layer, n_sentences, n_token_blocks, n_head, H, I;
    const int32_t *cu;
    half_t *x, *qkv, *ctx, *y, *ff;            // (ctx doubles as LayerNorm 1's rows in fragment order: free behind the out-projection)
    float *v32;
    unsigned *flags, *ranks;                   // the barrier's line (32 epochs, never reset: epochs grow from launch to launch); the rank counter
    unsigned epoch0, rank0;
    int xcd;                                   // the XCD whose workgroups form the team
    unsigned long long *timeline;              // tuning: [32 ranks][128] wall-clock stamps (100 MHz) around the phases, or null
    int *status, *host_flag;                   // failure: status word |= 4 / 8 (device), *host_flag = the same (mapped host word, or null)
};

__device__ __forceinline__ unsigned sk1_load_l2(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace

template <int NT, int D>
__global__ __launch_bounds__(256) void sentence_kernel(SentenceArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int shared_word;
    if ((int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u) != p.xcd) return;   // HW_REG_XCC_ID: not on the team's XCD
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    if (threadIdx.x == 0) shared_word = (int)(atomicAdd(p.ranks, 1u) - p.rank0);
    __syncthreads();
    const int rank = shared_word;
    __syncthreads();
    if (rank < 0 || rank >= SK1_TEAM) {        // (an XCD that was given more workgroups than CUs: not this part)
        if (threadIdx.x == 0) {
            atomicOr(p.status, 4);
            if (p.host_flag) __hip_atomic_store(p.host_flag, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    constexpr int H = 128 * NT;
    const int tb = p.n_token_blocks, I = p.I;
    unsigned epoch = p.epoch0;
    int n_stamps = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if (p.timeline && threadIdx.x == 0 && n_stamps < 128) p.timeline[rank * 128 + n_stamps++] = wall_clock64();
    };
    stamp();

    auto landed = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the block (and its first fragments) have landed
        __builtin_amdgcn_s_barrier();
    };
    // team barrier: wave 0 announces and polls; the other waves request what the next phase needs that does not depend on the
    // barrier (vmcnt retires in order: a poll behind a 96 KiB request would see its answer when the block has landed)
    auto team_sync = [&](auto prefetch) __attribute__((always_inline)) -> bool {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's stores are in L2
        __syncthreads();
        ++epoch;
        if (wave == 0) {
            if (lane == 0) __hip_atomic_store(p.flags + rank, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = false;
            for (int spins = 0; spins < (1 << 20); ++spins) {                 // (never hang the GPU: give up after ~a second)
                const unsigned f = lane < SK1_TEAM ? sk1_load_l2(p.flags + lane) : epoch;
                if (__all((int)(f - epoch) >= 0)) { ok = true; break; }
#if SK1_POLL_SLEEP
                __builtin_amdgcn_s_sleep(SK1_POLL_SLEEP);
#endif
            }
            if (lane == 0) shared_word = ok;
        } else {
            prefetch();
        }
        __syncthreads();
        const bool ok = shared_word != 0;
#if SK1_FENCE == 0
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (L1 lines of the rows other CUs rewrote)
#elif SK1_FENCE == 1
        asm volatile("buffer_inv sc0" ::: "memory");
#endif
        if (!ok && threadIdx.x == 0) {
            atomicOr(p.status, 8);
            if (p.host_flag) __hip_atomic_store(p.host_flag, 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        stamp();
        return ok;
    };
    auto none = []() {};
    // the weight block of tile `t` of W -> zone z, by the waves first .. 3
    auto request = [&](const half_t *W, int N, int K, int t, char *z, int first) __attribute__((always_inline)) {
        if (wave >= first && t < N / 32) skinny_request_weights(W + (size_t)t * 32 * K, K, z, wave - first, 4 - first, lane);
    };
    // a mat-mul phase: tile rank (block already requested into z0 if prefetched) and tile rank + 32 (zone z1)
    auto matmul = [&](auto mode_tag, const SkinnyArgs &a, char *z0, char *z1, bool prefetched) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_tag)::value;
        const int n_tiles = a.N / 32;
        if (!prefetched) request(a.W, a.N, a.K, rank, z0, 0);
        request(a.W, a.N, a.K, rank + SK1_TEAM, z1, 0);
        for (int t = rank, i = 0; t < n_tiles; t += SK1_TEAM, ++i) {
            if (wave < tb) skinny_wave<MODE>(a, i ? z1 : z0, t * 32, wave * 32 + l31, lane, landed);
            else landed();
        }
    };
    // a LayerNorm phase: token block `rank`: the rows -> LDS (run (n, g) of all lanes = one 1 KiB piece), wave 0 normalises
    auto layernorm = [&](auto pair_tag, const float *v, const float *gamma, const float *beta, half_t *out, half_t *out_frag) __attribute__((always_inline)) {
        constexpr bool PAIR = decltype(pair_tag)::value;
        if (rank >= tb) return;
        const int tok = rank * 32 + l31;
        const char *vrow = (const char *)(v + (size_t)tok * H + 4 * hi);
        for (int pc = wave; pc < 16 * NT; pc += 4)
            __builtin_amdgcn_global_load_lds(AS_GLOBAL(vrow + (pc >> 2) * 128 + (pc & 3) * 32), AS_LDS(smem + pc * 1024), 16, 0, 0);
        if (wave != 0) { landed(); return; }
        f16x4 y[4 * NT][4];
        layernorm_runs_of<PAIR, NT>([&](int n, int g) __attribute__((always_inline)) { return *(const f32x4 *)(smem + (n * 4 + g) * 1024 + lane * 16); },
                                    landed, gamma, beta, hi, y);
        half_t *orow = out + (size_t)tok * H;
#pragma unroll
        for (int n = 0; n < 4 * NT; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *(f16x4 *)(orow + 32 * n + 8 * g + 4 * hi) = y[n][g];
                // fragment order (w16p's): run 8 (g & 1) + 4 hi of the 16-group g >> 1 at position 8 hi + 4 (g & 1)
                if (out_frag) *(f16x4 *)(out_frag + (size_t)tok * H + 32 * n + 16 * (g >> 1) + 8 * hi + 4 * (g & 1)) = y[n][g];
            }
    };

    char *const Z0 = smem, *const Z1 = smem + SK1_Z1, *const Z2 = smem + SK1_Z2;
    if (p.n_layer == 0) { (void)team_sync(none); return; }    // (the probe at load time: does the team form, does a barrier pass)
    request(p.L[0].Wqkv, 3 * H, H, rank, Z1, 0);
    for (int il = 0; il < p.n_layer; ++il) {
        const SentenceLayer &L = p.L[il];
        SkinnyArgs a;
        a.V = nullptr; a.gamma = a.beta = nullptr; a.ln_out = nullptr;
        // ---- QKV projection (its first blocks were requested before the barrier)
        a.W = L.Wqkv; a.A = p.x; a.bias = L.bqkv; a.resid = nullptr; a.out16 = p.qkv; a.out32 = nullptr; a.N = 3 * H; a.K = H;
        matmul(std::integral_constant<int, SK_QKV>{}, a, Z1, Z2, true);
        stamp();
        if (!team_sync([&]() __attribute__((always_inline)) { request(L.Wo, H, H, rank, Z2, 1); })) return;
        // ---- attention
        for (int t = rank; t < p.n_sentences * p.n_head; t += SK1_TEAM) {
            attention_head<D, 256, 128>(p.qkv, p.cu, p.n_head, p.ctx, Z0, t / p.n_head, t % p.n_head);
            __syncthreads();
        }
        stamp();
        if (!team_sync(none)) return;
        // ---- out-projection: v32 = x + bo + ctx Wo^T
        a.W = L.Wo; a.A = p.ctx; a.bias = L.bo; a.resid = p.x; a.out16 = nullptr; a.out32 = p.v32; a.N = H; a.K = H;
        matmul(std::integral_constant<int, SK_PROJ>{}, a, Z2, Z1, true);
        stamp();
        if (!team_sync([&]() __attribute__((always_inline)) { request(L.W1, I, H, rank, Z1, 1); })) return;
        // ---- LayerNorm 1 -> y, and y in fragment order (in ctx's place)
        layernorm(std::true_type{}, p.v32, L.g1, L.be1, p.y, p.ctx);
        stamp();
        if (!team_sync(none)) return;
        // ---- up-projection + GELU
        a.W = L.W1; a.A = p.ctx; a.bias = L.b1; a.resid = nullptr; a.out16 = p.ff; a.out32 = nullptr; a.N = I; a.K = H;
        matmul(std::integral_constant<int, SK_UP>{}, a, Z1, Z2, true);
        stamp();
        if (!team_sync([&]() __attribute__((always_inline)) { request(L.W2, H, I, rank, Z0, 1); })) return;
        // ---- down-projection: v32 = b2 + y + ff W2^T
        a.W = L.W2; a.A = p.ff; a.bias = L.b2; a.resid = p.y; a.out16 = nullptr; a.out32 = p.v32; a.N = H; a.K = I;
        matmul(std::integral_constant<int, SK_DOWN>{}, a, Z0, Z1, true);
        const bool more = il + 1 < p.n_layer;
        stamp();
        if (!team_sync([&]() __attribute__((always_inline)) {
                if (more) {
                    request(p.L[il + 1].Wqkv, 3 * H, H, rank, Z1, 1);
                    // (the second tile's block stays with the phase itself: LayerNorm 2's rows are in the way of nothing, but
                    // Z2 is; requested there it would have to wait for this barrier anyway)
                }
            })) return;
        // ---- LayerNorm 2 -> x
        layernorm(std::false_type{}, p.v32, L.g2, L.be2, p.x, nullptr);
        stamp();
        if (more && !team_sync(none)) return;
    }
}

bool sentence_kernel_supported(const GemmWeight &Wqkv, const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2, int n_layer, int n_head,
                               int d_head) {
    const int H = W1.K, I = W1.N;
    return skinny_layer_supported(Wqkv, Wo, W1, W2) && n_layer <= SK1_MAX_LAYERS && (d_head == 32 || d_head == 64) && n_head * d_head == H &&
           3 * H / 32 <= 2 * SK1_TEAM && I / 32 <= 2 * SK1_TEAM && (size_t)32 * I * 2 <= (size_t)SK1_Z1 &&
           (size_t)32 * H * 2 <= (size_t)(SK1_Z2 - SK1_Z1);
}

int sentence_kernel_barriers(int n_layer) { return n_layer ? 7 * n_layer - 1 : 1; }

void launch_sentence_kernel(const ModelLayerWeights *layers, int n_layer, half_t *x, half_t *qkv, half_t *ctx, half_t *y, half_t *ff, float *v32,
                            const int32_t *cu_seqlens, int n_sentences, int T, int n_head, unsigned *flags, unsigned *ranks, unsigned epoch0,
                            unsigned rank0, int xcd, int *status, int *host_flag, hipStream_t stream, int H_probe, unsigned long long *timeline) {
    SentenceArgs a;
    for (int il = 0; il < n_layer; ++il) {
        const ModelLayerWeights &m = layers[il];
        a.L[il] = {m.Wqkv->w16, m.Wo->w16, m.W1->w16p, m.W2->w16p, m.bqkv, m.bo, m.g1, m.be1, m.b1, m.b2, m.g2, m.be2};
    }
    const int H = n_layer ? layers[0].W1->K : H_probe, d_head = H / n_head;            // (n_layer = 0: the probe, no weights)
    a.n_layer = n_layer; a.n_sentences = n_sentences; a.n_token_blocks = (T + 31) / 32; a.n_head = n_head; a.H = H; a.I = n_layer ? layers[0].W1->N : 0;
    a.cu = cu_seqlens; a.x = x; a.qkv = qkv; a.ctx = ctx; a.y = y; a.ff = ff; a.v32 = v32;
    a.flags = flags; a.ranks = ranks; a.epoch0 = epoch0; a.rank0 = rank0; a.xcd = xcd; a.status = status; a.host_flag = host_flag; a.timeline = timeline;
    static DeviceFlags configured[4];
    auto go = [&](auto kernel, int m) {
        configure_once(configured[m], [&] { (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SK1_LDS); });   // (+ 4 bytes of static LDS)
        BERT_LAUNCH(kernel, dim3(8 * SK1_TEAM), dim3(256), SK1_LDS, stream, a);
    };
    if (H == 256) { if (d_head == 32) go(sentence_kernel<2, 32>, 0); else go(sentence_kernel<2, 64>, 1); }
    else { if (d_head == 32) go(sentence_kernel<3, 32>, 2); else go(sentence_kernel<3, 64>, 3); }
}

}  // namespace bert_hip
