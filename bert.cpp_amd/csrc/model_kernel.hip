// model_kernel.hip — ONE launch for all encoder layers of a batch (f16 images, H = 256 / 384, d_head 32, sentences of at most
// 128 tokens) plus the pooling: reference bert.cpp:816-913, the whole loop over layers and what follows it.  Two forms: FULL
// windows (every sentence exactly 128 tokens: a window is a sentence is a 128-token block) and RAGGED ones (whole sentences
// with at most 128 tokens between them: the layer-tail phase runs on the window's rows of the packed batch).
//
// Why: with one kernel per layer half (qkv_attention2, layer_tail) a 256-sentence step is ONE workgroup per CU per launch,
// all 256 in lockstep — everybody's load burst (ctx + x: 48 MB at the HBM limit), everybody's compute, everybody's store
// burst, twelve times per forward pass; two free-running half-batch lanes already measure +4-6 % (tools/dual_lane_probe.py),
// but a call has to join its lanes.  A window's tokens depend on no other window's: here a workgroup carries ITS window
// through every layer — window-kernel phase (Q|K|V projection + attention), layer-tail phase (out-projection + LN + FFN + LN),
// the two kernels' bodies unchanged (same arithmetic, same bits: tested) — and nothing ever puts the workgroups back in step.
// The hand-overs (ctx from the attention waves to the tail, x from the tail to the next layer's projection waves) are plain
// global stores and loads of the SAME workgroup: 96 KiB each, behind a workgroup-scope fence (one L1 per CU, shared by the
// workgroup's waves: no invalidate needed outside threadgroup-split mode).  They do NOT stay inside the XCD's L2: 256
// workgroups x 192 KiB = 48 MiB per layer against 32 MiB of L2, and every store leaves the L2 towards the fabric anyway —
// measured 313 MB written and 1.06 GB of fabric traffic per launch (profiles/r3_pmc.txt), absorbed by the Infinity Cache at
// about 1.3 TB/s: far from a limit, but not free.  What the single launch removes is the lockstep, not the bytes.
// Full form only: the kernel trusts n_tokens = 128 n_sentences to mean "every sentence is exactly 128 tokens"; that holds
// whenever the caller's max_len promise (<= 128) does.  A batch that breaks it is flagged by the pooling guard (status word),
// and the rows of sentences sharing a 128-token block with the offender are not meaningful (include/bert_hip.h).
//
// Both bodies keep their own LDS layout (160 KiB each, used one after the other) and register budget (256 per wave).  This
// translation unit is compiled with the flags of both (Makefile): -fno-slp-vectorize (qkv_attention2) and
// -structurizecfg-skip-uniform-regions (layer_tail).
#define BERT_HIP_PHASES_ONLY
#include "qkv_attention2.hip"
#include "layer_tail.hip"
#undef BERT_HIP_PHASES_ONLY
#include "pool_normalize.h"

namespace bert_hip {

#ifdef BERT_HIP_MODEL_TIMELINE      // (tuning aid: tools/variant.sh tl model_kernel.hip "-DBERT_HIP_MODEL_TIMELINE -fno-slp-vectorize")
// phase boundaries of every workgroup on the constant 100 MHz counter: [0] start, [1 + 2 l] window phase of layer l done, [2 + 2 l] its tail
// phase done, [1 + 2 L] pooled
static __device__ unsigned long long g_tl_model[1024 * 32];
#define MK_STAMP(i) do { const int mk_i = (i); if (threadIdx.x == 0 && mk_i < 32) g_tl_model[(blockIdx.x & 1023) * 32 + mk_i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MK_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int MODEL_MAX_LAYERS = 12;

struct ModelLayerArgs {              // what differs from layer to layer
    const half_t *wqkv, *wo, *w1p, *w2p;
    const float *bqkv, *bo, *g1, *be1, *b1, *b2, *g2, *be2;
};
struct ModelArgs {
    half_t *x, *ctx;                 // [T][H] hidden state (in: embeddings + LN; out: the last layer's output), attention context
    const int32_t *cu;
    const int2 *groups;              // RAGGED: per window {first sentence, count} (qkv_attention2.hip), or nullptr: one sentence per window
    const int *n_groups;             // RAGGED: device word holding the number of windows (the grid is an upper bound), or nullptr
    int n_layer, n_head, n_sent, I, slot_mask;      // slot_mask: the windows' place granularity - 1 (Qkv2Args)
    float *pooled;                   // [n_sent][H] f32: the sentences' pooled, normalised rows (the workgroup pools its window itself), or nullptr
    int *status;                     // pooling's status word (a sentence outside [1, max_len])
    int max_len;
    ModelLayerArgs layer[MODEL_MAX_LAYERS];
};

}  // namespace

// RAGGED: windows of whole sentences with up to 128 tokens between them (the window phase's slots: every sentence starts at a
// multiple of 16), whose rows of x / ctx are tok0 .. tok0 + rows - 1 of the packed batch: the layer-tail phase runs on those.
template <int NT, bool RAGGED>
__global__ __launch_bounds__(512, 2) void model_kernel(ModelArgs m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int window = (int)blockIdx.x;              // full windows: = sentence = 128-token block
    int tok0 = window * 128, rows = 128, first = window, count = 1;
    if constexpr (RAGGED) {
        if (m.groups) {
            if (m.n_groups && window >= *m.n_groups) return;
            const int2 g = m.groups[window];
            first = g.x; count = g.y;
        }
        if (count <= 0 || first >= m.n_sent) return;
        tok0 = m.cu[first];
        // (a sentence longer than a window breaks the caller's promise: its rows behind the 128th are not computed — it is the
        // one that gets a NaN row from the length guard of the pooling kernel)
        rows = min(m.cu[first + count] - tok0, 128);
    }
    // The thread id is REBUILT per phase from the wave's index (a scalar register) and the lane count: with threadIdx.x itself
    // the compiler hoists both phases' lane-dependent address arithmetic out of the layer loop — eighty registers' worth —
    // and spills it; even one vector register alive across the phases is one more than the layer tail has (a scratch reload
    // among its hand-counted LDS-DMA pieces would break their vmcnt arithmetic).
    const int wave_index = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    auto thread_id = [&]() __attribute__((always_inline)) {
        int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), w = wave_index;
        asm volatile("" : "+v"(ln), "+s"(w));
        return w * 64 + ln;
    };
    MK_STAMP(0);
    const int n_layer = rows > 0 ? m.n_layer : 0;    // (a window of empty sentences: nothing to compute, NaN rows from the pooling below)
    for (int l = 0; l < n_layer; ++l) {
        const ModelLayerArgs &L = m.layer[l];
        int tid = thread_id();
        {
            Qkv2Args q;
            q.x = m.x; q.w = L.wqkv; q.qs = nullptr; q.sc = nullptr; q.bias = L.bqkv; q.cu = m.cu; q.groups = RAGGED ? m.groups : nullptr; q.n_groups = nullptr;
            q.out = m.ctx; q.n_head = m.n_head; q.n_sent = m.n_sent; q.spw = 1; q.slot_mask = m.slot_mask;
            qkv_attention2_body<2 * NT, NT, GW_F16>(q, smem, window, tid);
        }
        // ctx is written (attention waves), every LDS access of the phase has returned: hand over to the tail
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        MK_STAMP(1 + 2 * l);
        tid = thread_id();
        {
            TailArgs t;
            t.ctx = m.ctx; t.x = m.x; t.wo = L.wo; t.w1p = L.w1p; t.w2p = L.w2p;
            t.wo_qs = t.w1_qs = t.w2_qs = nullptr; t.wo_sc = t.w1_sc = t.w2_sc = nullptr;
            t.bo = L.bo; t.g1 = L.g1; t.be1 = L.be1; t.b1 = L.b1; t.b2 = L.b2; t.g2 = L.g2; t.be2 = L.be2; t.out = m.x; t.I = m.I;
            layer_tail_body<NT, GW_F16, RAGGED>(t, smem, tok0, rows, tid);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        MK_STAMP(2 + 2 * l);
    }
    // ---- mean-pool + L2 normalise of the window's sentences (pool_normalize.h: the pooling kernel's body and bits), while the
    // other workgroups are still in their layers: the rows were written by this workgroup and sit in the L2
    if (m.pooled) {
        const int tid = thread_id();
        for (int j = 0; j < count; ++j) {
            const int b = first + j, t0 = m.cu[b];
            int n = m.cu[b + 1] - t0;
            // (the full-window form computed rows 128 b .. 128 b + 127 as ONE sentence: in a batch that has this shape only by the
            // sum of its lengths — somebody longer than max_len = 128, somebody shorter — a sentence that is not exactly that block
            // gets the NaN row and the status word of the length guard instead of numbers made of its neighbours' tokens)
            if (!RAGGED && (t0 != tok0 || n != 128)) n = -1;
            pool_normalize_sentence(m.x, t0, n, b, 128 * NT, m.max_len, m.status, m.pooled, (float *)smem, tid, tid < 256);
            __syncthreads();                                  // (the next sentence reuses the partial rows)
        }
    }
    MK_STAMP(1 + 2 * n_layer);
}

// full: every sentence exactly 128 tokens (T = 128 B: the specialised form); otherwise windows of whole sentences — the caller's
// list (`groups`), or one sentence per window when the promised max_len leaves room for no second one
bool model_kernel_supported(const GemmWeight &Wqkv, const GemmWeight &Wo, const GemmWeight &W1, const GemmWeight &W2, int n_layer,
                            int n_head, int d_head, int max_len) {
    const int H = n_head * d_head;
    return n_layer <= MODEL_MAX_LAYERS && (H == 256 || H == 384) && max_len >= 1 && max_len <= 128 &&
           Wqkv.type == GW_F16 && Wo.type == GW_F16 && qkv_attention2_supported(Wqkv, n_head, d_head, max_len) && layer_tail_supported(Wo, W1, W2);
}

void launch_model_kernel(const ModelLayerWeights *layers, int n_layer, half_t *x, half_t *ctx, const int32_t *cu_seqlens, int n_sentences,
                         int n_tokens, const int2 *groups, int n_groups, const int *n_groups_dev, int n_head, float *pooled, int max_len,
                         int *status, int slots, hipStream_t stream) {
    ModelArgs m;
    m.pooled = pooled; m.status = status; m.max_len = max_len;
    m.x = x; m.ctx = ctx; m.cu = cu_seqlens; m.groups = groups; m.n_groups = n_groups_dev;
    m.n_layer = n_layer; m.n_head = n_head; m.n_sent = n_sentences; m.I = layers[0].W1->N; m.slot_mask = slots - 1;
    for (int l = 0; l < n_layer; ++l) {
        const ModelLayerWeights &s = layers[l];
        ModelLayerArgs &d = m.layer[l];
        d.wqkv = s.Wqkv->w16; d.wo = s.Wo->w16; d.w1p = s.W1->w16p; d.w2p = s.W2->w16p;
        d.bqkv = s.bqkv; d.bo = s.bo; d.g1 = s.g1; d.be1 = s.be1; d.b1 = s.b1; d.b2 = s.b2; d.g2 = s.g2; d.be2 = s.be2;
    }
    const int H = layers[0].W1->K, KT = H / 64, GB = KT / 2;
    const size_t lds_q = (size_t)3 * GB * Q2_TILE + 2 * (2 * Q2_WIN * 64 + 32 * Q2_VT_LD * 2) + (size_t)2 * H * sizeof(float);
    const size_t lds = std::max(lds_q, layer_tail_lds(H, m.I));
    const bool full = (long long)n_sentences * 128 == n_tokens;      // (with no sentence over 128 tokens: every window is one whole sentence)
    const int grid = full || !groups ? n_sentences : n_groups;
    static DeviceFlags configured[4];
    auto go = [&](auto kernel, int which) {
        configure_once(configured[which], [&] { (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        BERT_LAUNCH(kernel, dim3(grid), dim3(512), lds, stream, m);
#ifdef BERT_HIP_MODEL_TIMELINE
        static int shots = 0;
        if (grid >= 256 && shots++ == 30) {
            (void)hipDeviceSynchronize();
            static unsigned long long h[1024 * 32];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tl_model), sizeof(h));
            const int nw = grid < 1024 ? grid : 1024, ns = 2 + 2 * n_layer;
            unsigned long long t0 = ~0ull;
            for (int w = 0; w < nw; ++w) t0 = h[w * 32] < t0 ? h[w * 32] : t0;
            // per stamp: min / mean / max over the workgroups of (stamp - earliest start), in microseconds (100 MHz counter)
            fprintf(stderr, "model_kernel timeline, %d workgroups (us after the first workgroup's start: min mean max):\n", nw);
            for (int i = 0; i < ns; ++i) {
                double lo = 1e30, hi = 0, sum = 0;
                for (int w = 0; w < nw; ++w) { const double v = (double)(h[w * 32 + i] - t0) * 0.01; lo = v < lo ? v : lo; hi = v > hi ? v : hi; sum += v; }
                fprintf(stderr, "  stamp %2d: %8.2f %8.2f %8.2f\n", i, lo, sum / nw, hi);
            }
            // per phase: mean duration over workgroups
            fprintf(stderr, "  mean phase durations (window, tail) per layer:");
            for (int l = 0; l < n_layer; ++l) {
                double a = 0, b = 0;
                for (int w = 0; w < nw; ++w) { a += (double)(h[w * 32 + 1 + 2 * l] - h[w * 32 + 2 * l]) * 0.01; b += (double)(h[w * 32 + 2 + 2 * l] - h[w * 32 + 1 + 2 * l]) * 0.01; }
                fprintf(stderr, " (%.1f, %.1f)", a / nw, b / nw);
            }
            fprintf(stderr, "\n  mean end per XCD:");
            for (int x = 0; x < 8; ++x) {
                double e = 0; int c = 0;
                for (int w = x; w < nw; w += 8) { e += (double)(h[w * 32 + ns - 1] - t0) * 0.01; ++c; }
                fprintf(stderr, " %.1f", e / c);
            }
            fprintf(stderr, "\n  mean end per eighth of the grid:");
            for (int x = 0; x < 8; ++x) {
                double e = 0; int c = 0;
                for (int w = x * nw / 8; w < (x + 1) * nw / 8; ++w) { e += (double)(h[w * 32 + ns - 1] - t0) * 0.01; ++c; }
                fprintf(stderr, " %.1f", e / c);
            }
            double pl = 0;
            for (int w = 0; w < nw; ++w) pl += (double)(h[w * 32 + 1 + 2 * n_layer] - h[w * 32 + 2 * n_layer]) * 0.01;
            fprintf(stderr, "; pooling %.2f\n", pl / nw);
        }
#endif
    };
    if (full) { if (H == 256) go(model_kernel<2, false>, 0); else go(model_kernel<3, false>, 1); }
    else { if (H == 256) go(model_kernel<2, true>, 2); else go(model_kernel<3, true>, 3); }
}

}  // namespace bert_hip
