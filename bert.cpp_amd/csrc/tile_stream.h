// tile_stream.h — shared device helpers of the kernels that stream weight tiles through LDS rings (layer_tail.hip,
// qkv_attention2.hip): compile-time loops, hand-issued LDS reads, the tile swizzle, q4 block expansion, and the in-kernel
// timeline stamps of the tuning builds.  128-row x 64-half tiles travel HBM/L2 -> LDS by global_load_lds_dwordx4 several tiles
// ahead of their use and are retired with a counted s_waitcnt vmcnt(N) plus one s_barrier per tile; the 16-byte chunk of
// every row is XOR-swizzled on the SOURCE side (LDS-DMA writes lane-linearly) and un-swizzled by the ds_read_b128 reads.
#pragma once
#include "kernels.h"

#include <cstdio>
#include <type_traits>
#include <utility>

namespace bert_hip {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

#define AS_GLOBAL(p) ((const __attribute__((address_space(1))) void *)(p))
#define AS_LDS(p) ((__attribute__((address_space(3))) void *)(p))


// Tuning aid, compiled in only with -DBERT_HIP_TIMELINE: thread 0 of every workgroup stamps the shader clock
// (TL_STAMP) at the top of each interval; the launcher (TL_DUMP) prints the deltas of a few workgroups of its
// 21st large launch to stderr.  Not part of the product build.
#ifdef BERT_HIP_TIMELINE
static __device__ unsigned long long g_timeline[1024 * 256];
#define TL_STAMP(i) do { const int tl_i = (i); if (threadIdx.x == 0 && tl_i < 256) g_timeline[(blockIdx.x & 1023) * 256 + tl_i] = __builtin_readcyclecounter(); } while (0)
#define TL_STAMP_AT(sel, i) do { const int tl_i = (i); if ((sel) && tl_i < 256) g_timeline[(blockIdx.x & 1023) * 256 + tl_i] = __builtin_readcyclecounter(); } while (0)
// the constant 100 MHz counter next to the shader clock: (cycles between two stamps) / (real time between them) = the clock
#define TL_REALTIME_AT(sel, i) do { const int tl_i = (i); if ((sel) && tl_i < 256) g_timeline[(blockIdx.x & 1023) * 256 + tl_i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TL_DUMP(cond, nstamps) do {                                                                          \
    static int tl_shots = 0;                                                                                  \
    if ((cond) && tl_shots++ == 20) {                                                                         \
        (void)hipDeviceSynchronize();                                                                         \
        static unsigned long long tl_h[1024 * 256];                                                           \
        (void)hipMemcpyFromSymbol(tl_h, HIP_SYMBOL(g_timeline), sizeof(tl_h));                                \
        const int tl_n = (nstamps) < 256 ? (nstamps) : 256;                                                   \
        for (int b : {0, 1, 100, 255}) {                                                                      \
            fprintf(stderr, "timeline wg %3d:", b);                                                           \
            for (int i = 1; i < tl_n; ++i) fprintf(stderr, " %llu", tl_h[b * 256 + i] - tl_h[b * 256 + i - 1]); \
            fprintf(stderr, "  total %llu\n", tl_h[b * 256 + tl_n - 1] - tl_h[b * 256]);                      \
        }                                                                                                     \
    }                                                                                                         \
} while (0)
// raw stamps relative to stamp 0 (kernels whose waves stamp disjoint index ranges)
#define TL_DUMP_RAW(cond, nstamps) do {                                                                      \
    static int tl_shots = 0;                                                                                  \
    if ((cond) && tl_shots++ == 20) {                                                                         \
        (void)hipDeviceSynchronize();                                                                         \
        static unsigned long long tl_h[1024 * 256];                                                           \
        (void)hipMemcpyFromSymbol(tl_h, HIP_SYMBOL(g_timeline), sizeof(tl_h));                                \
        for (int b : {0, 100}) {                                                                              \
            fprintf(stderr, "rawtimeline wg %3d:", b);                                                        \
            for (int i = 0; i < (nstamps); ++i)                                                               \
                fprintf(stderr, " %lld", tl_h[b * 256 + i] ? (long long)(tl_h[b * 256 + i] - tl_h[b * 256]) : -1LL); \
            fprintf(stderr, "\n");                                                                            \
        }                                                                                                     \
    }                                                                                                         \
} while (0)
#else
#define TL_DUMP_RAW(cond, nstamps) do { } while (0)
#define TL_REALTIME_AT(sel, i) do { } while (0)
#define TL_STAMP(i) do { } while (0)
#define TL_STAMP_AT(sel, i) do { } while (0)
#define TL_DUMP(cond, nstamps) do { } while (0)
#endif

// ---- hand-issued LDS reads.  The compiler's wait insertion retires LDS reads with lgkmcnt(0) only; a wave that
// has a matrix pipe to itself must keep reads in flight under its MFMAs, so the self-pipelined kernels issue
// their fragment reads as asm and retire them with partial counts (LDS operations complete in order).  Rules:
// every such read is covered by an explicit wait that names the destination registers ("+v"), and no
// compiler-generated LDS access may sit between a group of reads and its partial wait ("memory" clobbers keep
// them out).
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
}
template <int OFF>
__device__ __forceinline__ f16x8 lds_read_b128_u(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

__device__ __forceinline__ int off64(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// ---- q4 blocks -> f16 tiles.  A thread expands one block of 32 weights (16 bytes of nibbles: byte j = element j | element
// j + 16 << 4; f16 d, or f16 {d, m}) into four 16-byte chunks of its row: v_perm_b32 builds (1024 + q) half pairs, packed f16
// math applies (q - 8) d or q d + m — the values the f16 image holds (engine.hip row_to_f16).  Shared by layer_tail.hip and qkv_attention2.hip.
// PERM: the k order inside every group of 16 is [0-3, 8-11, 4-7, 12-15] (GemmWeight::w16p), plain otherwise.
struct RawBlock { uint4 q; unsigned sc; };
template <int WT>
__device__ __forceinline__ RawBlock q4_load_block(const uint4 *qs, const void *sc, size_t index) {
    RawBlock r;
    r.q = qs[index];
    r.sc = WT == GW_Q4_0 ? (unsigned)((const unsigned short *)sc)[index] : ((const unsigned *)sc)[index];
    return r;
}
template <int WT, bool PERM, class ChunkPtr>
__device__ __forceinline__ void q4_expand_block(const RawBlock &r, ChunkPtr chunk_ptr) {
    const unsigned w[4] = {r.q.x, r.q.y, r.q.z, r.q.w};
    f16x2 d2, m2;
    if (WT == GW_Q4_0) {
        const _Float16 d = __builtin_bit_cast(_Float16, (unsigned short)(r.sc & 0xffffu));
        d2 = (f16x2){d, d};
        m2 = (f16x2){(_Float16)0, (_Float16)0};
    } else {
        const f16x2 dm = __builtin_bit_cast(f16x2, r.sc);
        d2 = (f16x2){dm[0], dm[0]};
        m2 = (f16x2){dm[1], dm[1]};
    }
    // The constants are made HERE, per call, behind opaque moves (gfx9 VOP3 takes no literals and one scalar operand: the
    // byte source of v_perm_b32 has to sit in a vector register): hoisted out of the caller's loop they cost three registers for
    // the whole kernel — or, in kernels that have none to spare, a scratch reload per use.
    unsigned magic, sel01, sel23, offb;
    asm volatile("v_mov_b32 %0, 0x64646464" : "=v"(magic));
    asm volatile("s_mov_b32 %0, 0x04010400" : "=s"(sel01));
    asm volatile("s_mov_b32 %0, 0x04030402" : "=s"(sel23));
    if (WT == GW_Q4_0) asm volatile("s_mov_b32 %0, 0x64086408" : "=s"(offb));      // 1032, 1032
    else asm volatile("s_mov_b32 %0, 0x64006400" : "=s"(offb));                    // 1024, 1024
    const f16x2 off = __builtin_bit_cast(f16x2, offb);
    auto four = [&](unsigned word, bool high, unsigned &o0, unsigned &o1) __attribute__((always_inline)) {
        const unsigned n4 = (high ? (word >> 4) : word) & 0x0f0f0f0fu;
        f16x2 v0 = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(magic, n4, sel01)) - off;
        f16x2 v1 = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(magic, n4, sel23)) - off;
        if (WT == GW_Q4_0) { v0 = v0 * d2; v1 = v1 * d2; }
        else { v0 = v0 * d2 + m2; v1 = v1 * d2 + m2; }
        o0 = __builtin_bit_cast(unsigned, v0);
        o1 = __builtin_bit_cast(unsigned, v1);
    };
#pragma unroll
    for (int h = 0; h < 2; ++h)                               // elements 0..15 (low nibbles) / 16..31 (high nibbles)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            // chunk 2 h + pr: plain = elements 8 pr .. + 8 of the half (words 2 pr, 2 pr + 1); PERM = {4 pr .., 8 + 4 pr ..} (words pr, pr + 2)
            uint4 out;
            four(w[PERM ? pr : 2 * pr], h, out.x, out.y);
            four(w[PERM ? pr + 2 : 2 * pr + 1], h, out.z, out.w);
            *(uint4 *)chunk_ptr(2 * h + pr) = out;
        }
}

// One 16-byte chunk (eight weights, plain k order: elements 8 c .. 8 c + 7 of the block) of a q4 block, the arithmetic of
// q4_expand_block: for kernels that spread a block's expansion over several issue gaps (gemm256.hip).
template <int WT>
__device__ __forceinline__ uint4 q4_expand_chunk(const RawBlock &r, int c) {
    f16x2 d2, m2;
    if (WT == GW_Q4_0) {
        const _Float16 d = __builtin_bit_cast(_Float16, (unsigned short)(r.sc & 0xffffu));
        d2 = (f16x2){d, d};
        m2 = (f16x2){(_Float16)0, (_Float16)0};
    } else {
        const f16x2 dm = __builtin_bit_cast(f16x2, r.sc);
        d2 = (f16x2){dm[0], dm[0]};
        m2 = (f16x2){dm[1], dm[1]};
    }
    unsigned magic, sel01, sel23, offb;
    asm volatile("v_mov_b32 %0, 0x64646464" : "=v"(magic));
    asm volatile("s_mov_b32 %0, 0x04010400" : "=s"(sel01));
    asm volatile("s_mov_b32 %0, 0x04030402" : "=s"(sel23));
    if (WT == GW_Q4_0) asm volatile("s_mov_b32 %0, 0x64086408" : "=s"(offb));      // 1032, 1032
    else asm volatile("s_mov_b32 %0, 0x64006400" : "=s"(offb));                    // 1024, 1024
    const f16x2 off = __builtin_bit_cast(f16x2, offb);
    auto four = [&](unsigned word, bool high, unsigned &o0, unsigned &o1) __attribute__((always_inline)) {
        const unsigned n4 = (high ? (word >> 4) : word) & 0x0f0f0f0fu;
        f16x2 v0 = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(magic, n4, sel01)) - off;
        f16x2 v1 = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(magic, n4, sel23)) - off;
        if (WT == GW_Q4_0) { v0 = v0 * d2; v1 = v1 * d2; }
        else { v0 = __builtin_elementwise_fma(v0, d2, m2); v1 = __builtin_elementwise_fma(v1, d2, m2); }   // (one rounding: engine.hip row_to_f16)
        o0 = __builtin_bit_cast(unsigned, v0);
        o1 = __builtin_bit_cast(unsigned, v1);
    };
    const bool high = c >= 2;
    const unsigned w0 = (c & 1) ? r.q.z : r.q.x, w1 = (c & 1) ? r.q.w : r.q.y;
    uint4 out;
    four(w0, high, out.x, out.y);
    four(w1, high, out.z, out.w);
    return out;
}

// LayerNorm of one token's row from f32 values, the way layer_tail.hip's lanes and wave pairs do it (the latency route:
// skinny.hip, and the per-head form of qkv_attention2.hip): lane = (token l31, half hi) holds the
// 4-feature runs (n, g) = features 32 n + 8 g + 4 hi .. + 3 of its token (H / 2 values, loaded 16 bytes at a time).
// PAIR (LayerNorm 1): the statistics are the sum of two half-row sums (features with (f & 127) < 64: layer_tail's U wave;
// the rest: its D wave), each summed block by block, register by register, then across the lane halves; !PAIR (LayerNorm
// 2): one sum over all blocks (the D wave owns whole rows).  Result: the normalised runs, f16.
template <bool PAIR, int NT, class Run, class Landed>
__device__ __forceinline__ void layernorm_runs_of(Run x, Landed landed, const float *gamma, const float *beta, int hi, f16x4 (&y)[4 * NT][4]) {
    constexpr int H = 128 * NT;
    // the parameter reads run AHEAD blocks (eight 16-byte loads each) in front of their use — spelled out: the compiler's own
    // order waits for every pair of loads in turn (48 L2 round trips)
    constexpr int AHEAD = 2;
    f32x4 gv[AHEAD + 1][4], bv[AHEAD + 1][4];
    auto request = [&](int n) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            gv[n % (AHEAD + 1)][g] = *(const f32x4 *)(gamma + 32 * n + 8 * g + 4 * hi);
            bv[n % (AHEAD + 1)][g] = *(const f32x4 *)(beta + 32 * n + 8 * g + 4 * hi);
        }
    };
#pragma unroll
    for (int n = 0; n < AHEAD; ++n) request(n);
    landed();                          // (runs that arrive by LDS-DMA: the wait for them, behind the first parameter requests)
    float s1 = 0.f, s2 = 0.f;
    auto add_block = [&](float &a1, float &a2, int n) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = x(n, g);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a1 += v[e]; a2 = __builtin_fmaf(v[e], v[e], a2); }
        }
    };
    if constexpr (PAIR) {
        float t1[2] = {0.f, 0.f}, t2[2] = {0.f, 0.f};
#pragma unroll
        for (int role = 0; role < 2; ++role)
#pragma unroll
            for (int b = 0; b < 2 * NT; ++b) add_block(t1[role], t2[role], (b >> 1) * 4 + role * 2 + (b & 1));
#pragma unroll
        for (int role = 0; role < 2; ++role) { t1[role] += __shfl_xor(t1[role], 32); t2[role] += __shfl_xor(t2[role], 32); }
        s1 = t1[0] + t1[1]; s2 = t2[0] + t2[1];
    } else {
#pragma unroll
        for (int n = 0; n < 4 * NT; ++n) add_block(s1, s2, n);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    }
    float rstd, nmr;
    layernorm_scale(s1, s2, 1.0f / H, rstd, nmr);
    asm volatile("" ::: "memory");     // (runs that come from LDS are read again instead of being kept: registers for the loads in flight)
#pragma unroll
    for (int n = 0; n < 4 * NT; ++n) {
        if (n + AHEAD < 4 * NT) request(n + AHEAD);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = x(n, g), gq = gv[n % (AHEAD + 1)][g], bq = bv[n % (AHEAD + 1)][g];
            // (f32 results, THEN f16: fused into v_fma_mixlo_f16 — one rounding — the compiler's choice depends on the kernel around
            // it.  One opaque hand-over per run of four, so that the f32 math itself still packs into v_pk_fma_f32.)
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(v[e], gq[e] * rstd, __builtin_fmaf(gq[e], nmr, bq[e]));
            asm("" : "+v"(r));
#pragma unroll
            for (int e = 0; e < 4; ++e) y[n][g][e] = (_Float16)r[e];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the runs from memory: all of the row in registers first
template <bool PAIR, int NT>
__device__ __forceinline__ void layernorm_runs(const float *row, const float *gamma, const float *beta, int hi, f16x4 (&y)[4 * NT][4]) {
    f32x4 x[4 * NT][4];
#pragma unroll
    for (int n = 0; n < 4 * NT; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) x[n][g] = *(const f32x4 *)(row + 32 * n + 8 * g + 4 * hi);
    layernorm_runs_of<PAIR, NT>([&](int n, int g) __attribute__((always_inline)) { return x[n][g]; }, [] {}, gamma, beta, hi, y);
}

}  // namespace bert_hip
