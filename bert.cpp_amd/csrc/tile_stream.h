// tile_stream.h — shared device helpers of the "tile stream" kernels (ffn_fused.hip, panel_gemm.hip):
// 128-row x 64-half operand tiles travel HBM/L2 -> LDS by global_load_lds_dwordx4 into a ring of
// 32-KiB slots, several tiles ahead of their use, and are retired with a counted s_waitcnt vmcnt(N)
// plus one s_barrier per tile.  The 16-byte chunk of every row is XOR-swizzled on the SOURCE side
// (LDS-DMA writes lane-linearly) and un-swizzled by the ds_read_b128 fragment reads.
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

constexpr int FF_SLOT = 32768;                 // ring slot: 16 KiB activation k-tile + 16 KiB weight tile
constexpr int FF_RING = 3 * FF_SLOT;
constexpr int FF_HC = FF_RING;                 // [128][128] f16 = 32 KiB
constexpr int FF_CONST = FF_HC + 32768;        // b1[I], then b2, gamma, beta [H], then LN scratch

__device__ __forceinline__ int off64(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int off_hc(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }

// 128 rows x 64 halfs (128-B rows) -> 16 KiB LDS tile; this wave moves rows [wave*16, wave*16+16).
// `base` is wave-uniform (SGPR pair), `loff[i]` the lane's byte offset inside the tile's source rows,
// so the load uses the scalar-base + 32-bit-VGPR-offset form and costs no address VGPRs per tile.
__device__ __forceinline__ void dma_tile8(const half_t *base, const unsigned (&loff)[2], char *tile, int wave) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds(AS_GLOBAL((const char *)base + loff[i]), AS_LDS(tile + (wave * 2 + i) * 1024), 16, 0, 0);
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// ---- q4_0 / q4_1 weight tiles (layout: kernels.h GemmWeight).  A [128 rows x 64 k] tile is 256 blocks of
// 32 weights; with 512 threads, thread tid dequantises HALF a block: row = tid >> 2, block = (tid >> 1) & 1,
// half = tid & 1 (0 = low nibbles = elements 0-15, 1 = high nibbles = elements 16-31).  The 16 bytes of
// nibbles + the scale are fetched with two ordinary loads one interval before they are needed and expanded
// in registers: v_perm_b32 builds (1024 + q) half pairs, packed f16 math applies (q - 8) * d or q * d + m.
struct QRegs { uint4 q; unsigned sc; };

template <int WT>
__device__ __forceinline__ QRegs q4_fetch(const uint4 *qs, const void *sc, size_t tile_index, int tid) {
    const size_t bi = tile_index * 256 + (tid >> 1);
    QRegs r;
    r.q = qs[bi];
    r.sc = WT == GW_Q4_0 ? (unsigned)((const unsigned short *)sc)[bi] : ((const unsigned *)sc)[bi];
    return r;
}

template <int WT>
__device__ __forceinline__ void q4_expand_to_lds(const QRegs &r, char *tile, int tid) {
    const int row = tid >> 2, blk = (tid >> 1) & 1, half = tid & 1;
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
    const f16x2 off = WT == GW_Q4_0 ? (f16x2){(_Float16)1032.0f, (_Float16)1032.0f}
                                     : (f16x2){(_Float16)1024.0f, (_Float16)1024.0f};
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        unsigned o[4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned word = w[jj * 2 + u];
            const unsigned n4 = (half ? (word >> 4) : word) & 0x0f0f0f0fu;
            const unsigned p01 = __builtin_amdgcn_perm(0x64646464u, n4, 0x04010400u);
            const unsigned p23 = __builtin_amdgcn_perm(0x64646464u, n4, 0x04030402u);
            f16x2 v0 = __builtin_bit_cast(f16x2, p01) - off, v1 = __builtin_bit_cast(f16x2, p23) - off;
            if (WT == GW_Q4_0) { v0 = v0 * d2; v1 = v1 * d2; }
            else { v0 = v0 * d2 + m2; v1 = v1 * d2 + m2; }
            o[2 * u] = __builtin_bit_cast(unsigned, v0);
            o[2 * u + 1] = __builtin_bit_cast(unsigned, v1);
        }
        uint4 out;
        out.x = o[0]; out.y = o[1]; out.z = o[2]; out.w = o[3];
        *(uint4 *)(tile + off64(row, blk * 4 + half * 2 + jj)) = out;
    }
}

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

__device__ __forceinline__ float gelu_fast(float x) {
    const float c1 = -2.0f * 0.79788456080286535588f * 1.44269504088896340736f;
    const float c2 = c1 * 0.044715f;
    const float t = x * __builtin_fmaf(x * x, c2, c1);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}

template <int G>
__device__ __forceinline__ void wait_vm_barrier() {
    // retire everything but the newest G DMA pieces of this wave, make own LDS writes visible, sync
    if (G == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (G == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


// Final epilogue of the kernels whose workgroup owns complete rows (H = 128 * NT features of 128
// tokens in acc[NT][2]): v = acc + bias + residual, LayerNorm (eps 1e-5, two-pass like ggml_norm),
// gamma/beta, f16, then a transpose through LDS (`stage`, >= 128*H*2 bytes) so that every global
// store is a full row segment.  lane owns tokens wt*64 + j*32 + l31 and features n*128 + wq*32 + 8g + 4hi + e.
// cbias/cgamma/cbeta are LDS copies; `red` is 512 floats of LDS scratch.  Must be called by all 512
// threads after the last ring/hc use (it starts with a barrier before touching `stage`).
template <int NT>
__device__ __forceinline__ void ln_epilogue(f32x16 (&acc)[NT][2], const float *cbias, const float *cgamma,
                                            const float *cbeta, float *red, const half_t *resid, half_t *out,
                                            char *stage, int tid, int wt, int wq, int l31, int hi) {
    constexpr int H = 128 * NT, CPR = H / 8;                   // CPR = 16-byte chunks per row
    half_t *Cs = (half_t *)stage;                              // [128 tokens][H] f16, 16-B chunk ^ (tok & 15)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // ---- residual tile -> LDS by DMA, already in the swizzled staging layout (every lane later reads and
    // then overwrites exactly its own 8-byte runs).  Coalesced 16-B pieces instead of scattered 8-B loads.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // all waves are past their last ring read
#pragma unroll
    for (int q = 0; q < 4 * NT; ++q) {
        const int piece = wave * 4 * NT + q;                   // 1-KiB piece of the 128*H*2-byte tile
        const int li = piece * 64 + lane, row = li / CPR, slot = li - row * CPR;
        const int c = (slot & ~15) | ((slot ^ row) & 15);
        __builtin_amdgcn_global_load_lds(AS_GLOBAL(resid + (size_t)row * H + c * 8), AS_LDS(stage + piece * 1024), 16, 0, 0);
    }
    float sum[2] = {0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *(const f32x4 *)(cbias + n * 128 + wq * 32 + 8 * g + 4 * hi);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[n][j][4 * g + e] += bv[e];
        }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = n * 128 + wq * 32 + 8 * g + 4 * hi, chunk = f0 >> 3;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int tok = wt * 64 + j * 32 + l31;
                const f16x4 rv = *(const f16x4 *)(Cs + (size_t)tok * H + (((chunk & ~15) | ((chunk ^ tok) & 15)) << 3) + (f0 & 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[n][j][4 * g + e] + (float)rv[e];
                    acc[n][j][4 * g + e] = v;
                    sum[j] += v;
                }
            }
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        sum[j] += __shfl_xor(sum[j], 32);
        if (hi == 0) red[wq * 128 + wt * 64 + j * 32 + l31] = sum[j];
    }
    __syncthreads();
    float mean[2], sq[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tok = wt * 64 + j * 32 + l31;
        mean[j] = ((red[tok] + red[128 + tok]) + (red[256 + tok] + red[384 + tok])) * (1.0f / H);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = acc[n][j][r] - mean[j];
                acc[n][j][r] = d;
                sq[j] += d * d;
            }
    __syncthreads();                                           // everyone has read the sums
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        sq[j] += __shfl_xor(sq[j], 32);
        if (hi == 0) red[wq * 128 + wt * 64 + j * 32 + l31] = sq[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tok = wt * 64 + j * 32 + l31;
        const float var = ((red[tok] + red[128 + tok]) + (red[256 + tok] + red[384 + tok])) * (1.0f / H);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f0 = n * 128 + wq * 32 + 8 * g + 4 * hi;
                const f32x4 gv = *(const f32x4 *)(cgamma + f0), bv = *(const f32x4 *)(cbeta + f0);
                f16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (_Float16)(gv[e] * (acc[n][j][4 * g + e] * rstd) + bv[e]);
                const int chunk = f0 >> 3;
                *(f16x4 *)(Cs + (size_t)tok * H + (((chunk & ~15) | ((chunk ^ tok) & 15)) << 3) + (f0 & 4)) = o;
            }
    }
    __syncthreads();
    for (int idx = tid; idx < 128 * CPR; idx += 512) {
        const int tok = idx / CPR, chunk = idx - tok * CPR;
        const uint4 v = *(const uint4 *)(Cs + (size_t)tok * H + (((chunk & ~15) | ((chunk ^ tok) & 15)) << 3));
        *(uint4 *)(out + (size_t)tok * H + chunk * 8) = v;
    }
}

}  // namespace bert_hip
