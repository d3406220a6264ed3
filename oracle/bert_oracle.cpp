// oracle/bert_oracle.cpp — CPU restatement of bert.cpp's tokenizer, model loader and bert_eval
// forward pass.  TEST INFRASTRUCTURE ONLY: nothing under oracle/ is linked into, imported by or
// called from the product (libbert.so); only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use it, and only as the checker / the timed CPU baseline.
//
// PARITY STATUS
//   * token ids: PINNED against the reference's own known-answer vectors
//     (reference examples/test_tokenizer.cpp:70-73) — tests/test_oracle.py (oracle) and tests/test_host.py (product tokenizer), re-run on the GPU box by tests/test_gpu_exactness.py.
//   * embeddings: "PARITY UNPINNED" at the ggml boundary.  All arithmetic of the reference lives
//     in ggerganov/ggml, an un-vendored, un-pinned git submodule (reference .gitmodules:1-3,
//     /root/reference/ggml is empty) — the reference cannot be built here, and it ships no
//     numeric golden vectors for bert_eval.  This file restates the published ggml algorithms of
//     the API era the reference calls (mid-2023: ggml_graph_compute_with_ctx present, ggml_norm
//     without eps argument) and is additionally cross-checked against an independent
//     implementation (HuggingFace BertModel, tests/golden/make_golden.py).
//
// What follows what (reference file:line):
//   oracle_tokenize ............. bert.cpp:199-325  (utf8_len, stripAccents, normalize, regex
//                                                    split, greedy WordPiece, truncation)
//   vocab maps .................. bert.cpp:379-403
//   oracle_load ................. bert.cpp:331-669  (file format; see also
//                                                    models/convert-to-ggml.py:68-108)
//   oracle_eval ................. bert.cpp:750-939  (one sentence = one pass of the loop body)
//   ggml op numerics ............ SURVEY.md Appendix C (upstream ggml, not in the tree)
//
// mode 0 ("ggml"):  ggml-faithful numerics — f16 weights: activations rounded to f16 at every
//                   weight mat-mul, f32 accumulate; q4_0: activations -> Q8_0 blocks, integer block
//                   dot; q4_1: activations -> Q8_1; exp and GELU through fp16 lookup semantics;
//                   LayerNorm eps 1e-5 with double accumulators.
// mode 1 ("plain"): weights dequantized exactly to f32, no activation rounding, expf / tanhf.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <regex>
#include <string>
#include <vector>
#if defined(__F16C__)
#include <immintrin.h>
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------------------------
// fp16 <-> fp32 (IEEE binary16, round to nearest even) — GGML_FP32_TO_FP16 / GGML_FP16_TO_FP32
// ---------------------------------------------------------------------------------------------
inline float h2f_soft(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {                                   // subnormal: normalise
            int e = 0;
            while (!(man & 0x400u)) { man <<= 1; ++e; }
            bits = sign | (uint32_t)(113 - e) << 23 | (man & 0x3ffu) << 13;
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | man << 13;
    else bits = sign | (exp + 112) << 23 | man << 13;
    float f; memcpy(&f, &bits, 4); return f;
}
inline uint16_t f2h_soft(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
    if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);             // NaN
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);            // >= 65520 rounds to inf
    if (x < 0x33000000u) return (uint16_t)sign;                          // < 2^-25 rounds to zero
    const uint32_t e = x >> 23, m = (x & 0x7fffffu) | 0x800000u;
    uint32_t half, rem, mid;
    if (e < 113) {                                                       // result is subnormal
        const uint32_t shift = 126 - e;
        half = m >> shift; rem = m & ((1u << shift) - 1); mid = 1u << (shift - 1);
    } else {
        half = ((e - 112) << 10) | ((m & 0x7fffffu) >> 13); rem = m & 0x1fffu; mid = 0x1000u;
    }
    if (rem > mid || (rem == mid && (half & 1u))) ++half;                // carries into the exponent correctly
    return (uint16_t)(sign | half);
}
#if defined(__F16C__)
inline float h2f(uint16_t h) { return _cvtsh_ss(h); }
inline uint16_t f2h(float f) { return _cvtss_sh(f, 0); }
#else
inline float h2f(uint16_t h) { return h2f_soft(h); }
inline uint16_t f2h(float f) { return f2h_soft(f); }
#endif
inline float round_f16(float f) { return h2f(f2h(f)); }

// ---------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------
enum { T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q4_1 = 3 };
constexpr int QK = 32;

struct Matrix {          // a 2-D weight [rows = ne1][cols = ne0], row-major, in file type `type`
    int type = 0, rows = 0, cols = 0;
    std::vector<float> f;      // exact f32 image (f32 / f16 -> f32 / q4 dequantized)
    std::vector<int8_t> q;     // q4 only: (q-8) for q4_0, q for q4_1, one int8 per weight
    std::vector<float> d, m;   // q4 only: per-block scale (and min)
};

struct Layer {
    Matrix q_w, k_w, v_w, o_w, ff_i_w, ff_o_w;
    std::vector<float> q_b, k_b, v_b, o_b, ff_i_b, ff_o_b, ln_att_w, ln_att_b, ln_out_w, ln_out_b;
};

struct Vocab {            // bert.cpp:60-67
    std::map<std::string, int32_t> token_to_id, subword_token_to_id;
    std::map<int32_t, std::string> id_to_token, id_to_subword_token;
};

}  // namespace

struct oracle_ctx {
    int32_t n_vocab = 0, n_max_tokens = 0, n_embd = 0, n_intermediate = 0, n_head = 0, n_layer = 0, ftype = 0;
    Vocab vocab;
    Matrix word_emb, type_emb, pos_emb;
    std::vector<float> ln_e_w, ln_e_b;
    std::vector<Layer> layers;
    bool weights_loaded = false;
};

namespace {

void fill_matrix(Matrix &M, int type, int rows, int cols, const uint8_t *src) {
    M.type = type; M.rows = rows; M.cols = cols;
    const size_t n = (size_t)rows * cols;
    M.f.resize(n);
    if (type == T_F32) memcpy(M.f.data(), src, n * 4);
    else if (type == T_F16) {
        const uint16_t *h = (const uint16_t *)src;
        for (size_t i = 0; i < n; ++i) M.f[i] = h2f(h[i]);
    } else {
        const int bs = type == T_Q4_0 ? 18 : 20;
        const size_t nb = n / QK;
        M.q.resize(n); M.d.resize(nb); if (type == T_Q4_1) M.m.resize(nb);
        for (size_t b = 0; b < nb; ++b) {
            const uint8_t *p = src + b * bs;
            uint16_t dh; memcpy(&dh, p, 2);
            const float d = h2f(dh);
            float mn = 0.f;
            const uint8_t *qs = p + 2;
            if (type == T_Q4_1) { uint16_t mh; memcpy(&mh, p + 2, 2); mn = h2f(mh); qs = p + 4; M.m[b] = mn; }
            M.d[b] = d;
            for (int j = 0; j < QK / 2; ++j) {      // dequantize_row_q4_0 / _q4_1
                const int x0 = qs[j] & 0x0F, x1 = qs[j] >> 4;
                if (type == T_Q4_0) {
                    M.q[b * QK + j] = (int8_t)(x0 - 8); M.q[b * QK + j + QK / 2] = (int8_t)(x1 - 8);
                    M.f[b * QK + j] = (x0 - 8) * d;     M.f[b * QK + j + QK / 2] = (x1 - 8) * d;
                } else {
                    M.q[b * QK + j] = (int8_t)x0;       M.q[b * QK + j + QK / 2] = (int8_t)x1;
                    M.f[b * QK + j] = x0 * d + mn;      M.f[b * QK + j + QK / 2] = x1 * d + mn;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// tokenizer — bert.cpp:199-325
// ---------------------------------------------------------------------------------------------
size_t utf8_len(char src) {                                   // bert.cpp:199-204
    static const size_t lookup[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4};
    return lookup[static_cast<uint8_t>(src) >> 4];
}

const std::map<std::string, char> &accent_map() {            // bert.cpp:209-219 (52 entries)
    static const std::map<std::string, char> m = [] {
        std::map<std::string, char> a;
        const char *groups[][2] = {
            {"ÀÁÂÃÄÅ", "A"}, {"àáâãäå", "a"}, {"ÈÉÊË", "E"}, {"èéêë", "e"}, {"ÌÍÎÏ", "I"}, {"ìíîï", "i"},
            {"ÒÓÔÕÖ", "O"}, {"òóôõö", "o"}, {"ÙÚÛÜ", "U"}, {"ùúûü", "u"}, {"Ý", "Y"}, {"ý", "y"},
            {"Ç", "C"}, {"ç", "c"}, {"Ñ", "N"}, {"ñ", "n"}};
        for (auto &g : groups) {
            std::string s = g[0];
            for (size_t i = 0; i + 1 < s.size(); i += 2) a[s.substr(i, 2)] = g[1][0];
        }
        return a;
    }();
    return m;
}

std::string strip_accents(const std::string &in) {            // bert.cpp:206-238
    std::string out;
    const auto &am = accent_map();
    for (size_t i = 0; i < in.length();) {
        const int len = (int)utf8_len(in[i]);
        std::string cur = in.substr(i, len);
        auto it = am.find(cur);
        if (it != am.end()) out += it->second; else out += cur;
        i += len;
    }
    return out;
}

std::string normalize_prompt(const std::string &text) {       // bert.cpp:240-251
    std::string t = strip_accents(text);
    for (size_t i = 0; i < t.size(); i += utf8_len(t[i])) {
        char c = t[i];
        if (c >= 'A' && c <= 'Z') t[i] = c - 'A' + 'a';
    }
    return t;
}

void tokenize(const Vocab &vocab, const char *text, int32_t *tokens, int32_t *n_tokens, int32_t n_max_tokens,
              bool quiet) {                                    // bert.cpp:252-325
    std::string str = normalize_prompt(text);
    std::vector<std::string> words;
    {
        static const std::regex re(R"([[:punct:]]|[[:alpha:]]+|[[:digit:]]+)");
        // regex_search + suffix loop of the reference == successive non-overlapping matches
        for (auto it = std::sregex_iterator(str.begin(), str.end(), re); it != std::sregex_iterator(); ++it)
            words.push_back(it->str());
    }
    int32_t t = 0;
    tokens[t++] = 101;
    for (const auto &word : words) {
        if (word.empty()) continue;
        int i = 0;
        const int n = (int)word.size();
        const auto *token_map = &vocab.token_to_id;
        while (i < n) {
            if (t >= n_max_tokens - 1) break;
            int j = n;
            bool found = false;
            while (j > i) {
                auto it = token_map->find(word.substr(i, j - i));
                if (it != token_map->end()) {
                    tokens[t++] = it->second;
                    i = j;
                    token_map = &vocab.subword_token_to_id;
                    found = true;
                    break;
                }
                --j;
            }
            if (!found) {     // no prefix matched: drop one byte, continue in the subword map
                if (!quiet) fprintf(stderr, "%s: unknown token '%s'\n", "bert_tokenize", word.substr(i, 1).data());
                token_map = &vocab.subword_token_to_id;
                ++i;
            }
        }
    }
    tokens[t++] = 102;
    *n_tokens = t;
}

// ---------------------------------------------------------------------------------------------
// ggml op restatements
// ---------------------------------------------------------------------------------------------
inline float gelu_f32(float x) {                               // ggml_gelu_f32
    const float GELU_COEF_A = 0.044715f, SQRT_2_OVER_PI = 0.79788456080286535587989211986876f;
    return 0.5f * x * (1.0f + tanhf(SQRT_2_OVER_PI * x * (1.0f + GELU_COEF_A * x * x)));
}

struct Tables {            // table_exp_f16 / table_gelu_f16 semantics, built lazily
    std::vector<uint16_t> gelu, expt;
    Tables() : gelu(65536), expt(65536) {
        for (int i = 0; i < 65536; ++i) {
            const float f = h2f((uint16_t)i);
            gelu[i] = f2h(gelu_f32(f));
            expt[i] = f2h(expf(f));
        }
    }
};
const Tables &tables() { static const Tables t; return t; }

// ---- SIMD helpers (GCC vector extensions: lowered to AVX-512 / AVX2 / NEON by -march=native) ----
typedef float v16f __attribute__((vector_size(64)));
typedef int v16i __attribute__((vector_size(64)));
typedef short v32s __attribute__((vector_size(64)));
typedef short v16s __attribute__((vector_size(32)));
typedef signed char v32c __attribute__((vector_size(32)));

inline float hsum(v16f v) {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    return s;
}

// 4 tokens x 4 outputs register block of f32 dot products (f32 accumulate, like ggml_vec_dot_f32 /
// the F16C path of ggml_vec_dot_f16: products and sums in f32 SIMD lanes, reduced at the end).
void dot_block_f32(const float *x, int ldx, const float *w, int ldw, int K, int nt, int nn, float out[4][4]) {
    v16f acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = (v16f){0};
    int k = 0;
    for (; k + 16 <= K; k += 16) {
        v16f xv[4], wv[4];
        for (int a = 0; a < 4; ++a) memcpy(&xv[a], x + (size_t)(a < nt ? a : 0) * ldx + k, 64);
        for (int b = 0; b < 4; ++b) memcpy(&wv[b], w + (size_t)(b < nn ? b : 0) * ldw + k, 64);
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] += xv[a] * wv[b];
    }
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) {
        float s = hsum(acc[a][b]);
        if (a < nt && b < nn) for (int kk = k; kk < K; ++kk) s += x[(size_t)a * ldx + kk] * w[(size_t)b * ldw + kk];
        out[a][b] = s;
    }
}

// y[t][n] = sum_k W[n][k] * x[t][k] + b[n]   (ggml_mul_mat(W, X) + repeat(b)), X is [T][K] f32
void linear(const Matrix &W, const std::vector<float> &bias, const float *X, int T, float *Y, int mode) {
    const int K = W.cols, N = W.rows;
    if (mode == 1 || W.type == T_F32 || W.type == T_F16) {
        // f16 weights in ggml mode: src1 is converted to f16 in the work buffer, f32 accumulate
        std::vector<float> xr;
        const float *Xs = X;
        if (mode == 0 && W.type == T_F16) {
            xr.resize((size_t)T * K);
            for (size_t i = 0; i < xr.size(); ++i) xr[i] = round_f16(X[i]);
            Xs = xr.data();
        }
        const int tb = (T + 3) / 4, nb = (N + 3) / 4;
#pragma omp parallel for schedule(static) collapse(2)
        for (int ti = 0; ti < tb; ++ti)
            for (int ni = 0; ni < nb; ++ni) {
                const int t0 = ti * 4, n0 = ni * 4, nt = std::min(4, T - t0), nn = std::min(4, N - n0);
                float o[4][4];
                dot_block_f32(Xs + (size_t)t0 * K, K, W.f.data() + (size_t)n0 * K, K, K, nt, nn, o);
                for (int a = 0; a < nt; ++a)
                    for (int b = 0; b < nn; ++b) Y[(size_t)(t0 + a) * N + n0 + b] = o[a][b] + bias[n0 + b];
            }
        return;
    }
    // q4_0 x q8_0  /  q4_1 x q8_1
    const int nb = K / QK;
    std::vector<int8_t> xq((size_t)T * K);
    std::vector<float> xd((size_t)T * nb), xs((size_t)T * nb);
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; ++t)
        for (int b = 0; b < nb; ++b) {          // quantize_row_q8_0 / quantize_row_q8_1
            const float *x = X + (size_t)t * K + b * QK;
            float amax = 0.f;
            for (int j = 0; j < QK; ++j) amax = std::max(amax, fabsf(x[j]));
            const float d = amax / ((1 << 7) - 1);
            const float id = d ? 1.0f / d : 0.0f;
            int sum = 0;
            for (int j = 0; j < QK; ++j) {
                const int8_t v = (int8_t)roundf(x[j] * id);
                xq[(size_t)t * K + b * QK + j] = v; sum += v;
            }
            if (W.type == T_Q4_0) xd[(size_t)t * nb + b] = round_f16(d);       // block_q8_0.d is fp16
            else { xd[(size_t)t * nb + b] = d; xs[(size_t)t * nb + b] = d * sum; }  // block_q8_1 {float d, s}
        }
    // integer block dot: products of (q4 [-8,15]) x (q8 [-127,127]) in int16 lanes, pairs summed in
    // int32, scaled by d4*d8 per block and accumulated in f32 SIMD lanes — the shape of ggml's AVX2
    // ggml_vec_dot_q4_0_q8_0 (integer partial sums -> float -> fmadd with the block scale).
    const int tbk = (T + 3) / 4;
#pragma omp parallel for schedule(static) collapse(2)
    for (int ti = 0; ti < tbk; ++ti)
        for (int n = 0; n < N; ++n) {
            const int t0 = ti * 4, nt = std::min(4, T - t0);
            const int8_t *w = W.q.data() + (size_t)n * K;
            v16f acc[4] = {(v16f){0}, (v16f){0}, (v16f){0}, (v16f){0}};
            float macc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int b = 0; b < nb; ++b) {
                v32c wc; memcpy(&wc, w + b * QK, 32);
                const v32s w16 = __builtin_convertvector(wc, v32s);
                const float d4 = W.d[(size_t)n * nb + b];
                for (int a = 0; a < nt; ++a) {
                    const size_t t = (size_t)(t0 + a);
                    v32c xc; memcpy(&xc, xq.data() + t * K + b * QK, 32);
                    const v32s p = w16 * __builtin_convertvector(xc, v32s);
                    v16s lo, hi; memcpy(&lo, &p, 32); memcpy(&hi, (const char *)&p + 32, 32);
                    const v16i s32 = __builtin_convertvector(lo, v16i) + __builtin_convertvector(hi, v16i);
                    const float sc = d4 * xd[t * nb + b];
                    acc[a] += __builtin_convertvector(s32, v16f) * sc;
                    if (W.type == T_Q4_1) macc[a] += W.m[(size_t)n * nb + b] * xs[t * nb + b];
                }
            }
            for (int a = 0; a < nt; ++a) Y[(size_t)(t0 + a) * N + n] = hsum(acc[a]) + macc[a] + bias[n];
        }
}

// ggml_norm (eps = 1e-5, double accumulators) followed by gamma * x + beta   (bert.cpp:806-814)
void layer_norm(float *X, int T, int H, const std::vector<float> &g, const std::vector<float> &b) {
    const float eps = 1e-5f;
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; ++t) {
        float *x = X + (size_t)t * H;
        double sum = 0.0;
        for (int i = 0; i < H; ++i) sum += (double)x[i];
        const float mean = (float)(sum / H);
        double sum2 = 0.0;
        for (int i = 0; i < H; ++i) { const float v = x[i] - mean; x[i] = v; sum2 += (double)(v * v); }
        const float variance = (float)(sum2 / H);
        const float scale = 1.0f / sqrtf(variance + eps);
        for (int i = 0; i < H; ++i) x[i] = g[i] * (x[i] * scale) + b[i];
    }
}

void get_row(const Matrix &M, int row, float *dst) {           // ggml_get_rows: dequantize to f32
    memcpy(dst, M.f.data() + (size_t)row * M.cols, sizeof(float) * M.cols);
}

}  // namespace

// =================================================================================================
// C ABI (ctypes-friendly)
// =================================================================================================
extern "C" {

oracle_ctx *oracle_load(const char *fname, int vocab_only) {   // bert.cpp:331-669
    std::ifstream fin(fname, std::ios::binary);
    if (!fin) { fprintf(stderr, "oracle_load: failed to open '%s'\n", fname); return nullptr; }
    uint32_t magic = 0;
    fin.read((char *)&magic, 4);
    if (magic != 0x67676d6c) { fprintf(stderr, "oracle_load: bad magic in '%s'\n", fname); return nullptr; }
    auto *c = new oracle_ctx;
    int32_t hp[7];
    fin.read((char *)hp, sizeof(hp));
    c->n_vocab = hp[0]; c->n_max_tokens = hp[1]; c->n_embd = hp[2]; c->n_intermediate = hp[3];
    c->n_head = hp[4]; c->n_layer = hp[5]; c->ftype = hp[6];
    std::string word;
    for (int i = 0; i < c->n_vocab; ++i) {                     // bert.cpp:379-403
        uint32_t len = 0;
        fin.read((char *)&len, 4);
        word.resize(len);
        fin.read(&word[0], len);
        if (word.size() >= 2 && word[0] == '#' && word[1] == '#') {
            c->vocab.subword_token_to_id[word.substr(2)] = i;
            c->vocab.id_to_subword_token[i] = word;
        }
        if (c->vocab.token_to_id.count(word) == 0) {
            c->vocab.token_to_id[word] = i;
            c->vocab.id_to_token[i] = word;
        }
    }
    if (vocab_only) return c;
    if (c->ftype < 0 || c->ftype > 3) { fprintf(stderr, "oracle_load: bad f16 value %d\n", c->ftype); delete c; return nullptr; }
    c->layers.resize(c->n_layer);
    const int H = c->n_embd;
    std::vector<uint8_t> buf;
    while (true) {
        int32_t n_dims, length, ftype;
        fin.read((char *)&n_dims, 4); fin.read((char *)&length, 4); fin.read((char *)&ftype, 4);
        if (fin.eof()) break;
        int64_t ne[2] = {1, 1}, nel = 1;
        for (int i = 0; i < n_dims; ++i) { int32_t v; fin.read((char *)&v, 4); ne[i] = v; nel *= v; }
        std::string name(length, 0);
        fin.read(&name[0], length);
        size_t nbytes = ftype == T_F32 ? nel * 4 : ftype == T_F16 ? nel * 2 : ftype == T_Q4_0 ? nel / QK * 18 : nel / QK * 20;
        buf.resize(nbytes);
        fin.read((char *)buf.data(), nbytes);
        if (!fin) { fprintf(stderr, "oracle_load: truncated tensor '%s'\n", name.c_str()); delete c; return nullptr; }
        auto vec = [&](std::vector<float> &v) { v.resize(nel); memcpy(v.data(), buf.data(), nel * 4); };
        auto mat = [&](Matrix &M) { fill_matrix(M, ftype, (int)ne[1], (int)ne[0], buf.data()); };
        if (name == "embeddings.word_embeddings.weight") mat(c->word_emb);
        else if (name == "embeddings.token_type_embeddings.weight") mat(c->type_emb);
        else if (name == "embeddings.position_embeddings.weight") mat(c->pos_emb);
        else if (name == "embeddings.LayerNorm.weight") vec(c->ln_e_w);
        else if (name == "embeddings.LayerNorm.bias") vec(c->ln_e_b);
        else if (name.rfind("encoder.layer.", 0) == 0) {
            const size_t p = name.find('.', 14);
            const int il = std::stoi(name.substr(14, p - 14));
            if (il < 0 || il >= c->n_layer) { fprintf(stderr, "oracle_load: unknown tensor '%s'\n", name.c_str()); delete c; return nullptr; }
            Layer &L = c->layers[il];
            const std::string s = name.substr(p + 1);
            if (s == "attention.self.query.weight") mat(L.q_w); else if (s == "attention.self.query.bias") vec(L.q_b);
            else if (s == "attention.self.key.weight") mat(L.k_w); else if (s == "attention.self.key.bias") vec(L.k_b);
            else if (s == "attention.self.value.weight") mat(L.v_w); else if (s == "attention.self.value.bias") vec(L.v_b);
            else if (s == "attention.output.dense.weight") mat(L.o_w); else if (s == "attention.output.dense.bias") vec(L.o_b);
            else if (s == "attention.output.LayerNorm.weight") vec(L.ln_att_w); else if (s == "attention.output.LayerNorm.bias") vec(L.ln_att_b);
            else if (s == "intermediate.dense.weight") mat(L.ff_i_w); else if (s == "intermediate.dense.bias") vec(L.ff_i_b);
            else if (s == "output.dense.weight") mat(L.ff_o_w); else if (s == "output.dense.bias") vec(L.ff_o_b);
            else if (s == "output.LayerNorm.weight") vec(L.ln_out_w); else if (s == "output.LayerNorm.bias") vec(L.ln_out_b);
            else { fprintf(stderr, "oracle_load: unknown tensor '%s'\n", name.c_str()); delete c; return nullptr; }
        } else { fprintf(stderr, "oracle_load: unknown tensor '%s'\n", name.c_str()); delete c; return nullptr; }
    }
    (void)H;
    c->weights_loaded = true;
    return c;
}

void oracle_free(oracle_ctx *c) { delete c; }
int32_t oracle_n_embd(oracle_ctx *c) { return c->n_embd; }
int32_t oracle_n_max_tokens(oracle_ctx *c) { return c->n_max_tokens; }
int32_t oracle_n_layer(oracle_ctx *c) { return c->n_layer; }
int32_t oracle_ftype(oracle_ctx *c) { return c->ftype; }

const char *oracle_vocab_id_to_token(oracle_ctx *c, int32_t id) {   // bert.cpp:121-134
    auto it = c->vocab.id_to_token.find(id);
    if (it != c->vocab.id_to_token.end()) return it->second.c_str();
    it = c->vocab.id_to_subword_token.find(id);
    if (it != c->vocab.id_to_subword_token.end()) return it->second.c_str();
    return "[UNK TOKEN from bert_vocab]";
}

void oracle_tokenize(oracle_ctx *c, const char *text, int32_t *tokens, int32_t *n_tokens, int32_t n_max_tokens) {
    tokenize(c->vocab, text, tokens, n_tokens, n_max_tokens, /*quiet=*/true);
}

// One sentence.  out: [H].  hidden (optional): [(L+1)][N][H] — after the embedding LayerNorm and
// after every encoder layer.  Returns 0 on success.
int oracle_eval(oracle_ctx *c, int mode, int n_threads, const int32_t *tokens, int32_t N, float *out, float *hidden) {
    if (!c->weights_loaded) return -1;
    if (N > c->n_max_tokens || N <= 0) { fprintf(stderr, "Too many tokens, maximum is %d\n", c->n_max_tokens); return -2; }
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    const int H = c->n_embd, I = c->n_intermediate, nh = c->n_head, dh = H / nh;
    std::vector<float> x((size_t)N * H), tmp(H);
    // embeddings: word + token_type(0) + position   (bert.cpp:796-803)
    for (int t = 0; t < N; ++t) {
        float *r = x.data() + (size_t)t * H;
        if (tokens[t] < 0 || tokens[t] >= c->n_vocab) return -3;
        get_row(c->word_emb, tokens[t], r);
        get_row(c->type_emb, 0, tmp.data());
        for (int i = 0; i < H; ++i) r[i] = tmp[i] + r[i];
        get_row(c->pos_emb, t, tmp.data());
        for (int i = 0; i < H; ++i) r[i] = tmp[i] + r[i];
    }
    layer_norm(x.data(), N, H, c->ln_e_w, c->ln_e_b);
    if (hidden) memcpy(hidden, x.data(), sizeof(float) * N * H);

    std::vector<float> q((size_t)N * H), k((size_t)N * H), v((size_t)N * H), ctxv((size_t)N * H), y((size_t)N * H),
        ff((size_t)N * I);
    const float scale = 1.0f / sqrtf((float)dh);
    const Tables *tb = mode == 0 ? &tables() : nullptr;
    for (int il = 0; il < c->n_layer; ++il) {
        const Layer &L = c->layers[il];
        linear(L.q_w, L.q_b, x.data(), N, q.data(), mode);      // bert.cpp:822-841
        linear(L.k_w, L.k_b, x.data(), N, k.data(), mode);
        linear(L.v_w, L.v_b, x.data(), N, v.data(), mode);
        // attention per head (f32 x f32 mat-muls), bert.cpp:843-856
#pragma omp parallel for schedule(static) collapse(2)
        for (int h = 0; h < nh; ++h)
            for (int i = 0; i < N; ++i) {
                std::vector<float> s(N);
                const float *qi = q.data() + (size_t)i * H + h * dh;
                float mx = -INFINITY;
                for (int j = 0; j < N; ++j) {
                    const float *kj = k.data() + (size_t)j * H + h * dh;
                    float a = 0.f;
                    for (int e = 0; e < dh; ++e) a += kj[e] * qi[e];
                    a *= scale;
                    s[j] = a; mx = std::max(mx, a);
                }
                double sum = 0.0;
                for (int j = 0; j < N; ++j) {
                    float val;
                    if (tb) val = h2f(tb->expt[f2h(s[j] - mx)]);          // ggml_soft_max: fp16 exp table
                    else val = expf(s[j] - mx);
                    s[j] = val; sum += (double)val;
                }
                const float inv = (float)(1.0 / sum);
                for (int j = 0; j < N; ++j) s[j] *= inv;
                float *o = ctxv.data() + (size_t)i * H + h * dh;
                for (int e = 0; e < dh; ++e) {
                    float a = 0.f;
                    for (int j = 0; j < N; ++j) a += v[(size_t)j * H + h * dh + e] * s[j];
                    o[e] = a;
                }
            }
        // attention output + residual + LN  (bert.cpp:859-875)
        linear(L.o_w, L.o_b, ctxv.data(), N, y.data(), mode);
        for (size_t i = 0; i < y.size(); ++i) y[i] = y[i] + x[i];
        layer_norm(y.data(), N, H, L.ln_att_w, L.ln_att_b);
        // FFN  (bert.cpp:878-901)
        linear(L.ff_i_w, L.ff_i_b, y.data(), N, ff.data(), mode);
        if (tb) for (size_t i = 0; i < ff.size(); ++i) ff[i] = h2f(tb->gelu[f2h(ff[i])]);   // fp16 GELU table
        else for (size_t i = 0; i < ff.size(); ++i) ff[i] = gelu_f32(ff[i]);
        linear(L.ff_o_w, L.ff_o_b, ff.data(), N, x.data(), mode);
        for (size_t i = 0; i < x.size(); ++i) x[i] = y[i] + x[i];
        layer_norm(x.data(), N, H, L.ln_out_w, L.ln_out_b);
        if (hidden) memcpy(hidden + (size_t)(il + 1) * N * H, x.data(), sizeof(float) * N * H);
    }
    // mean pool (mat-vec with a 1/N vector) + L2 normalise  (bert.cpp:904-913)
    const float invn = 1.0f / N;
    double len2 = 0.0;
    for (int i = 0; i < H; ++i) {
        float a = 0.f;
        for (int t = 0; t < N; ++t) a += x[(size_t)t * H + i] * invn;
        out[i] = a; len2 += (double)(a * a);
    }
    const float sc = 1.0f / sqrtf((float)len2);
    for (int i = 0; i < H; ++i) out[i] *= sc;
    return 0;
}

// Sequential loop over sentences, exactly like bert.cpp:750 (threads are intra-op, like ggml).
int oracle_eval_batch(oracle_ctx *c, int mode, int n_threads, int32_t B, const int32_t *const *tokens,
                      const int32_t *n_tokens, float *const *out) {
    for (int b = 0; b < B; ++b) {
        int r = oracle_eval(c, mode, n_threads, tokens[b], n_tokens[b], out[b], nullptr);
        if (r) return r;
    }
    return 0;
}

uint16_t oracle_f2h_soft(float f) { return f2h_soft(f); }
float oracle_h2f_soft(uint16_t h) { return h2f_soft(h); }
uint16_t oracle_f2h(float f) { return f2h(f); }

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
