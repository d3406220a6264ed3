// bert-quantize — rewrites a bert.cpp model file with its 2-D "*weight" tensors as q4_0 / q4_1 blocks.
//
//   bert-quantize model-f32-or-f16.bin model-quant.bin type      type = 2 (q4_0) | 3 (q4_1)
//
// Same command line, selection rule and output bytes as the reference tool (models/quantize.cpp:27-268: header
// and vocab copied, the ftype hparam replaced by `type` :93, a tensor is quantized iff it has 2 dims and its name
// ends in "weight" :154-167, source f32 or f16 only :170-173, everything else copied raw), with the block
// quantizers of the ggml API that file was written against restated here:
//   q4_0 block (18 B): d = (element of largest magnitude, signed) / -8 as f16, 32 nibbles min(15, (int8)(x/d + 8.5)),
//                      byte j = elem j | elem j+16 << 4
//   q4_1 block (20 B): d = (max - min) / 15 as f16, m = min as f16, nibbles min(15, (int8)((x - min)/d + 0.5))
// The same arithmetic is what bert.cpp_amd/ggml_file.py (quantize_q4_0/1) does in NumPy; tests/test_tools.py
// checks the two byte for byte.  Streaming: one tensor in memory at a time, blocks quantized on all host cores.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kBlock = 32;

float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff, bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {                                   // subnormal: renormalise
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400));
            bits = sign | (uint32_t)(112 - e) << 23 | (man & 0x3ff) << 13;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | man << 13;
    } else {
        bits = sign | (exp + 112) << 23 | man << 13;
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

uint16_t f32_to_f16(float f) {                     // round to nearest even, like the F16C conversion ggml uses
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000);
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return sign | (x > 0x7f800000u ? 0x7e00 : 0x7c00);
    if (x >= 0x477ff000u) return sign | 0x7c00;    // rounds to >= 65520 -> inf
    if (x < 0x33000001u) return sign;              // < 2^-25 (or exactly 2^-25: ties to even 0)
    if (x < 0x38800000u) {                         // subnormal half
        const int shift = 126 - (int)(x >> 23);    // 14..24
        uint32_t man = (x & 0x7fffffu) | 0x800000u;
        uint32_t q = man >> shift, rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1))) ++q;
        return sign | (uint16_t)q;
    }
    uint32_t q = (x - 0x38000000u) >> 13, rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1))) ++q;
    return sign | (uint16_t)q;
}

void quantize_block_q4_0(const float *x, uint8_t *out, int64_t *hist) {
    float amax = 0.0f, vmax = 0.0f;
    for (int j = 0; j < kBlock; ++j)
        if (std::fabs(x[j]) > amax) { amax = std::fabs(x[j]); vmax = x[j]; }
    const float d = vmax / -8.0f;
    const float inv = d != 0.0f ? 1.0f / d : 0.0f;
    const uint16_t dh = f32_to_f16(d);
    memcpy(out, &dh, 2);
    uint8_t q[kBlock];
    for (int j = 0; j < kBlock; ++j) {
        q[j] = (uint8_t)std::min(15, (int)(int8_t)(x[j] * inv + 8.5f));
        ++hist[q[j]];
    }
    for (int j = 0; j < kBlock / 2; ++j) out[2 + j] = (uint8_t)(q[j] | q[j + kBlock / 2] << 4);
}

void quantize_block_q4_1(const float *x, uint8_t *out, int64_t *hist) {
    float lo = x[0], hi = x[0];
    for (int j = 1; j < kBlock; ++j) { lo = std::min(lo, x[j]); hi = std::max(hi, x[j]); }
    const float d = (hi - lo) / 15.0f;
    const float inv = d != 0.0f ? 1.0f / d : 0.0f;
    const uint16_t dh = f32_to_f16(d), mh = f32_to_f16(lo);
    memcpy(out, &dh, 2);
    memcpy(out + 2, &mh, 2);
    uint8_t q[kBlock];
    for (int j = 0; j < kBlock; ++j) {
        q[j] = (uint8_t)std::min(15, (int)(int8_t)((x[j] - lo) * inv + 0.5f));
        ++hist[q[j]];
    }
    for (int j = 0; j < kBlock / 2; ++j) out[4 + j] = (uint8_t)(q[j] | q[j + kBlock / 2] << 4);
}

struct File {
    FILE *f = nullptr;
    ~File() { if (f) fclose(f); }
};

bool rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n; }
bool wr(FILE *f, const void *p, size_t n) { return fwrite(p, 1, n, f) == n; }

bool ends_with(const std::string &s, const char *suffix) {
    const size_t n = strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

bool quantize_model(const char *in_path, const char *out_path, int32_t itype) {
    if (itype != 2 && itype != 3) {
        fprintf(stderr, "bert-quantize: invalid quantization type %d (2 = q4_0, 3 = q4_1)\n", itype);
        return false;
    }
    File in, out;
    if (!(in.f = fopen(in_path, "rb"))) { fprintf(stderr, "bert-quantize: failed to open '%s' for reading\n", in_path); return false; }
    if (!(out.f = fopen(out_path, "wb"))) { fprintf(stderr, "bert-quantize: failed to open '%s' for writing\n", out_path); return false; }

    uint32_t magic = 0;
    if (!rd(in.f, &magic, 4) || magic != 0x67676d6cu) { fprintf(stderr, "bert-quantize: invalid model file '%s' (bad magic)\n", in_path); return false; }
    wr(out.f, &magic, 4);

    int32_t hp[7];
    if (!rd(in.f, hp, sizeof(hp))) { fprintf(stderr, "bert-quantize: truncated header\n"); return false; }
    static const char *hp_names[7] = {"n_vocab", "n_max_tokens", "n_embd", "n_intermediate", "n_head", "n_layer", "f16"};
    for (int i = 0; i < 7; ++i) printf("bert-quantize: %-14s = %d\n", hp_names[i], hp[i]);
    const int32_t n_vocab = hp[0];
    hp[6] = itype;
    wr(out.f, hp, sizeof(hp));

    std::string word;
    for (int32_t i = 0; i < n_vocab; ++i) {
        uint32_t len = 0;
        if (!rd(in.f, &len, 4) || len > (1u << 20)) { fprintf(stderr, "bert-quantize: bad vocab record %d\n", i); return false; }
        word.resize(len);
        if (len && !rd(in.f, &word[0], len)) { fprintf(stderr, "bert-quantize: truncated vocab\n"); return false; }
        wr(out.f, &len, 4);
        wr(out.f, word.data(), len);
    }

    static const char *type_names[4] = {"f32", "f16", "q4_0", "q4_1"};
    const size_t block_bytes = itype == 2 ? 18 : 20;
    const unsigned n_workers = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
    size_t bytes_before = 0, bytes_after = 0;
    std::vector<int64_t> hist_all(16, 0);
    std::vector<uint8_t> raw, packed;
    std::vector<float> f32;

    for (;;) {
        int32_t n_dims = 0, name_len = 0, ttype = 0;
        if (!rd(in.f, &n_dims, 4)) break;                       // clean EOF ends the tensor list
        if (!rd(in.f, &name_len, 4) || !rd(in.f, &ttype, 4) || n_dims < 1 || n_dims > 2 || name_len < 0 || name_len > 4096) {
            fprintf(stderr, "bert-quantize: malformed tensor record\n");
            return false;
        }
        int32_t ne[2] = {1, 1};
        int64_t n_elem = 1;
        for (int i = 0; i < n_dims; ++i) {
            if (!rd(in.f, &ne[i], 4) || ne[i] <= 0) { fprintf(stderr, "bert-quantize: malformed tensor shape\n"); return false; }
            n_elem *= ne[i];
        }
        std::string name((size_t)name_len, '\0');
        if (name_len && !rd(in.f, &name[0], (size_t)name_len)) { fprintf(stderr, "bert-quantize: truncated tensor name\n"); return false; }
        if (ttype < 0 || ttype > 3) { fprintf(stderr, "bert-quantize: tensor '%s' has unknown type %d\n", name.c_str(), ttype); return false; }
        printf("%48s - [%5d, %5d], type = %6s ", name.c_str(), ne[0], ne[1], type_names[ttype]);

        const bool quantize = n_dims == 2 && ends_with(name, "weight");
        size_t src_bytes;
        if (ttype == 0) src_bytes = (size_t)n_elem * 4;
        else if (ttype == 1) src_bytes = (size_t)n_elem * 2;
        else src_bytes = (size_t)(n_elem / kBlock) * (ttype == 2 ? 18 : 20);
        if (quantize && ttype > 1) {
            fprintf(stderr, "\nbert-quantize: unsupported source type %s for quantization of '%s'\n", type_names[ttype], name.c_str());
            return false;
        }
        if (quantize && ne[0] % kBlock != 0) {
            fprintf(stderr, "\nbert-quantize: row length %d of '%s' is not a multiple of %d\n", ne[0], name.c_str(), kBlock);
            return false;
        }
        raw.resize(src_bytes);
        if (!rd(in.f, raw.data(), src_bytes)) { fprintf(stderr, "\nbert-quantize: truncated data of '%s'\n", name.c_str()); return false; }

        const int32_t out_type = quantize ? itype : ttype;
        wr(out.f, &n_dims, 4);
        wr(out.f, &name_len, 4);
        wr(out.f, &out_type, 4);
        wr(out.f, ne, 4 * (size_t)n_dims);
        wr(out.f, name.data(), (size_t)name_len);

        if (!quantize) {
            wr(out.f, raw.data(), src_bytes);
            bytes_before += src_bytes;
            bytes_after += src_bytes;
            printf("size = %8.3f MB\n", src_bytes / 1048576.0);
            continue;
        }

        const float *src = (const float *)raw.data();
        if (ttype == 1) {
            f32.resize((size_t)n_elem);
            const uint16_t *h = (const uint16_t *)raw.data();
            for (int64_t i = 0; i < n_elem; ++i) f32[(size_t)i] = f16_to_f32(h[i]);
            src = f32.data();
        }
        const int64_t n_blocks = n_elem / kBlock;
        packed.resize((size_t)n_blocks * block_bytes);
        std::vector<std::vector<int64_t>> hists(n_workers, std::vector<int64_t>(16, 0));
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < n_workers; ++w) {
            pool.emplace_back([&, w] {
                const int64_t b0 = n_blocks * w / n_workers, b1 = n_blocks * (w + 1) / n_workers;
                for (int64_t b = b0; b < b1; ++b) {
                    if (itype == 2) quantize_block_q4_0(src + b * kBlock, packed.data() + (size_t)b * 18, hists[w].data());
                    else quantize_block_q4_1(src + b * kBlock, packed.data() + (size_t)b * 20, hists[w].data());
                }
            });
        }
        for (auto &t : pool) t.join();
        wr(out.f, packed.data(), packed.size());
        bytes_before += (size_t)n_elem * 4;
        bytes_after += packed.size();

        printf("quantizing .. size = %8.2f MB -> %8.2f MB | hist: ", n_elem * 4 / 1048576.0, packed.size() / 1048576.0);
        for (int i = 0; i < 16; ++i) {
            int64_t c = 0;
            for (auto &h : hists) c += h[(size_t)i];
            hist_all[(size_t)i] += c;
            printf("%5.3f ", (double)c / (double)n_elem);
        }
        printf("\n");
    }
    if (ferror(in.f) || ferror(out.f) || fflush(out.f) != 0) { fprintf(stderr, "bert-quantize: I/O error\n"); return false; }

    printf("bert-quantize: model size  = %8.2f MB\n", bytes_before / 1048576.0);
    printf("bert-quantize: quant size  = %8.2f MB\n", bytes_after / 1048576.0);
    int64_t total = 0;
    for (int64_t c : hist_all) total += c;
    if (total > 0) {
        printf("bert-quantize: hist: ");
        for (int64_t c : hist_all) printf("%5.3f ", (double)c / (double)total);
        printf("\n");
    }
    return true;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: %s model-f32.bin model-quant.bin type\n", argv[0]);
        fprintf(stderr, "  type = 2 - q4_0\n");
        fprintf(stderr, "  type = 3 - q4_1\n");
        return 1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (!quantize_model(argv[1], argv[2], atoi(argv[3]))) {
        fprintf(stderr, "bert-quantize: failed to quantize model from '%s'\n", argv[1]);
        return 1;
    }
    printf("\nbert-quantize: quantize time = %8.2f ms\n",
           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return 0;
}
