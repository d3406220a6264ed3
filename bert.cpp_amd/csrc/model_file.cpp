// model_file.cpp — see model_file.h.
#include "model_file.h"

#include <cstdio>
#include <cmath>
#include <cstring>

namespace bert_hip {

size_t wtype_row_bytes(int32_t type, int64_t ne0) {
    switch (type) {
        case W_F32: return (size_t)ne0 * 4;
        case W_F16: return (size_t)ne0 * 2;
        case W_Q4_0: return (size_t)(ne0 / 32) * 18;
        case W_Q4_1: return (size_t)(ne0 / 32) * 20;
    }
    return 0;
}

namespace {

struct Cursor {
    const uint8_t *p, *end;
    bool take(void *dst, size_t n) {
        if ((size_t)(end - p) < n) return false;
        memcpy(dst, p, n);
        p += n;
        return true;
    }
};

// The tensor set bert.cpp creates (reference bert.cpp:484-555): name -> expected (ne0, ne1, is_2d)
struct Expect { int64_t ne0, ne1; bool two_d; };
std::map<std::string, Expect> expected_tensors(const HParams &h) {
    std::map<std::string, Expect> m;
    const int64_t H = h.n_embd, I = h.n_intermediate;
    m["embeddings.word_embeddings.weight"] = {H, h.n_vocab, true};
    m["embeddings.token_type_embeddings.weight"] = {H, 2, true};
    m["embeddings.position_embeddings.weight"] = {H, h.n_max_tokens, true};
    m["embeddings.LayerNorm.weight"] = {H, 1, false};
    m["embeddings.LayerNorm.bias"] = {H, 1, false};
    for (int i = 0; i < h.n_layer; ++i) {
        const std::string p = "encoder.layer." + std::to_string(i) + ".";
        for (const char *qkv : {"query", "key", "value"}) {
            m[p + "attention.self." + qkv + ".weight"] = {H, H, true};
            m[p + "attention.self." + qkv + ".bias"] = {H, 1, false};
        }
        m[p + "attention.output.dense.weight"] = {H, H, true};
        m[p + "attention.output.dense.bias"] = {H, 1, false};
        m[p + "attention.output.LayerNorm.weight"] = {H, 1, false};
        m[p + "attention.output.LayerNorm.bias"] = {H, 1, false};
        m[p + "intermediate.dense.weight"] = {H, I, true};
        m[p + "intermediate.dense.bias"] = {I, 1, false};
        m[p + "output.dense.weight"] = {I, H, true};
        m[p + "output.dense.bias"] = {H, 1, false};
        m[p + "output.LayerNorm.weight"] = {H, 1, false};
        m[p + "output.LayerNorm.bias"] = {H, 1, false};
    }
    return m;
}

}  // namespace

bool ModelFile::load(const char *fname, bool vocab_only, std::string &err) {
    FILE *f = fopen(fname, "rb");
    if (!f) { err = std::string("failed to open '") + fname + "'"; return false; }
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 32) { fclose(f); err = std::string("invalid model file '") + fname + "' (too short)"; return false; }
    if (vocab_only) {
        // header + vocab are at the front; read a bounded prefix (vocab of 30522 entries ~ 350 KB)
        const long want = sz < (64L << 20) ? sz : (64L << 20);
        blob.resize((size_t)want);
    } else {
        blob.resize((size_t)sz);
    }
    const size_t got = fread(blob.data(), 1, blob.size(), f);
    fclose(f);
    if (got != blob.size()) { err = std::string("short read on '") + fname + "'"; return false; }

    Cursor c{blob.data(), blob.data() + blob.size()};
    uint32_t magic = 0;
    c.take(&magic, 4);
    if (magic != 0x67676d6c) { err = std::string("invalid model file '") + fname + "' (bad magic)"; return false; }
    int32_t hv[7];
    if (!c.take(hv, sizeof(hv))) { err = "truncated header"; return false; }
    hp.n_vocab = hv[0]; hp.n_max_tokens = hv[1]; hp.n_embd = hv[2]; hp.n_intermediate = hv[3];
    hp.n_head = hv[4]; hp.n_layer = hv[5]; hp.f16 = hv[6];
    if (hp.n_vocab <= 0 || hp.n_max_tokens <= 0 || hp.n_embd <= 0 || hp.n_intermediate <= 0 || hp.n_head <= 0 ||
        hp.n_layer < 0 || hp.n_embd % hp.n_head != 0 || hp.n_vocab > (1 << 24) || hp.n_layer > 4096) {
        err = std::string("invalid model file '") + fname + "' (implausible hyper-parameters)";
        return false;
    }
    vocab.clear();
    vocab.reserve(hp.n_vocab);
    for (int i = 0; i < hp.n_vocab; ++i) {
        uint32_t len = 0;
        if (!c.take(&len, 4) || (size_t)(c.end - c.p) < len) { err = "truncated vocab"; return false; }
        vocab.emplace_back((const char *)c.p, len);
        c.p += len;
    }
    if (vocab_only) return true;

    if (hp.f16 < 0 || hp.f16 > 3) {
        err = std::string("invalid model file '") + fname + "' (bad f16 value " + std::to_string(hp.f16) + ")";
        return false;
    }
    const auto expect = expected_tensors(hp);
    tensors.clear();
    total_tensor_bytes = 0;
    // q4 files exist in two block layouts and the records carry no byte counts: the current one (18-byte q4_0 / 20-byte
    // q4_1 blocks: f16 scales, nibbles j | j+16 per byte) and the one of early-2023 ggml (20 / 24 bytes: f32 scales, nibbles
    // 2j | 2j+1 — the sizes the reference's README prints are these, README.md:105-107).  Every expected tensor appears
    // exactly once, so the length of the tensor section tells the two apart.
    legacy_q4 = false;
    if (hp.f16 == W_Q4_0 || hp.f16 == W_Q4_1) {
        size_t cur = 0, leg = 0;
        for (const auto &kv : expect) {
            const Expect &e = kv.second;
            const size_t hdr = 12 + 4 * (e.two_d ? 2 : 1) + kv.first.size();
            if (e.two_d && e.ne0 % 32 == 0) {
                const size_t nblk = (size_t)(e.ne0 / 32) * e.ne1;
                cur += hdr + nblk * (hp.f16 == W_Q4_0 ? 18 : 20);
                leg += hdr + nblk * (hp.f16 == W_Q4_0 ? 20 : 24);
            } else {
                cur += hdr + (size_t)e.ne0 * e.ne1 * 4;
                leg += hdr + (size_t)e.ne0 * e.ne1 * 4;
            }
        }
        const size_t have = (size_t)(c.end - c.p);
        if (have == leg && have != cur) legacy_q4 = true;
    }
    converted.clear();
    while (c.p < c.end) {
        int32_t n_dims = 0, name_len = 0, ftype = 0;
        if (!c.take(&n_dims, 4) || !c.take(&name_len, 4) || !c.take(&ftype, 4)) { err = "truncated tensor header"; return false; }
        if (n_dims < 1 || n_dims > 2 || name_len <= 0 || name_len > 4096) { err = "corrupt tensor header"; return false; }
        int64_t ne[2] = {1, 1}, nel = 1;
        for (int i = 0; i < n_dims; ++i) {
            int32_t v = 0;
            if (!c.take(&v, 4) || v <= 0) { err = "corrupt tensor dims"; return false; }
            ne[i] = v;
            nel *= v;
        }
        if ((size_t)(c.end - c.p) < (size_t)name_len) { err = "truncated tensor name"; return false; }
        std::string name((const char *)c.p, name_len);
        c.p += name_len;
        auto ex = expect.find(name);
        if (ex == expect.end()) { err = "unknown tensor '" + name + "' in model file"; return false; }
        if (ex->second.ne0 * ex->second.ne1 != nel) { err = "tensor '" + name + "' has wrong size in model file"; return false; }
        if (ex->second.ne0 != ne[0] || ex->second.ne1 != ne[1]) {
            err = "tensor '" + name + "' has wrong shape in model file: got [" + std::to_string(ne[0]) + ", " +
                  std::to_string(ne[1]) + "], expected [" + std::to_string(ex->second.ne0) + ", " +
                  std::to_string(ex->second.ne1) + "]";
            return false;
        }
        if (ftype < 0 || ftype > 3) { err = "unknown ftype " + std::to_string(ftype) + " in model file"; return false; }
        // the reference allocates 2-D tensors in the file-wide type and 1-D tensors in f32 and
        // rejects a record whose byte size disagrees (bert.cpp:652-658)
        const int32_t want_type = ex->second.two_d ? hp.f16 : (int32_t)W_F32;
        if (ftype != want_type) {
            err = "tensor '" + name + "' has wrong size in model file: stored type " + std::to_string(ftype) +
                  ", expected type " + std::to_string(want_type);
            return false;
        }
        if ((ftype == W_Q4_0 || ftype == W_Q4_1) && ne[0] % 32 != 0) { err = "tensor '" + name + "': row length not a multiple of 32"; return false; }
        const size_t nbytes = wtype_row_bytes(ftype, ne[0]) * (size_t)ne[1];
        const bool legacy = legacy_q4 && (ftype == W_Q4_0 || ftype == W_Q4_1);
        const size_t file_bytes = legacy ? (size_t)(ne[0] / 32) * ne[1] * (ftype == W_Q4_0 ? 20 : 24) : nbytes;
        if ((size_t)(c.end - c.p) < file_bytes) { err = "tensor '" + name + "' is truncated"; return false; }
        HostTensor t;
        t.type = ftype; t.n_dims = n_dims; t.ne0 = ne[0]; t.ne1 = ne[1]; t.data = c.p; t.nbytes = nbytes;
        if (legacy) {
            // re-block into the current layout (same q's; the f32 scale / minimum are rounded to f16, which is what the
            // current format stores): everything downstream sees one layout
            converted.emplace_back(nbytes);
            uint8_t *dst = converted.back().data();
            const int nsc = ftype == W_Q4_0 ? 1 : 2, lbs = 4 * nsc + 16, nbs = 2 * nsc + 16;
            const size_t nblk = (size_t)(ne[0] / 32) * ne[1];
            for (size_t b = 0; b < nblk; ++b) {
                const uint8_t *src = c.p + b * lbs;
                uint8_t *out = dst + b * nbs;
                for (int k = 0; k < nsc; ++k) {
                    float v;
                    memcpy(&v, src + 4 * k, 4);
                    const _Float16 h = (_Float16)v;
                    // (a finite scale above f16's range would silently become inf: refuse the file instead.  A scale below
                    // f16's smallest subnormal flushes to zero — the block's values change by less than 2.4e-7 — and loads)
                    const float back = (float)h;
                    if (std::isfinite(v) && !std::isfinite(back)) {
                        err = "tensor '" + name + "': a legacy q4 block scale (" + std::to_string(v) + ") does not fit f16";
                        return false;
                    }
                    memcpy(out + 2 * k, &h, 2);
                }
                uint8_t el[32];
                for (int j = 0; j < 16; ++j) { el[2 * j] = src[4 * nsc + j] & 0x0F; el[2 * j + 1] = src[4 * nsc + j] >> 4; }
                for (int j = 0; j < 16; ++j) out[2 * nsc + j] = (uint8_t)(el[j] | (el[j + 16] << 4));
            }
            t.data = dst;
        }
        tensors[name] = t;
        total_tensor_bytes += nbytes;
        c.p += file_bytes;
    }
    // the reference leaves missing tensors uninitialised; fail loudly instead
    for (const auto &kv : expect)
        if (!tensors.count(kv.first)) { err = "tensor '" + kv.first + "' is missing from model file"; return false; }
    return true;
}

}  // namespace bert_hip
