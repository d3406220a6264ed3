// model_file.h — parser of bert.cpp's on-disk model format.
//
// Format reader contract: reference bert.cpp:342-669 (magic, 7 x i32 hparams, vocab records,
// tensor records until EOF with per-tensor ftype, name-keyed lookup, shape and byte-size checks).
// The reference reads tensors straight into a ggml arena; here they land in host byte buffers
// that the engine then repacks into its HBM layouts (engine.hip).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace bert_hip {

enum WType : int32_t { W_F32 = 0, W_F16 = 1, W_Q4_0 = 2, W_Q4_1 = 3 };

struct HParams {             // reference bert.cpp:18-27, file order :361-367
    int32_t n_vocab = 0, n_max_tokens = 0, n_embd = 0, n_intermediate = 0, n_head = 0, n_layer = 0, f16 = 0;
};

struct HostTensor {
    int32_t type = 0;        // WType of the stored bytes
    int32_t n_dims = 0;
    int64_t ne0 = 1, ne1 = 1;   // ne0 = contiguous (in-features), ne1 = rows (out-features)
    const uint8_t *data = nullptr;
    size_t nbytes = 0;
};

struct ModelFile {
    HParams hp;
    std::vector<std::string> vocab;
    std::map<std::string, HostTensor> tensors;
    std::vector<uint8_t> blob;   // whole file; HostTensor::data points into it
    size_t total_tensor_bytes = 0;
    bool legacy_q4 = false;                          // the file uses the 20 / 24-byte q4 blocks of early-2023 ggml
    std::vector<std::vector<uint8_t>> converted;     // their tensors, re-blocked into the current layout

    // Returns false and fills `err` on any malformed input.  vocab_only stops after the vocab.
    bool load(const char *fname, bool vocab_only, std::string &err);
    const HostTensor *find(const std::string &name) const {
        auto it = tensors.find(name);
        return it == tensors.end() ? nullptr : &it->second;
    }
};

size_t wtype_row_bytes(int32_t type, int64_t ne0);

}  // namespace bert_hip
