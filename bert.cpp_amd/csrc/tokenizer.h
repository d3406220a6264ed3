// tokenizer.h — host-side WordPiece tokenizer of the engine.
//
// Produces, bit for bit, the ids of the reference's bert_tokenize (reference bert.cpp:252-325,
// with stripAccents :206-238, bert_normalize_prompt :240-251, utf8_len :199-204 and the vocab
// maps built at :379-403), including its quirks: only 52 Latin-1 letters
// lose their accent, only ASCII is lower-cased, every other non-ASCII byte is a separator that
// emits nothing, an unmatched byte is skipped without emitting [UNK], and truncation leaves room
// for exactly one [SEP].  The implementation is different: a byte-class scanner instead of
// std::regex, and hashed string_view lookups bounded by the longest vocab entry instead of
// std::map<std::string> with substr copies.
#pragma once
#include <cstdint>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace bert_hip {

class Tokenizer {
public:
    // Vocab entries in id order (reference bert.cpp:383-401).
    void build(std::vector<std::string> &&words);

    // tokens must have room for n_max_tokens ids.
    void tokenize(const char *text, int32_t *tokens, int32_t *n_tokens, int32_t n_max_tokens) const;

    // reference bert.cpp:121-134
    const char *id_to_token(int32_t id) const;

    size_t size() const { return words_.size(); }
    bool quiet = false;   // suppress the per-byte "unknown token" stderr line of the reference

private:
    std::vector<std::string> words_;
    std::vector<uint8_t> has_token_, has_subword_;
    std::unordered_map<std::string_view, int32_t> token_to_id_, subword_to_id_;
    size_t max_token_len_ = 0, max_subword_len_ = 0;
};

}  // namespace bert_hip
