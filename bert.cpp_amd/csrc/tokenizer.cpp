// tokenizer.cpp — see tokenizer.h.  Behavioural contract: reference bert.cpp:199-325, 379-403.
#include "tokenizer.h"

#include <cstdio>
#include <cstring>

namespace bert_hip {

namespace {

// Sequence length implied by the high nibble of a lead byte (reference bert.cpp:199-204): bytes
// 0x00-0xBF count as 1 (continuation bytes included), 0xC0-0xDF as 2, 0xE0-0xEF as 3, 0xF0+ as 4.
inline size_t lead_len(unsigned char c) {
    return c < 0xC0 ? 1 : c < 0xE0 ? 2 : c < 0xF0 ? 3 : 4;
}

// Second byte (after 0xC3) of the 52 accented letters the reference strips -> ASCII replacement,
// 0 = not mapped.  U+00C0..U+00FF encode as C3 80..C3 BF.   (reference bert.cpp:209-219)
struct AccentTable {
    char map[64];
    AccentTable() {
        memset(map, 0, sizeof(map));
        auto set = [&](int first, int last, char r) { for (int c = first; c <= last; ++c) map[c - 0xC0] = r; };
        set(0xC0, 0xC5, 'A'); set(0xE0, 0xE5, 'a');      // À-Å  à-å
        set(0xC8, 0xCB, 'E'); set(0xE8, 0xEB, 'e');      // È-Ë  è-ë
        set(0xCC, 0xCF, 'I'); set(0xEC, 0xEF, 'i');      // Ì-Ï  ì-ï
        set(0xD2, 0xD6, 'O'); set(0xF2, 0xF6, 'o');      // Ò-Ö  ò-ö
        set(0xD9, 0xDC, 'U'); set(0xF9, 0xFC, 'u');      // Ù-Ü  ù-ü
        map[0xDD - 0xC0] = 'Y'; map[0xFD - 0xC0] = 'y';  // Ý ý
        map[0xC7 - 0xC0] = 'C'; map[0xE7 - 0xC0] = 'c';  // Ç ç
        map[0xD1 - 0xC0] = 'N'; map[0xF1 - 0xC0] = 'n';  // Ñ ñ
    }
};
const AccentTable kAccents;

// "C"-locale character classes of the reference's regex  [[:punct:]] | [[:alpha:]]+ | [[:digit:]]+
enum : uint8_t { C_SEP = 0, C_ALPHA = 1, C_DIGIT = 2, C_PUNCT = 3 };
struct ClassTable {
    uint8_t cls[256];
    ClassTable() {
        memset(cls, C_SEP, sizeof(cls));
        for (int c = 'a'; c <= 'z'; ++c) cls[c] = C_ALPHA;
        for (int c = 'A'; c <= 'Z'; ++c) cls[c] = C_ALPHA;   // can survive lower-casing, see normalize()
        for (int c = '0'; c <= '9'; ++c) cls[c] = C_DIGIT;
        for (int c = 33; c <= 47; ++c) cls[c] = C_PUNCT;
        for (int c = 58; c <= 64; ++c) cls[c] = C_PUNCT;
        for (int c = 91; c <= 96; ++c) cls[c] = C_PUNCT;
        for (int c = 123; c <= 126; ++c) cls[c] = C_PUNCT;
    }
};
const ClassTable kClasses;

// stripAccents + ASCII lower-casing, byte-exact with the reference on malformed UTF-8 as well:
// both passes advance by lead_len() of whatever byte they stand on.
void normalize(const char *text, std::string &out) {
    const size_t n = strlen(text);
    out.clear();
    out.reserve(n);
    for (size_t i = 0; i < n;) {
        const unsigned char c = (unsigned char)text[i];
        const size_t len = lead_len(c);
        if (len == 2 && c == 0xC3 && i + 1 < n) {
            const unsigned char c2 = (unsigned char)text[i + 1];
            if (c2 >= 0x80 && c2 <= 0xBF && kAccents.map[c2 - 0x80]) {
                out.push_back(kAccents.map[c2 - 0x80]);
                i += 2;
                continue;
            }
        }
        const size_t take = i + len <= n ? len : n - i;
        out.append(text + i, take);
        i += len;
    }
    for (size_t i = 0; i < out.size(); i += lead_len((unsigned char)out[i])) {
        const char c = out[i];
        if (c >= 'A' && c <= 'Z') out[i] = (char)(c - 'A' + 'a');
    }
}

}  // namespace

void Tokenizer::build(std::vector<std::string> &&words) {
    words_ = std::move(words);
    const size_t n = words_.size();
    has_token_.assign(n, 0);
    has_subword_.assign(n, 0);
    token_to_id_.clear();
    subword_to_id_.clear();
    token_to_id_.reserve(n * 2);
    max_token_len_ = max_subword_len_ = 0;
    for (size_t i = 0; i < n; ++i) {
        const std::string &w = words_[i];
        if (w.size() >= 2 && w[0] == '#' && w[1] == '#') {
            // later duplicates overwrite earlier ones (reference uses operator[] assignment)
            subword_to_id_[std::string_view(w).substr(2)] = (int32_t)i;
            has_subword_[i] = 1;
            if (w.size() - 2 > max_subword_len_) max_subword_len_ = w.size() - 2;
        }
        // first occurrence wins; '##' pieces are ALSO whole-word entries under their full spelling
        if (token_to_id_.emplace(std::string_view(w), (int32_t)i).second) {
            has_token_[i] = 1;
            if (w.size() > max_token_len_) max_token_len_ = w.size();
        }
    }
}

const char *Tokenizer::id_to_token(int32_t id) const {
    if (id >= 0 && (size_t)id < words_.size() && (has_token_[id] || has_subword_[id])) return words_[id].c_str();
    return "[UNK TOKEN from bert_vocab]";
}

void Tokenizer::tokenize(const char *text, int32_t *tokens, int32_t *n_tokens, int32_t n_max_tokens) const {
    thread_local std::string str;
    normalize(text, str);
    const char *s = str.data();
    const size_t n = str.size();

    int32_t t = 0;
    tokens[t++] = 101;   // [CLS]

    size_t p = 0;
    while (p < n) {
        const uint8_t cls = kClasses.cls[(unsigned char)s[p]];
        if (cls == C_SEP) { ++p; continue; }
        size_t e = p + 1;
        if (cls != C_PUNCT)
            while (e < n && kClasses.cls[(unsigned char)s[e]] == cls) ++e;
        // word = s[p, e)
        const char *w = s + p;
        const size_t wn = e - p;
        p = e;
        if (t >= n_max_tokens - 1) break;   // every later word would hit the same check first

        size_t i = 0;
        const auto *map = &token_to_id_;
        size_t max_len = max_token_len_;
        while (i < wn) {
            if (t >= n_max_tokens - 1) break;
            size_t j = wn - i < max_len ? wn : i + max_len;   // longer candidates cannot be in the map
            bool found = false;
            for (; j > i; --j) {
                auto it = map->find(std::string_view(w + i, j - i));
                if (it != map->end()) {
                    tokens[t++] = it->second;
                    i = j;
                    found = true;
                    break;
                }
            }
            if (!found) {
                if (!quiet) fprintf(stderr, "bert_tokenize: unknown token '%c'\n", w[i]);
                ++i;
            }
            map = &subword_to_id_;
            max_len = max_subword_len_;
        }
    }
    tokens[t++] = 102;   // [SEP]
    *n_tokens = t;
}

}  // namespace bert_hip
