// tokenizer.hpp -- llama2.c tokenizer.bin reader + greedy best-score BPE merge, restating
// engine/src/tokenizer/bpe.rs (host-side string code).
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace rama_host {

struct Tokenizer {
    std::vector<std::string> vocab;
    std::vector<float> vocab_scores;
    std::unordered_map<std::string, size_t> word_token_map;
    size_t max_token_length = 0;

    // bpe.rs:19-45: u32 max_token_length, then per token: f32 score, i32 length, bytes.
    // A later duplicate string overwrites the map entry (HashMap::insert).
    static Tokenizer from_file(const std::string& path, size_t vocab_size) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("couldn't open " + path);
        Tokenizer t;
        uint32_t mtl = 0;
        f.read(reinterpret_cast<char*>(&mtl), 4);
        t.max_token_length = mtl;
        for (size_t idx = 0; idx < vocab_size; idx++) {
            float score; int32_t len;
            f.read(reinterpret_cast<char*>(&score), 4);
            f.read(reinterpret_cast<char*>(&len), 4);
            if (!f || len < 0) throw std::runtime_error("tokenizer file truncated");   // read_n assert, read.rs:37-42
            std::string s((size_t)len, '\0');
            f.read(s.data(), len);
            if (!f) throw std::runtime_error("tokenizer file truncated");
            t.vocab_scores.push_back(score);
            t.vocab.push_back(s);
            t.word_token_map[s] = idx;
        }
        return t;
    }

    // str::trim (bpe.rs:53) strips the code points with the Unicode White_Space property, not just ASCII blanks
    static bool is_white_space(uint32_t c) {
        return (c >= 0x09 && c <= 0x0D) || c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) ||
               c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
    }
    // the UTF-8 scalar that starts at text[i] (n = its length); malformed bytes count as one-byte characters that are no white space
    static uint32_t scalar_at(const std::string& text, size_t i, size_t end, size_t& n) {
        const unsigned char c = (unsigned char)text[i];
        n = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 1;
        if (i + n > end) { n = 1; return 0xFFFFFFFFu; }
        if (n == 1) return c < 0x80 ? c : 0xFFFFFFFFu;
        uint32_t v = c & (0xFF >> (n + 1));
        for (size_t k = 1; k < n; k++) {
            const unsigned char d = (unsigned char)text[i + k];
            if ((d >> 6) != 2) { n = 1; return 0xFFFFFFFFu; }
            v = (v << 6) | (d & 0x3F);
        }
        return v;
    }

    // bpe.rs:50-96.  Throws where the reference panics (a character missing from the
    // vocabulary, bpe.rs:55; a prompt that trims to nothing, bpe.rs:66 `len() - 1`).
    std::vector<size_t> encode(const std::string& text) const {
        size_t a = 0, b = text.size();
        for (size_t n; a < b && is_white_space(scalar_at(text, a, b, n)); a += n) {}       // str::trim, front
        while (b > a) {                                            // and back: step to the start of the last scalar
            size_t st = b - 1;
            while (st > a && ((unsigned char)text[st] >> 6) == 2) st--;
            size_t n;
            if (!is_white_space(scalar_at(text, st, b, n)) || st + n != b) break;
            b = st;
        }
        std::vector<size_t> tokens;
        for (size_t i = a; i < b;) {                               // .chars(): one UTF-8 scalar at a time
            unsigned char c = (unsigned char)text[i];
            size_t n = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : 4;
            if (i + n > b) n = b - i;
            std::string ch = text.substr(i, n);
            i += n;
            if (ch == "\n") continue;                              // bpe.rs:54
            auto it = word_token_map.find(ch);
            if (it == word_token_map.end()) throw std::runtime_error("character not in vocabulary (the reference panics, bpe.rs:55)");
            tokens.push_back(it->second);
        }
        if (tokens.empty()) throw std::runtime_error("empty prompt after trim (the reference underflows, bpe.rs:66)");
        std::string buf;
        for (;;) {
            float best_score = -1e10f;
            size_t best_token_id = (size_t)-1, best_idx = (size_t)-1;
            for (size_t idx = 0; idx + 1 < tokens.size(); idx++) {
                buf.assign(vocab[tokens[idx]]);
                buf.append(vocab[tokens[idx + 1]]);
                auto it = word_token_map.find(buf);
                if (it != word_token_map.end() && vocab_scores[it->second] > best_score) {
                    best_score = vocab_scores[it->second];
                    best_token_id = it->second;
                    best_idx = idx;
                }
            }
            if (best_idx == (size_t)-1) break;                     // bpe.rs:86-88
            tokens[best_idx] = best_token_id;
            tokens.erase(tokens.begin() + (long)best_idx + 1);
        }
        return tokens;
    }
};

// bpe.rs:101-115: "<s>" anywhere -> ""; "<0xAB>"-shaped -> that byte as a char (U+00AB, i.e. UTF-8
// encoded when >= 0x80); anything else unchanged.  Throws where from_str_radix().unwrap() panics
// (e.g. "<unk>").
inline std::string decode(const std::string& s) {
    if (s.find("<s>") != std::string::npos) return "";
    if (!s.empty() && s.front() == '<' && s.back() == '>') {
        if (s.size() < 5) throw std::runtime_error("decode: byte token too short (the reference panics)");
        auto hex = [](char c) -> int {
            if (c >= '0' && c <= '9') return c - '0';
            if (c >= 'a' && c <= 'f') return c - 'a' + 10;
            if (c >= 'A' && c <= 'F') return c - 'A' + 10;
            return -1;
        };
        int hi = hex(s[3]), lo = hex(s[4]);
        if (hi < 0 || lo < 0) throw std::runtime_error("decode: not a byte token (the reference panics, bpe.rs:110)");
        unsigned c = (unsigned)(hi * 16 + lo);
        std::string out;
        if (c < 0x80) out.push_back((char)c);
        else { out.push_back((char)(0xC0 | (c >> 6))); out.push_back((char)(0x80 | (c & 0x3F))); }
        return out;
    }
    return s;
}

}  // namespace rama_host
