// tokenizer_tool.cpp -- exercise host/tokenizer.hpp without a GPU (tests):
//   tokenizer_tool <tokenizer.bin> <vocab_size> encode <text>   -> space-separated ids
//   tokenizer_tool <tokenizer.bin> <vocab_size> decode <id>...  -> decoded pieces, concatenated
// exit code 101 where the reference would panic.
#include "tokenizer.hpp"

#include <cstdio>
#include <cstdlib>
#include <iostream>

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: tokenizer_tool <tokenizer.bin> <vocab_size> encode|decode ...\n"); return 2; }
    try {
        auto tok = rama_host::Tokenizer::from_file(argv[1], (size_t)std::atol(argv[2]));
        std::string cmd = argv[3];
        if (cmd == "encode") {
            auto ids = tok.encode(argv[4]);
            for (size_t i = 0; i < ids.size(); i++) std::printf("%s%zu", i ? " " : "", ids[i]);
            std::printf("\n");
        } else if (cmd == "decode") {
            for (int i = 4; i < argc; i++) std::cout << rama_host::decode(tok.vocab.at((size_t)std::atol(argv[i])));
            std::cout << std::endl;
        } else return 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "panic: %s\n", e.what());
        return 101;
    }
    return 0;
}
