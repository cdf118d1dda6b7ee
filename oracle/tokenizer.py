"""Python restatement of engine/src/tokenizer/bpe.rs (TEST INFRASTRUCTURE ONLY): tokenizer.bin
reader, greedy best-score pair merge, decode.  Raises where the reference panics."""
from __future__ import annotations

import struct


class Tokenizer:
    def __init__(self, path, vocab_size: int):
        """bpe.rs:19-45"""
        self.vocab, self.vocab_scores, self.word_token_map = [], [], {}
        with open(path, "rb") as f:
            self.max_token_length = struct.unpack("<I", f.read(4))[0]
            for idx in range(vocab_size):
                score, n = struct.unpack("<fi", f.read(8))
                s = f.read(n)
                if len(s) != n:
                    raise EOFError("tokenizer file truncated")
                s = s.decode("utf-8")            # String::from_utf8(..).unwrap()
                self.vocab.append(s)
                self.vocab_scores.append(score)
                self.word_token_map[s] = idx     # HashMap::insert: a later duplicate wins

    # str::trim (bpe.rs:53) strips the Unicode White_Space code points
    WHITE_SPACE = "".join(map(chr, [0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x20, 0x85, 0xA0, 0x1680, *range(0x2000, 0x200B), 0x2028, 0x2029, 0x202F, 0x205F, 0x3000]))

    def encode(self, s: str):
        """bpe.rs:50-96"""
        tokens = []
        for c in s.strip(self.WHITE_SPACE):
            if c == "\n":
                continue
            tokens.append(self.word_token_map[c])       # KeyError = the reference's unwrap() panic
        if not tokens:
            raise IndexError("tokens.len() - 1 underflows (bpe.rs:66)")
        while True:
            best_score, best_id, best_idx = -1e10, None, None
            for idx in range(len(tokens) - 1):
                tid = self.word_token_map.get(self.vocab[tokens[idx]] + self.vocab[tokens[idx + 1]])
                if tid is not None and self.vocab_scores[tid] > best_score:
                    best_score, best_id, best_idx = self.vocab_scores[tid], tid, idx
            if best_idx is None:
                break
            tokens[best_idx] = best_id
            del tokens[best_idx + 1]
        return tokens


def decode(s: str) -> str:
    """bpe.rs:101-115"""
    if "<s>" in s:
        return ""
    if len(s) > 0 and s[0] == "<" and s[-1] == ">":
        return chr(int(s[3:5], 16))       # ValueError = from_str_radix(..).unwrap() panic
    return s
