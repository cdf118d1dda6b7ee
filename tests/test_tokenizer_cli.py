"""Host-side string code: the C++ tokenizer (rama_amd/csrc/host/tokenizer.hpp)
against the Python restatement of bpe.rs (oracle/tokenizer.py) on synthetic vocabularies (CPU),
and -- marked gpu -- the engine CLI's printed text against oracle generate() + decode."""
import os
import random
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O
from oracle.tokenizer import Tokenizer, decode
from tests.helpers import GOLDEN, load_case

REPO = Path(__file__).resolve().parent.parent
TOOL = REPO / "rama_amd" / "bin" / "tokenizer_tool"
ENGINE = REPO / "rama_amd" / "bin" / "engine"
REF_TOK = Path("/root/reference/engine/tokenizer.bin")


def write_tokenizer(path, entries):
    with open(path, "wb") as f:
        f.write(struct.pack("<I", max(len(s.encode()) for s, _ in entries)))
        for s, score in entries:
            b = s.encode()
            f.write(struct.pack("<fi", score, len(b)))
            f.write(b)


def synthetic_vocab(n, seed):
    """<unk>, <s>, </s>, byte tokens, single chars (incl. a 2-byte one), then scored merges"""
    rng = random.Random(seed)
    entries = [("<unk>", 0.0), ("<s>", 0.0), ("</s>", 0.0), ("<0x0A>", 0.0), ("<0xE9>", 0.0)]
    chars = list("abcdefgh ") + ["é"]
    entries += [(c, -1.0 - i) for i, c in enumerate(chars)]
    seen = {s for s, _ in entries}
    pool = [c for c in chars]
    while len(entries) < n:
        a, b = rng.choice(pool), rng.choice(pool)
        m = a + b
        if m in seen or len(m) > 6:
            continue
        seen.add(m)
        pool.append(m)
        entries.append((m, rng.uniform(-10, 0)))
    return entries


def run_tool(*args):
    return subprocess.run([str(TOOL), *map(str, args)], capture_output=True, text=True)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_cpp_tokenizer_matches_restatement(tmp_path, seed):
    entries = synthetic_vocab(64, seed)
    p = tmp_path / "tok.bin"
    write_tokenizer(p, entries)
    tok = Tokenizer(p, len(entries))
    rng = random.Random(100 + seed)
    for _ in range(60):
        text = "".join(rng.choice("abcdefgh é") for _ in range(rng.randint(1, 24)))
        if not text.strip():
            continue
        r = run_tool(p, len(entries), "encode", text)
        assert r.returncode == 0, r.stderr
        assert [int(v) for v in r.stdout.split()] == tok.encode(text), text
    ids = list(range(3, len(entries)))
    r = run_tool(p, len(entries), "decode", *ids)
    assert r.returncode == 0, r.stderr
    assert r.stdout.rstrip("\n") == "".join(decode(tok.vocab[i]) for i in ids).rstrip("\n")


def test_tokenizer_edge_cases(tmp_path):
    entries = synthetic_vocab(40, 3)
    p = tmp_path / "tok.bin"
    write_tokenizer(p, entries)
    tok = Tokenizer(p, len(entries))
    assert run_tool(p, len(entries), "encode", "  abc\n").stdout.split() == [str(v) for v in tok.encode("  abc\n")]
    assert tok.encode("a\nb") == tok.encode("ab")                       # bpe.rs:54 skips '\n'
    assert run_tool(p, len(entries), "encode", "xyz").returncode == 101   # unknown char: reference panics (bpe.rs:55)
    with pytest.raises(KeyError):
        tok.encode("xyz")
    assert run_tool(p, len(entries), "encode", "   ").returncode == 101   # trims to nothing: bpe.rs:66 underflow
    assert run_tool(p, len(entries), "decode", 1).stdout == "\n"          # "<s>" -> ""
    assert run_tool(p, len(entries), "decode", 0).returncode == 101       # "<unk>": from_str_radix panics
    assert run_tool(p, len(entries), "decode", 4).stdout.encode() == "é\n".encode()   # <0xE9> -> U+00E9
    assert run_tool(tmp_path / "missing.bin", 4, "encode", "a").returncode == 101


def test_trim_is_unicode_white_space(tmp_path):
    """str::trim (bpe.rs:53) strips every White_Space code point at both ends -- U+00A0, U+2003, U+3000, U+0085 ... --
    and nothing else (U+200B ZERO WIDTH SPACE and U+001F are no white space for Rust: they reach the vocabulary lookup)"""
    entries = synthetic_vocab(40, 5)
    p = tmp_path / "tok.bin"
    write_tokenizer(p, entries)
    tok = Tokenizer(p, len(entries))
    want = [str(v) for v in tok.encode("abc")]
    for pad in ("\u00a0", "\u2003", "\u3000", "\u0085", "\u1680 \t", "\u2028\u2029", "\u202f\u205f"):
        r = run_tool(p, len(entries), "encode", pad + "abc" + pad)
        assert r.returncode == 0 and r.stdout.split() == want, (pad.encode("unicode_escape"), r.stderr)
        assert tok.encode(pad + "abc" + pad) == tok.encode("abc")
    for nonspace in ("\u200b", "\u001f"):
        assert run_tool(p, len(entries), "encode", nonspace + "abc").returncode == 101      # not trimmed, not in the vocabulary
        with pytest.raises(KeyError):
            tok.encode(nonspace + "abc")
    assert run_tool(p, len(entries), "encode", "\u3000\u00a0").returncode == 101          # trims to nothing


@pytest.mark.skipif(not REF_TOK.exists(), reason="reference tokenizer.bin only exists in the build container")
def test_reference_tokenizer_prompt_ids():
    """'once upon a time' -> [10646, 2501, 263, 931] (no sentencepiece dummy prefix)"""
    assert run_tool(REF_TOK, 32000, "encode", "once upon a time").stdout.split() == ["10646", "2501", "263", "931"]
    assert Tokenizer(REF_TOK, 32000).encode("once upon a time") == [10646, 2501, 263, 931]


# ------------------------------------------------------------------ the CLI on the GPU

def cli_vocab(vocab_size):
    base = [("<unk>", 0.0), ("<s>", 0.0), ("</s>", 0.0)]
    chars = " .,!?abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
    return (base + [(c, -float(i)) for i, c in enumerate(chars)])[:vocab_size]


@pytest.mark.gpu
@pytest.mark.parametrize("ref_order", [None, "0", "7"])
@pytest.mark.parametrize("path", ["fused", "ops", "chained"])
def test_engine_cli_text_parity(tmp_path, path, ref_order):
    """the CLI runs PARITY mode by default (RAMA_REF_ORDER unset: every op in cpu.rs's rounding order, on weights it uploads tensor by
    tensor like hbm.rs:55-90); RAMA_REF_ORDER=0 selects the fast path; anything outside 0..2 is refused"""
    name = "ckpt_untied"
    cfg, w, g = load_case(name)
    entries = cli_vocab(cfg.vocab_size)
    assert len(entries) == cfg.vocab_size
    tokp = tmp_path / "tok.bin"
    write_tokenizer(tokp, entries)
    tok = Tokenizer(tokp, cfg.vocab_size)
    prompt, steps = "hi b", 12
    want_ids = O.Oracle(cfg, w).generate_greedy(tok.encode(prompt), steps)
    # "<unk>" would make decode panic like the reference; the expected text must not contain it
    want_text = "".join(decode(tok.vocab[i]) if i != 0 else None for i in want_ids) if 0 not in want_ids else None
    import os
    env = dict(os.environ, RAMA_PATH=path)
    env.pop("RAMA_REF_ORDER", None)
    if ref_order is not None:
        env["RAMA_REF_ORDER"] = ref_order
    r = subprocess.run([str(ENGINE), "-m", str(GOLDEN / f"{name}.bin"), "-t", str(tokp), "-p", prompt, "-s", str(steps), "-r", "0"],
                       capture_output=True, text=True, env=env, timeout=120)
    if ref_order == "7":
        assert r.returncode == 2 and "RAMA_REF_ORDER" in r.stderr
        return
    if want_text is None:
        assert r.returncode == 101
        return
    assert r.returncode == 0, r.stderr
    body, _, tail = r.stdout.partition("\n--------------------------------\n")
    assert body == want_text, (body, want_text)
    assert tail.startswith("elapsed: ") and "avg tok/s: " in tail


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["fused", "chained"])
def test_engine_cli_temperature_one_text_parity(tmp_path, path):
    """`-r 1` (the reference README's bench setting, README.md:80-83): top-p sampling with the
    re-seeded constant draw; "chained" samples on the device, "fused" through the trait's sample()"""
    import os
    name = "ckpt_untied"
    cfg, w, g = load_case(name)
    tokp = tmp_path / "tok.bin"
    write_tokenizer(tokp, cli_vocab(cfg.vocab_size))
    tok = Tokenizer(tokp, cfg.vocab_size)
    prompt, steps, u = "hi b", 12, 0.2721174359321594
    pt = tok.encode(prompt)
    orc = O.Oracle(cfg, w)
    token, want_ids = 1, []
    for pos in range(steps):
        lo = orc.forward(token, pos)
        nxt = pt[pos] if pos < len(pt) else O.sample(lo.copy(), 1.0, 0.9, u)
        want_ids.append(int(nxt)); token = nxt
    r = subprocess.run([str(ENGINE), "-m", str(GOLDEN / f"{name}.bin"), "-t", str(tokp), "-p", prompt, "-s", str(steps), "-r", "1", "-l", "0.9"],
                       capture_output=True, text=True, env=dict(os.environ, RAMA_PATH=path), timeout=120)
    if 0 in want_ids:
        assert r.returncode == 101
        return
    assert r.returncode == 0, r.stderr
    body = r.stdout.partition("\n--------------------------------\n")[0]
    assert body == "".join(decode(tok.vocab[i]) for i in want_ids)


@pytest.mark.gpu
@pytest.mark.parametrize("temperature", ["0", "1"])
def test_engine_cli_pipeline_stage_mode_one_rank(tmp_path, temperature):
    """RAMA_WORLD / RAMA_RANK / RAMA_PIPE_ID_FILE: the CLI as a pipeline stage (csrc/pipe.hip: unique id
    through a file, rama_model_load_stage, the native tick loop, device sampler).  One GPU here, so one
    rank: the text must equal generate()'s."""
    import os
    name = "ckpt_untied"
    cfg, w, g = load_case(name)
    tokp = tmp_path / "tok.bin"
    write_tokenizer(tokp, cli_vocab(cfg.vocab_size))
    tok = Tokenizer(tokp, cfg.vocab_size)
    prompt, steps, u = "hi b", 12, 0.2721174359321594
    pt = tok.encode(prompt)
    orc = O.Oracle(cfg, w)
    token, want_ids = 1, []
    for pos in range(steps):
        lo = orc.forward(token, pos)
        nxt = pt[pos] if pos < len(pt) else O.sample(lo.copy(), float(temperature), 0.9, u)
        want_ids.append(int(nxt)); token = nxt
    env = dict(os.environ, RAMA_WORLD="1", RAMA_RANK="0", RAMA_PIPE_ID_FILE=str(tmp_path / "pipe.id"), RAMA_DEVICE="0")
    r = subprocess.run([str(ENGINE), "-m", str(GOLDEN / f"{name}.bin"), "-t", str(tokp), "-p", prompt, "-s", str(steps), "-r", temperature, "-l", "0.9"],
                       capture_output=True, text=True, env=env, timeout=120)
    if 0 in want_ids:
        assert r.returncode in (101, 0)
        return
    assert r.returncode == 0, r.stderr
    body = r.stdout.partition("\n--------------------------------\n")[0]
    assert body == "".join(decode(tok.vocab[i]) for i in want_ids)


@pytest.mark.gpu
def test_engine_cli_rejects_too_many_steps(tmp_path):
    tokp = tmp_path / "tok.bin"
    write_tokenizer(tokp, cli_vocab(64))
    r = subprocess.run([str(ENGINE), "-m", str(GOLDEN / "ckpt_tied.bin"), "-t", str(tokp), "-s", "17", "-r", "0"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "seq_len" in r.stderr


@pytest.mark.parametrize("var,value", [("RAMA_REF_ORDER", "parity"), ("RAMA_REF_ORDER", ""), ("RAMA_REF_ORDER", "1x"), ("RAMA_REF_ORDER", "4"), ("RAMA_LANE_REDUCE", "pairwise"), ("RAMA_LANE_REDUCE", "3")])
def test_cli_refuses_garbage_mode_variables(var, value):
    """[r6] (round 5's advisor) `atoi` turned RAMA_REF_ORDER=parity into 0 -- FAST mode, silently, although the default is parity.  The C++ host now parses with
    an end-pointer check BEFORE anything touches a GPU: garbage or an out-of-range value exits 2 with a message naming the variable (runs without a GPU)"""
    if not ENGINE.exists():
        pytest.skip("engine binary not built")
    r = subprocess.run([str(ENGINE), "-m", str(GOLDEN / "ckpt_tied.bin"), "-t", str(REPO / "tests" / "golden" / "nonexistent_tokenizer.bin"), "-s", "3", "-r", "0"],
                       capture_output=True, text=True, env=dict(os.environ, **{var: value}), timeout=60)
    assert r.returncode == 2 and var in r.stderr, (r.returncode, r.stderr[-300:])
