"""The oracle (CPU restatement of cpu.rs + infer.rs) against fixtures produced by the
reference's own PyTorch model definition (tools/make_goldens.py) and against the one
known-answer vector the reference's test suite holds (gpu.rs:249-288)."""
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O
from oracle import synth as S
from tests.helpers import (BIG_SYNTH_CASES, CKPT_CASES, LOGIT_ATOL, SYNTH_CASES, load_case)


@pytest.mark.parametrize("name", CKPT_CASES + SYNTH_CASES + BIG_SYNTH_CASES)
def test_oracle_logits_match_reference_model(name):
    cfg, w, g = load_case(name)
    orc = O.Oracle(cfg, w)
    tokens = g["tokens"].tolist()
    worst = 0.0
    for pos, tok in enumerate(tokens):
        logits = orc.forward(tok, pos)
        worst = max(worst, float(np.abs(logits - g["logits"][pos]).max()))
    assert worst <= LOGIT_ATOL, f"{name}: max |oracle - reference model| = {worst:.3e}"


@pytest.mark.parametrize("name", CKPT_CASES + SYNTH_CASES)
def test_oracle_layer0_intermediates(name):
    """Each stage at its own scale (a whole-network tolerance can hide an O(1)-wrong stage)."""
    cfg, w, g = load_case(name)
    orc = O.Oracle(cfg, w)
    tokens = g["tokens"].tolist()
    T = len(tokens)
    for pos in range(T - 1):
        orc.forward(tokens[pos], pos)
    # replay the last position op by op for layer 0
    pos, tok = T - 1, tokens[-1]
    d, hd, hs = cfg.dim, cfg.hidden_dim, cfg.head_size
    s = orc.s
    s["x"][:] = orc.w["token_embedding_table"][tok]
    O.rmsnorm(s["xb"], s["x"], orc.w["rms_att_weight"][0], d)
    np.testing.assert_allclose(s["xb"], g["l0_xb_attnorm"], atol=2e-6, rtol=1e-5)
    O.matmul(s["q"], orc.w["wq"][0], s["xb"], d, d)
    O.matmul(s["k"], orc.w["wk"][0], s["xb"], d, d)
    O.matmul(s["v"], orc.w["wv"][0], s["xb"], d, d)
    np.testing.assert_allclose(s["q"], g["l0_q_prerope"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(s["k"], g["l0_k_prerope"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(s["v"], g["l0_v"], atol=2e-6, rtol=1e-5)
    pr = orc.w["freq_cis_real"][pos]
    pi = orc.w["freq_cis_imag"][pos]
    for h in range(cfg.n_heads):
        O.apply_position(s["q"][h * hs:], s["k"][h * hs:], pr, pi, hs)
    s["key_cache"][pos * d:(pos + 1) * d] = s["k"]
    s["value_cache"][pos * d:(pos + 1) * d] = s["v"]
    orc.multi_head_attention(0, pos)
    np.testing.assert_allclose(s["xb"], g["l0_att_out"], atol=2e-6, rtol=1e-5)
    O.matmul(s["xb2"], orc.w["wo"][0], s["xb"], d, d)
    np.testing.assert_allclose(s["xb2"], g["l0_xb2"], atol=2e-6, rtol=1e-5)
    O.array_add(s["x"], s["xb2"], d)
    O.rmsnorm(s["xb"], s["x"], orc.w["rms_ffn_weight"][0], d)
    np.testing.assert_allclose(s["xb"], g["l0_xb_ffnnorm"], atol=2e-6, rtol=1e-5)
    O.matmul(s["hb"], orc.w["w1"][0], s["xb"], d, hd)
    O.matmul(s["hb2"], orc.w["w3"][0], s["xb"], d, hd)
    O.sinu(s["hb"], hd)
    O.array_mult(s["hb"], s["hb2"], hd)
    np.testing.assert_allclose(s["hb"], g["l0_hb"], atol=2e-6, rtol=1e-5)
    O.matmul(s["xb"], orc.w["w2"][0], s["hb"], hd, d)
    O.array_add(s["x"], s["xb"], d)
    np.testing.assert_allclose(s["x"], g["l0_x_out"], atol=2e-6, rtol=1e-5)


def test_matmul_reference_known_answer():
    """gpu.rs:249-288 test_blas: 2x3 . 3x4 with one 5.0; the commented-out assert expects
    18s (22 where the 5.0 lands).  Device::matmul(o, a=l, b=r, width=3, o_rows=4, o_cols=2)
    is that product; the CPU body needs width % 4 == 0, so k is zero-padded to 4."""
    l = np.array([3, 3, 3, 0, 3, 5, 3, 0, 3, 3, 3, 0, 3, 3, 3, 0], dtype=np.float32)   # [4][4]
    r = np.array([2, 2, 2, 2, 2, 2, 0, 0], dtype=np.float32)                             # [4][2]
    o = np.ones(8, dtype=np.float32)
    O.matmul(o, l, r, 4, 4, 2)
    assert o.tolist() == [18, 18, 22, 22, 18, 18, 18, 18]
    with pytest.raises(ValueError):
        O.matmul(o, l, r, 3, 4, 2)   # reference panics at cpu.rs:143


def test_argmax_last_max_wins():
    x = np.array([0.0, 3.0, 1.0, 3.0, 2.0], dtype=np.float32)
    assert O.argmax(x) == 3            # cpu.rs:165-167: ties -> last index
    assert O.argmax(np.array([5.0, 1.0], dtype=np.float32)) == 0
    assert O.argmax(np.array([7.0], dtype=np.float32)) == 0


def test_sample_top_p_semantics():
    logits = np.log(np.array([0.5, 0.3, 0.15, 0.05], dtype=np.float32))
    # topp 0.9: cutoff=(0.1)/3=0.0333 keeps all four; sorted 0.5,0.3,0.15,0.05; cum>0.9 at i=2
    assert O.sample(logits.copy(), 1.0, 0.9, 0.0) == 0
    assert O.sample(logits.copy(), 1.0, 0.9, 0.6) == 1     # r=0.57 -> cdf .5,.8
    assert O.sample(logits.copy(), 1.0, 0.9, 0.99) == 2    # falls through to last_index
    assert O.sample(logits.copy(), 0.0, 0.9, 0.5) == 0     # T==0 -> argmax
    a = logits.copy(); b = logits.copy()
    O.sample(a, 2.0, 0.9, 0.5); O.sample(b, 1.0, 0.9, 0.5)
    assert np.array_equal(a, b)                             # T>1 has no effect (cpu.rs:170-172)


def test_forward_range_composes():
    cfg, w, g = load_case("synth_d64_h4")
    a, b = O.Oracle(cfg, w), O.Oracle(cfg, w)
    for pos, tok in enumerate(g["tokens"].tolist()[:6]):
        la = a.forward(tok, pos).copy()
        b.forward_range(tok, pos, 0, 1, True, False)
        b.forward_range(tok, pos, 1, cfg.n_layers, False, True)
        assert np.array_equal(la, b.s["logits"])


def test_f64_arbiter_close():
    cfg, w, g = load_case("synth_d288_h6")
    a, b = O.Oracle(cfg, w), O.Oracle(cfg, w)
    for pos, tok in enumerate(g["tokens"].tolist()[:8]):
        l32 = a.forward(tok, pos).copy()
        l64 = b.forward_f64(tok, pos).copy()
        assert np.abs(l32 - l64).max() < 2e-5


def test_synth_fill_c_equals_numpy():
    for seed, tag, n, off in [(0, 1, 1000, 0), (7, 9, 4097, 123456789), (3, 12, 65536, 1 << 33)]:
        sc = np.float32(0.02 / S.IH4_STD)
        assert np.array_equal(O.fill_synth(n, seed, tag, sc, 0.0, off), S.fill_numpy(n, seed, tag, sc, 0.0, off))
    a = O.fill_synth(1 << 20, 5, 3, np.float32(0.02 / S.IH4_STD))
    assert abs(a.std() - 0.02) < 2e-4 and abs(a.mean()) < 1e-4
    # offset consistency: a slice generated on its own equals the slice of the whole
    whole = O.fill_synth(5000, 2, 4, sc)
    part = O.fill_synth(1000, 2, 4, sc, 0.0, 3000)
    assert np.array_equal(whole[3000:4000], part)


def test_checkpoint_reader_layout(golden_dir):
    cfg, w = O.read_checkpoint(golden_dir / "ckpt_untied.bin")
    assert not cfg.shared_weight and cfg.vocab_size == 64
    assert w["wcls"].shape == (64, 32) and w["wcls"] is not w["token_embedding_table"]
    cfg2, w2 = O.read_checkpoint(golden_dir / "ckpt_tied.bin")
    assert cfg2.shared_weight and w2["wcls"] is w2["token_embedding_table"]
    hs = cfg.head_size
    assert w["freq_cis_real"].shape == (cfg.seq_len, hs // 2)
    assert np.allclose(w["freq_cis_real"][0], 1.0) and np.allclose(w["freq_cis_imag"][0], 0.0)


def test_v1_ak42_fixture_is_not_a_v0_file(golden_dir):
    """the reference's version1_export wrote this one: magic "ak42", 256-byte header (export.py:132-180)"""
    raw = (golden_dir / "ckpt_v1_ak42.bin").read_bytes()
    assert raw[:4] == b"24ka" and int.from_bytes(raw[:4], "little") == 0x616b3432
    assert int.from_bytes(raw[4:8], "little") == 1                      # version
    assert int.from_bytes(raw[8:12], "little") == 32                    # dim of the tiny model
    with pytest.raises(ValueError, match="ak42"):
        O.read_checkpoint(golden_dir / "ckpt_v1_ak42.bin")


# ------------------------------------------------------------------ [r6] the two orders the reference leaves to its crates

def test_lane_reduce_probe_vector():
    """cpu.rs:148 `v.reduce_add()`: the oracle's three lane orders on the probe vector the Rust shim asks `wide` with at start-up
    (integration/rust/hip.rs probe_lane_reduce): three DIFFERENT fp32 results, the bit patterns the shim matches on"""
    e = np.float32(2.0 ** -24)
    probe = np.array([1.0, e, -1.0, np.float32(1.5) * e], np.float32)
    ones = np.ones(4, np.float32)
    want = {"pairwise": 0x34000000, "strided": 0x34200000, "sequential": 0x33C00000}
    text = (Path(__file__).resolve().parent.parent / "integration" / "rust" / "hip.rs").read_text()
    for order, bits_ in want.items():
        with O.orders(lane_reduce=order):
            o = np.zeros(1, np.float32)
            O.matmul(o, probe, ones, 4, 1)
        assert int(o.view(np.uint32)[0]) == bits_, (order, hex(int(o.view(np.uint32)[0])))
        assert f"{bits_:08x}" in text.replace("_", "").lower(), f"hip.rs does not match on {bits_:#x}"
    assert O.lib().oracle_get_lane_reduce() == 0          # the context manager restored the default


def test_lane_reduce_and_softmax_split_orders():
    """the oracle's switches against plain numpy restatements: the final 4-lane sum in each order (cpu.rs:141-148), rayon's halving tree over the
    softmax's exponentials (cpu.rs:190) with 2^levels leaves -- and level 0 / pairwise are the defaults every golden fixture was checked with"""
    rng = np.random.default_rng(5)
    rows, width = 37, 260
    a = rng.standard_normal((rows, width)).astype(np.float32)
    b = rng.standard_normal(width).astype(np.float32)
    lanes = []
    for j in range(4):
        acc = np.zeros(rows, np.float32)
        for k in range(j, width, 4):
            acc = acc + a[:, k] * b[k]
        lanes.append(acc)
    want = {"pairwise": (lanes[0] + lanes[1]) + (lanes[2] + lanes[3]), "strided": (lanes[0] + lanes[2]) + (lanes[1] + lanes[3]),
            "sequential": ((lanes[0] + lanes[1]) + lanes[2]) + lanes[3]}
    for order, w_ in want.items():
        with O.orders(lane_reduce=order):
            o = np.zeros(rows, np.float32)
            O.matmul(o, a, b, width, rows)
        assert np.array_equal(o.view(np.uint32), w_.astype(np.float32).view(np.uint32)), order

    def tree(x, levels):
        if levels <= 0 or x.size < 2:
            s = np.float32(0.0)
            for v in x:
                s = np.float32(s + v)
            return s
        mid = x.size // 2
        return np.float32(tree(x[:mid], levels - 1) + tree(x[mid:], levels - 1))

    for n in (1, 2, 3, 33, 200, 1901):
        x = (rng.standard_normal(n) * 3).astype(np.float32)
        ex = O.expf(x - x.max())
        for levels in (0, 1, 2, 5, 12):
            with O.orders(softmax_split=levels):
                y = x.copy()
                O.softmax(y, n)
            assert np.array_equal(y.view(np.uint32), (ex / tree(ex, levels)).astype(np.float32).view(np.uint32)), (n, levels)
    assert O.lib().oracle_get_softmax_split() == 0
