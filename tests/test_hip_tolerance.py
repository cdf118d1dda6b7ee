"""-m gpu: the TOLERANCE-MODE experiment (rama_set_tuning "ref_order" = 2; bench.py --mode tol).

north_star's bar is "logits within 1e-4 of the reference CPU path", not bit equality, and round 3's review asked for a
mode that keeps the reference's rounding only where its error was thought to live: the chain-order matvec kernels of
parity mode (every product and sum rounded like engine/src/device/cpu.rs:127-153), with the rmsnorm sums (cpu.rs:99-117)
tree-shaped and folded into the matvec that consumes them and attention (cpu.rs:23-52) by the fast path's kernel.
Built and measured: 218 tok/s at llama2-7B (0.72 of the roofline) but 1.4e-4 from the CPU path over 200 full-depth
positions -- no closer than the fast path (profiles/r04_tolerance_sweep_7b_200pos.jsonl); at the depths tested HERE
(<= 12 layers, <= 24 positions) it is well inside the bar.  The mode and its per-op switches stay as the instrument that
says which op carries how much of the distance.  What must hold, on every fixture and BASELINE shape:
  * logits within 1e-4 (expected ~1e-6) of the oracle at every position, greedy tokens identical;
  * cache rows and the residual stream within 1e-5 of the oracle's;
  * the chained generate() loop (hipGraph or eager) gives the oracle's tokens;
  * the per-op "tol_mask" switches (what tools/tol_sweep.py uses to say which op carries the distance) all stay
    within the bar on the small shapes.
The full-depth 7B x 200 positions assertion lives in tests/test_hip_parity_7b.py."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import synth as S

from .helpers import CKPT_CASES, LOGIT_ATOL, SYNTH_CASES, BIG_SYNTH_CASES, load_case, to_rama_cfg

pytestmark = pytest.mark.gpu

PROMPT = [10646, 2501, 263, 931]


@pytest.fixture(scope="module")
def dev():
    import rama_amd
    d = rama_amd.Hip(0)
    yield d
    d.close()


def _model_for(dev, name, cfg, g):
    import rama_amd
    from .helpers import GOLDEN
    if name.startswith("ckpt_"):
        return rama_amd.Model.load(dev, GOLDEN / f"{name}.bin")
    return rama_amd.Model.synth(dev, to_rama_cfg(cfg), int(g["seed"]), rope=(g["freq_cis_real"], g["freq_cis_imag"]))


@pytest.mark.parametrize("name", CKPT_CASES + SYNTH_CASES + BIG_SYNTH_CASES)
def test_fixture_logits_and_state_within_bar(dev, name):
    import rama_amd
    cfg, w, g = load_case(name)
    toks = g["tokens"].tolist()[:10]
    orc = O.Oracle(cfg, w)
    model = _model_for(dev, name, cfg, g)
    eng = rama_amd.Engine(dev, model)
    eng.set_tuning("ref_order", 2)
    try:
        worst = 0.0
        for pos, t in enumerate(toks):
            lo = orc.forward(t, pos)
            eng.forward(t, pos)
            lg = eng.logits()
            worst = max(worst, float(np.abs(lg - lo).max()))
            assert np.abs(lg - lo).max() <= LOGIT_ATOL, (name, pos, float(np.abs(lg - lo).max()))
            assert int(np.flatnonzero(lg == lg.max())[-1]) == O.argmax(lo)
        n = cfg.n_layers * cfg.seq_len * cfg.dim
        for buf in ("key_cache", "value_cache"):
            got, want = eng.buffer(buf, n), orc.s[buf].reshape(-1)
            assert np.abs(got - want).max() <= 1e-5, (name, buf, float(np.abs(got - want).max()))
        for buf, m in (("q", cfg.dim), ("hb", cfg.hidden_dim)):
            got, want = eng.buffer(buf, m), orc.s[buf].reshape(-1)
            assert np.abs(got - want).max() <= 1e-5 * max(1.0, float(np.abs(want).max())), (name, buf)
    finally:
        eng.set_tuning("ref_order", 0)
    eng.free(); model.free()


@pytest.mark.parametrize("shape", ["stories15M", "stories110M", "llama2-7B-2layers"])
@pytest.mark.parametrize("graph", [False, True])
def test_baseline_shapes_generate(dev, shape, graph):
    """full width and vocabulary: per-position logits within the bar, and the device-chained loop (eager and replayed from
    hipGraphs) produces the oracle's greedy tokens"""
    import rama_amd
    shapes = {"stories15M": (288, 768, 6, 6, 32000, 256, True), "stories110M": (768, 2048, 12, 12, 32000, 1024, True),
              "llama2-7B-2layers": (4096, 11008, 2, 32, 32000, 2048, False)}
    d, h, L, H, V, seq, shared = shapes[shape]
    cfg = O.Config(d, h, L, H, H, V, seq, shared)
    rope = S.rope_tables(seq, d // H)
    w = S.synth_weights(cfg, 0, rope=rope)
    orc = O.Oracle(cfg, w)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 0, rope=rope)
    eng = rama_amd.Engine(dev, model)
    eng.set_tuning("ref_order", 2)
    steps = 24 if d < 4096 else 8
    try:
        if not graph:
            token = 1
            for pos in range(steps):
                lo = orc.forward(token, pos)
                eng.forward(token, pos)
                dlt = float(np.abs(eng.logits() - lo).max())
                assert dlt <= LOGIT_ATOL, (shape, pos, dlt)
                token = PROMPT[pos] if pos < len(PROMPT) else O.argmax(lo)
        eng2 = rama_amd.Engine(dev, model)
        eng2.set_graph_mode(graph)
        eng2.set_tuning("prefill", 0)
        try:
            got = eng2.generate_greedy(PROMPT, steps)
        finally:
            eng2.set_tuning("prefill", 1)
            eng2.set_graph_mode(False)
        assert got == O.Oracle(cfg, w).generate_greedy(PROMPT, steps)
        eng2.free()
    finally:
        eng.set_tuning("ref_order", 0)
    eng.free(); model.free()


def test_long_context_attention_variants(dev):
    """tolerance mode takes the fast path's attention variants: one workgroup per head, fewer waves below position 256,
    split-T beyond it where a head's cache is large -- positions on both sides of the switch stay within the bar"""
    import rama_amd
    cfg = O.Config(256, 512, 2, 2, 2, 300, 1100, False)      # head size 128, 1 MiB+ of cache per head: split-T from 256 on
    rope = S.rope_tables(cfg.seq_len, cfg.head_size)
    w = S.synth_weights(cfg, 3, rope=rope)
    orc = O.Oracle(cfg, w)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 3, rope=rope)
    eng = rama_amd.Engine(dev, model)
    eng.set_tuning("ref_order", 2)
    try:
        rng = np.random.default_rng(1)
        toks = [1] + [int(v) for v in rng.integers(2, cfg.vocab_size, 299)]
        for pos, t in enumerate(toks):
            lo = orc.forward(t, pos)
            eng.forward(t, pos)
            if pos in (0, 1, 63, 64, 200, 254, 255, 256, 257, 299):
                dlt = float(np.abs(eng.logits() - lo).max())
                assert dlt <= LOGIT_ATOL, (pos, dlt)
    finally:
        eng.set_tuning("ref_order", 0)
    eng.free(); model.free()


@pytest.mark.parametrize("mask", [1, 2, 4, 8, 16, 32, 64, 127])
def test_op_switches(dev, mask):
    """every "tol_mask" switch (an op of tolerance mode swapped for the fast path's or for parity mode's) runs and stays
    within the bar at a small shape; mask 96 (parity attention + exact norms, all matvecs chain-order) is parity mode's bits"""
    import rama_amd
    cfg, w, g = load_case("synth_d288_h6")
    toks = g["tokens"].tolist()[:8]
    orc = O.Oracle(cfg, w)
    model = _model_for(dev, "synth_d288_h6", cfg, g)
    eng = rama_amd.Engine(dev, model)
    eng.set_tuning("ref_order", 2)
    eng.set_tuning("tol_mask", mask)
    try:
        for pos, t in enumerate(toks):
            lo = orc.forward(t, pos)
            eng.forward(t, pos)
            assert np.abs(eng.logits() - lo).max() <= LOGIT_ATOL, (mask, pos)
    finally:
        eng.set_tuning("tol_mask", 0)
        eng.set_tuning("ref_order", 0)
    eng.free(); model.free()


def test_exact_ops_mask_gives_parity_bits(dev):
    import rama_amd
    cfg, w, g = load_case("synth_d768_h12")
    toks = g["tokens"].tolist()[:6]
    orc = O.Oracle(cfg, w)
    model = _model_for(dev, "synth_d768_h12", cfg, g)
    eng = rama_amd.Engine(dev, model)
    eng.set_tuning("ref_order", 2)
    eng.set_tuning("tol_mask", 96)
    try:
        for pos, t in enumerate(toks):
            lo = orc.forward(t, pos)
            eng.forward(t, pos)
            assert np.array_equal(eng.logits().view(np.uint32), lo.view(np.uint32)), pos
    finally:
        eng.set_tuning("tol_mask", 0)
        eng.set_tuning("ref_order", 0)
    eng.free(); model.free()


def test_wide_model_keeps_norm_launches(dev):
    """dim > 8192 does not fit a workgroup's registers: the norms stay launches of their own (exact sums), the rest is
    tolerance mode"""
    import rama_amd
    cfg = O.Config(8448, 64, 1, 66, 66, 40, 8, False)
    rope = S.rope_tables(cfg.seq_len, cfg.head_size)
    w = S.synth_weights(cfg, 5, rope=rope)
    orc = O.Oracle(cfg, w)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 5, rope=rope)
    eng = rama_amd.Engine(dev, model)
    eng.set_tuning("ref_order", 2)
    try:
        for pos, t in enumerate([1, 5, 9]):
            lo = orc.forward(t, pos)
            eng.forward(t, pos)
            assert np.abs(eng.logits() - lo).max() <= LOGIT_ATOL, pos
    finally:
        eng.set_tuning("ref_order", 0)
    eng.free(); model.free()
