"""-m gpu: forward()/generate() parity.  Three product paths -- (A) forward() composed from
the 1:1 Device ops exactly like infer.rs:8-53, (B) the fused rama_forward, (C) the
device-chained greedy loop (eager and hipGraph) -- against the CPU oracle on identical
tokens, and against the golden logits of the reference's own PyTorch model.
Bar (north_star): max |logit difference| <= 1e-4 absolute, fp32."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import synth as S
from tests.helpers import (BIG_SYNTH_CASES, CKPT_CASES, GOLDEN, LOGIT_ATOL, SYNTH_CASES, gpu_views,
                           load_case, synth_at, to_rama_cfg)

pytestmark = pytest.mark.gpu

STATE_ATOL = 2e-5   # per-buffer bound for the full RunState comparison (activations are O(1))


@pytest.fixture(scope="module")
def dev():
    import rama_amd
    d = rama_amd.Hip(0)
    yield d
    d.close()


@pytest.mark.parametrize("name", CKPT_CASES + SYNTH_CASES)
def test_forward_trait_ops_full_state_parity(dev, name):
    """(A): every RunState buffer after every position, via to_cpu (device.rs:21)."""
    import rama_amd
    cfg, w, g = load_case(name)
    orc = O.Oracle(cfg, w)
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    worst_logit = 0.0
    for pos, tok in enumerate(g["tokens"].tolist()):
        orc.forward(tok, pos)
        rama_amd.forward(rcfg, wv, rsv, tok, pos, dev)
        cpu_state = {}
        dev.to_cpu(rsv, cpu_state)
        for buf in ("x", "xb", "xb2", "hb", "hb2", "q", "k", "v", "logits", "key_cache", "value_cache"):
            d = float(np.abs(cpu_state[buf] - orc.s[buf]).max())
            assert d <= STATE_ATOL, f"{name} pos {pos} buffer {buf}: {d:.3e}"
        att_g = cpu_state["att"].reshape(cfg.n_heads, cfg.seq_len)[:, :pos + 1]
        att_o = orc.s["att"].reshape(cfg.n_heads, cfg.seq_len)[:, :pos + 1]
        assert np.abs(att_g - att_o).max() <= 1e-6
        worst_logit = max(worst_logit, float(np.abs(cpu_state["logits"] - g["logits"][pos]).max()))
    assert worst_logit <= LOGIT_ATOL
    rs.free(); ws.free()


@pytest.mark.parametrize("name", CKPT_CASES + SYNTH_CASES + BIG_SYNTH_CASES)
def test_forward_fused_vs_oracle_and_golden(dev, name):
    """(B): logits, KV caches and the residual stream of the fused path."""
    import rama_amd
    cfg, w, g = load_case(name)
    orc = O.Oracle(cfg, w)
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    worst_o = worst_g = 0.0
    for pos, tok in enumerate(g["tokens"].tolist()):
        lo = orc.forward(tok, pos)
        rama_amd.forward_fused(rcfg, wv, rsv, tok, pos, dev)
        lg = dev.download(rsv.logits)
        worst_o = max(worst_o, float(np.abs(lg - lo).max()))
        worst_g = max(worst_g, float(np.abs(lg - g["logits"][pos]).max()))
    assert worst_o <= LOGIT_ATOL, f"{name}: fused vs oracle {worst_o:.3e}"
    assert worst_g <= LOGIT_ATOL, f"{name}: fused vs reference-model golden {worst_g:.3e}"
    for buf in ("key_cache", "value_cache"):
        assert np.abs(dev.download(getattr(rsv, buf)) - orc.s[buf]).max() <= STATE_ATOL
    rs.free(); ws.free()


@pytest.mark.parametrize("name", ["synth_d288_h6", "synth_d768_h12"])
def test_fused_layer0_intermediates_vs_golden(dev, name):
    """each fused stage at its own scale against the reference model's hooks (a whole-network
    tolerance hides an O(1)-wrong stage): q/k after RoPE via the cache, attention output,
    SwiGLU output, residual after layer 0."""
    import rama_amd
    cfg, w, g = load_case(name)
    w1 = {k: (v[:1] if k in ("rms_att_weight", "rms_ffn_weight", "wq", "wk", "wv", "wo", "w1", "w2", "w3") else v) for k, v in w.items()}
    cfg1 = O.Config(cfg.dim, cfg.hidden_dim, 1, cfg.n_heads, cfg.n_kv_heads, cfg.vocab_size, cfg.seq_len, cfg.shared_weight)
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg1, w1)
    toks = g["tokens"].tolist()
    for pos, tok in enumerate(toks):
        rama_amd.forward_fused(rcfg, wv, rsv, tok, pos, dev)
    pos = len(toks) - 1
    np.testing.assert_allclose(dev.download(rsv.v), g["l0_v"], atol=3e-6, rtol=1e-5)
    hs = cfg.head_size
    pr, pi = w["freq_cis_real"][pos], w["freq_cis_imag"][pos]
    kr = g["l0_k_prerope"].astype(np.float64).reshape(cfg.n_heads, hs // 2, 2)
    k_rot = np.stack([kr[..., 0] * pr - kr[..., 1] * pi, kr[..., 0] * pi + kr[..., 1] * pr], axis=-1).reshape(-1)
    np.testing.assert_allclose(dev.download(rsv.k), k_rot, atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(dev.download(rsv.key_cache)[pos * cfg.dim:(pos + 1) * cfg.dim], k_rot, atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(dev.download(rsv.xb), g["l0_att_out"], atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(dev.download(rsv.hb), g["l0_hb"], atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(dev.download(rsv.x), g["l0_x_out"], atol=3e-6, rtol=1e-5)
    rs.free(); ws.free()


@pytest.mark.parametrize("name", ["ckpt_tied", "ckpt_untied"])
def test_model_load_v0_checkpoint(dev, name):
    """rama_model_load (mmap + one H2D) on the file the reference exporter wrote."""
    import rama_amd
    cfg, w, g = load_case(name)
    m = rama_amd.Model.load(dev, GOLDEN / f"{name}.bin")
    assert m.cfg == to_rama_cfg(cfg)
    assert m.bytes == (GOLDEN / f"{name}.bin").stat().st_size - 28
    for t in ("token_embedding_table", "wq", "w2", "rms_final_weight", "freq_cis_imag", "wcls"):
        assert np.array_equal(m.tensor(t, w[t].size), np.asarray(w[t]).reshape(-1)), t
    eng = rama_amd.Engine(dev, m)
    worst = 0.0
    for pos, tok in enumerate(g["tokens"].tolist()):
        eng.forward(tok, pos)
        worst = max(worst, float(np.abs(eng.logits() - g["logits"][pos]).max()))
    assert worst <= LOGIT_ATOL
    eng.free(); m.free()


def test_model_load_rejects_bad_files(dev, tmp_path):
    import rama_amd
    p = tmp_path / "trunc.bin"
    p.write_bytes((GOLDEN / "ckpt_tied.bin").read_bytes()[:-4])
    with pytest.raises(rama_amd.RamaError):
        rama_amd.Model.load(dev, p)
    with pytest.raises(rama_amd.RamaError):
        rama_amd.Model.load(dev, tmp_path / "missing.bin")


@pytest.mark.parametrize("name", ["ckpt_tied", "ckpt_untied"])
def test_model_save_reproduces_the_exporters_file(dev, tmp_path, name):
    """load -> rama_model_save writes back, byte for byte, the v0 file the reference's
    export.py produced (header incl. the negated vocab for the untied classifier, tensor order)"""
    import rama_amd
    m = rama_amd.Model.load(dev, GOLDEN / f"{name}.bin")
    out = tmp_path / "resaved.bin"
    m.save(out)
    assert out.read_bytes() == (GOLDEN / f"{name}.bin").read_bytes()
    m.free()


def test_model_save_synth_roundtrip(dev, tmp_path):
    """synthetic model -> v0 file -> the oracle's independent reader sees the oracle's own
    weights; loading the file back gives the same logits"""
    import rama_amd
    cfg, w, g = load_case("synth_d64_h4")
    m = rama_amd.Model.synth(dev, to_rama_cfg(cfg), int(g["seed"]), rope=(g["freq_cis_real"], g["freq_cis_imag"]))
    out = tmp_path / "synth.bin"
    m.save(out)
    cfg2, w2 = O.read_checkpoint(out)
    assert cfg2 == cfg
    for t, _ in O.weight_shapes(cfg):
        assert np.array_equal(np.asarray(w2[t]).reshape(-1), np.asarray(w[t]).reshape(-1)), t
    m2 = rama_amd.Model.load(dev, out)
    a, b = rama_amd.Engine(dev, m), rama_amd.Engine(dev, m2)
    for pos, tok in enumerate(g["tokens"].tolist()[:6]):
        a.forward(tok, pos); b.forward(tok, pos)
        assert np.array_equal(a.logits(), b.logits())
    # a pipeline stage cannot be saved
    st = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 0, stage=rama_amd._lib.rama_stage(0, 1, 1, 0))
    with pytest.raises(rama_amd.RamaError):
        st.save(tmp_path / "stage.bin")
    for e in (a, b): e.free()
    for mm in (m, m2, st): mm.free()


@pytest.mark.parametrize("name", SYNTH_CASES)
def test_model_synth_equals_oracle_synth(dev, name):
    """weights generated in HBM by the fill kernel == the oracle's generator, bit for bit"""
    import rama_amd
    cfg, w, g = load_case(name)
    m = rama_amd.Model.synth(dev, to_rama_cfg(cfg), int(g["seed"]), rope=(g["freq_cis_real"], g["freq_cis_imag"]))
    for t, _ in O.weight_shapes(cfg):
        assert np.array_equal(m.tensor(t, w[t].size), np.asarray(w[t]).reshape(-1)), t
    m.free()


@pytest.mark.parametrize("name,graph", [("synth_d64_h4", False), ("synth_d64_h4", True), ("synth_d288_h6", True), ("ckpt_untied", False)])
def test_generate_greedy_tokens(dev, name, graph):
    """(C): generate() at T = 0: BOS at pos 0, forced prompt, then argmax; token ids equal the
    oracle's loop on the same weights (mod.rs:169-206)."""
    import rama_amd
    cfg, w, g = load_case(name)
    prompt = g["tokens"].tolist()[1:5]
    steps = min(cfg.seq_len, 24)
    want = O.Oracle(cfg, w).generate_greedy(prompt, steps)
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    got_host = rama_amd.generate(rcfg, prompt, 0.0, steps, 0.9, wv, rsv, dev, fused=True)
    assert got_host == want
    rs2 = rama_amd.RunState.from_config(rcfg, dev); rsv2 = rama_amd.RunStateView.from_rs(rs2)
    dev.lib.rama_set_graph_mode(dev.ctx, int(graph))
    got_dev = rama_amd.generate_greedy_device(rcfg, prompt, steps, wv, rsv2, dev)
    dev.lib.rama_set_graph_mode(dev.ctx, 0)
    assert got_dev == want
    # the chained loop (prompt positions through rama_prefill, which rounds RoPE / the cache rows
    # in a differently scheduled kernel) leaves the stepwise logits to within fp32 noise ...
    assert np.abs(dev.download(rsv2.logits) - dev.download(rsv.logits)).max() <= 1e-5
    # ... and bit for bit when the prompt also goes token by token
    rs3 = rama_amd.RunState.from_config(rcfg, dev); rsv3 = rama_amd.RunStateView.from_rs(rs3)
    rama_amd._lib.check(dev.lib.rama_set_tuning(dev.ctx, b"prefill", 0))
    try:
        assert rama_amd.generate_greedy_device(rcfg, prompt, steps, wv, rsv3, dev) == want
    finally:
        rama_amd._lib.check(dev.lib.rama_set_tuning(dev.ctx, b"prefill", 1))
    assert np.array_equal(dev.download(rsv3.logits), dev.download(rsv.logits))
    rs3.free()
    rs.free(); rs2.free(); ws.free()


def test_generate_with_topp_sampling_matches_oracle(dev):
    import rama_amd
    cfg, w, g = load_case("synth_d64_h4")
    u = 0.2721174359321594
    prompt, steps = g["tokens"].tolist()[1:3], 12
    orc = O.Oracle(cfg, w)
    token, want = 1, []
    for pos in range(steps):
        lo = orc.forward(token, pos)
        nxt = prompt[pos] if pos < len(prompt) else O.sample(lo, 1.0, 0.9, u)
        want.append(int(nxt)); token = nxt
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    got = rama_amd.generate(rcfg, prompt, 1.0, steps, 0.9, wv, rsv, dev, fused=True)
    assert got == want
    rs.free(); ws.free()


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("name,temperature,topp,u,prompt_len", [("synth_d64_h4", 1.0, 0.9, 0.2721174359321594, 2), ("synth_d288_h6", 0.8, 0.95, 0.6, 0),
                                                                ("ckpt_untied", 1.0, 0.9, 0.03743588924407959, 5), ("synth_d64_h4", 1.5, 0.5, 0.9, 3)])
def test_generate_sampled_on_device_matches_oracle(dev, name, temperature, topp, u, prompt_len, graph):
    """rama_generate with temperature != 0: the whole loop (prefill of the prompt, top-p sampling,
    cursor advance) stays on the device; tokens must equal forward() + Device::sample of the oracle"""
    import rama_amd
    cfg, w, g = load_case(name)
    steps = min(14, cfg.seq_len)
    prompt = g["tokens"].tolist()[1:1 + prompt_len]
    orc = O.Oracle(cfg, w)
    token, want = 1, []
    for pos in range(steps):
        lo = orc.forward(token, pos)
        nxt = prompt[pos] if pos < len(prompt) else O.sample(lo.copy(), temperature, topp, u)
        want.append(int(nxt)); token = nxt
    if name.startswith("ckpt"):
        m = rama_amd.Model.load(dev, GOLDEN / f"{name}.bin")
    else:
        m = rama_amd.Model.synth(dev, to_rama_cfg(cfg), int(g["seed"]), rope=(g["freq_cis_real"], g["freq_cis_imag"]))
    eng = rama_amd.Engine(dev, m)
    eng.set_graph_mode(graph)
    try:
        assert eng.generate(prompt, steps, temperature, topp, u) == want
        assert eng.generate(prompt, steps, temperature, topp, u) == want       # replay (graph cache, scratch reuse)
        assert eng.generate_greedy(prompt, steps) == eng.generate(prompt, steps, 0.0)
    finally:
        eng.set_graph_mode(False)
        eng.decode_sampler(0.0)
    eng.free(); m.free()


def test_stage_split_equals_whole(dev):
    """layer-pipeline stages on one device: [0,1) + [1,L) with x handed over == full forward"""
    import rama_amd
    from rama_amd._lib import rama_stage
    cfg, w, g = load_case("synth_d288_h6")
    rcfg = to_rama_cfg(cfg)
    seed, rope = int(g["seed"]), (g["freq_cis_real"], g["freq_cis_imag"])
    full = rama_amd.Engine(dev, rama_amd.Model.synth(dev, rcfg, seed, rope=rope))
    s0 = rama_amd.Engine(dev, rama_amd.Model.synth(dev, rcfg, seed, rama_stage(0, 1, 1, 0), rope))
    s1 = rama_amd.Engine(dev, rama_amd.Model.synth(dev, rcfg, seed, rama_stage(1, cfg.n_layers, 0, 1), rope))
    orc = O.Oracle(cfg, w)
    for pos, tok in enumerate(g["tokens"].tolist()[:10]):
        full.forward(tok, pos)
        s0.forward(tok, pos)
        s1.set_buffer("x", s0.buffer("x", cfg.dim))       # the inter-stage hand-off
        s1.forward(tok, pos)
        assert np.array_equal(full.logits(), s1.logits())
        assert np.abs(full.logits() - orc.forward(tok, pos)).max() <= LOGIT_ATOL
    for e in (full, s0, s1):
        e.free(); e.model.free()


@pytest.mark.parametrize("shape,solo", [
    ((64, 176, 2, 4, 4, 96, 300, True), -1),          # a quarter of a chunk per row
    ((288, 768, 3, 6, 6, 512, 64, True), -1),          # stories15M's layer: 2 chunks, the second nearly empty
    ((288, 768, 3, 6, 6, 512, 64, True), 1),           # ... one workgroup per CU
    ((768, 2048, 2, 12, 12, 300, 40, False), -1),      # stories110M's layer: 2-row W2 units, 8 chunks ahead, alone on the CU
    ((768, 2048, 2, 12, 12, 300, 40, False), 0),       # ... two workgroups per CU
    ((512, 1408, 2, 2, 2, 200, 24, True), -1),         # head_size 256 (64 lanes per cache row), W2 rows of 5.5 chunks
    ((1024, 2752, 1, 8, 8, 128, 16, True), -1),        # head_size 128; W2 rows wider than what is requested ahead (11 chunks)
    ((132, 360, 2, 3, 3, 77, 20, False), -1),          # head_size 44, ragged everything
])
def test_one_launch_stage_equals_separate_launches(dev, shape, solo):
    """layer_fused.hpp (tuning "fused", default for dim <= 1024): every layer and the classifier of a step in one launch, the
    phases chained through tagged vectors.  Against the oracle -- greedy tokens, logits, caches -- and against the separate
    launches: every run-state buffer the two leave behind (x, xb, hb, q, k, v, logits, caches) agrees within the bar.  Eager
    and replayed from a graph, and run twice over (the second pass meets the first pass's tags in every vector)."""
    import rama_amd
    from rama_amd._lib import check
    cfg = O.Config(*shape)
    w = S.synth_weights(cfg, seed=31)
    prompt, steps = [3, 1, 4], min(cfg.seq_len, 24)
    orc = O.Oracle(cfg, w)
    want = orc.generate_greedy(prompt, steps)
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    state = {}
    try:
        check(dev.lib.rama_set_tuning(dev.ctx, b"fused_solo", solo))
        for fused, graph in ((0, 0), (1, 0), (1, 1), (1, 0)):
            check(dev.lib.rama_set_tuning(dev.ctx, b"fused", fused))
            dev.lib.rama_set_graph_mode(dev.ctx, graph)
            got = rama_amd.generate_greedy_device(rcfg, prompt, steps, wv, rsv, dev)
            assert got == want, (fused, graph)
            bufs = {b: dev.download(getattr(rsv, b)) for b in ("x", "xb", "hb", "q", "k", "v", "logits", "key_cache", "value_cache")}
            assert np.abs(bufs["logits"] - orc.s["logits"]).max() <= LOGIT_ATOL
            for b in ("key_cache", "value_cache"):
                assert np.abs(bufs[b] - orc.s[b]).max() <= STATE_ATOL, b
            if fused == 0:
                state = bufs
            else:
                for b, v in bufs.items():
                    assert np.abs(v - state[b]).max() <= (LOGIT_ATOL if b == "logits" else STATE_ATOL), (b, fused, graph)
    finally:
        dev.lib.rama_set_graph_mode(dev.ctx, 0)
        check(dev.lib.rama_set_tuning(dev.ctx, b"fused", -1))
        check(dev.lib.rama_set_tuning(dev.ctx, b"fused_solo", -1))
    rs.free(); ws.free()


def test_one_launch_stage_on_layer_ranges(dev):
    """the one-launch stage on pipeline stages: [0,2) without classifier, [2,L) without embedding, x handed over, against
    the whole model's separate launches and the oracle"""
    import rama_amd
    from rama_amd._lib import rama_stage, check
    cfg, w, g = load_case("synth_d288_h6")
    rcfg = to_rama_cfg(cfg)
    seed, rope = int(g["seed"]), (g["freq_cis_real"], g["freq_cis_imag"])
    full = rama_amd.Engine(dev, rama_amd.Model.synth(dev, rcfg, seed, rope=rope))
    s0 = rama_amd.Engine(dev, rama_amd.Model.synth(dev, rcfg, seed, rama_stage(0, 2, 1, 0), rope))
    s1 = rama_amd.Engine(dev, rama_amd.Model.synth(dev, rcfg, seed, rama_stage(2, cfg.n_layers, 0, 1), rope))
    orc = O.Oracle(cfg, w)
    try:
        for pos, tok in enumerate(g["tokens"].tolist()[:12]):
            check(dev.lib.rama_set_tuning(dev.ctx, b"fused", 0))
            full.forward(tok, pos)
            check(dev.lib.rama_set_tuning(dev.ctx, b"fused", 1))
            s0.forward(tok, pos)
            s1.set_buffer("x", s0.buffer("x", cfg.dim))
            s1.forward(tok, pos)
            assert np.abs(full.logits() - s1.logits()).max() <= LOGIT_ATOL
            assert np.abs(s1.logits() - orc.forward(tok, pos)).max() <= LOGIT_ATOL
    finally:
        check(dev.lib.rama_set_tuning(dev.ctx, b"fused", -1))
    for e in (full, s0, s1):
        e.free(); e.model.free()


def test_forward_argument_errors(dev):
    import rama_amd
    cfg, w, g = load_case("synth_d64_h4")
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    with pytest.raises(rama_amd.RamaError):      # pos beyond the cache (reference: no check, UB)
        rama_amd.forward_fused(rcfg, wv, rsv, 1, cfg.seq_len, dev)
    with pytest.raises(rama_amd.RamaError):
        rama_amd.forward_fused(rcfg, wv, rsv, cfg.vocab_size, 0, dev)
    gqa = rama_amd.Config(rcfg.dim, rcfg.hidden_dim, rcfg.n_layers, rcfg.n_heads, 2, rcfg.vocab_size, rcfg.seq_len, True)
    with pytest.raises(rama_amd.RamaError):      # n_kv_heads != n_heads: outside the reference path
        rama_amd.forward_fused(gqa, wv, rsv, 1, 0, dev)
    rs.free(); ws.free()


# ------------------------------------------------------------------ full-size properties

def test_7b_classifier_one_hot_column_exact(dev):
    """BASELINE size: the 32000 x 4096 classifier matvec (524 MB) on synthetic weights; with
    x = e_j the result must be column j bit for bit, recomputed from the generator."""
    import rama_amd
    V, d = 32000, 4096
    sc = np.float32(0.02 / S.IH4_STD)
    Wd = dev.alloc(V * d)
    rama_amd._lib.check(dev.lib.rama_fill_synth(dev.ctx, Wd.ptr, V * d, 0, 12, 0, sc, np.float32(0.0)))
    o = rama_amd.MutView(dev.alloc(V))
    for j in (0, 1023, 1024, 4095):
        x = np.zeros(d, np.float32); x[j] = 1.0
        xs = dev.allocate(x)
        dev.matmul(o, rama_amd.View(Wd), rama_amd.View(xs), d, V, 1)
        col = synth_at(np.arange(V, dtype=np.uint64) * np.uint64(d) + np.uint64(j), 0, 12, sc)
        assert np.array_equal(dev.download(o), col)
        xs.free()
    Wd.free()


def test_7b_layer_matvec_linearity_and_rows(dev):
    """11008 x 4096 and 4096 x 11008 at full size: sampled rows against float64 dot products
    regenerated from the synthetic generator, and linearity W(ax+by) = aWx + bWy."""
    import rama_amd
    sc = np.float32(0.02 / S.IH4_STD)
    rng = np.random.default_rng(5)
    for rows, width, tag in ((11008, 4096, 8), (4096, 11008, 9)):
        Wd = dev.alloc(rows * width)
        rama_amd._lib.check(dev.lib.rama_fill_synth(dev.ctx, Wd.ptr, rows * width, 1, tag, 0, sc, np.float32(0.0)))
        x, y = rnd_vec(width, 1), rnd_vec(width, 2)
        ox, oy, oz = (rama_amd.MutView(dev.alloc(rows)) for _ in range(3))
        dev.matmul(ox, rama_amd.View(Wd), rama_amd.View(dev.allocate(x)), width, rows, 1)
        dev.matmul(oy, rama_amd.View(Wd), rama_amd.View(dev.allocate(y)), width, rows, 1)
        dev.matmul(oz, rama_amd.View(Wd), rama_amd.View(dev.allocate((2 * x - 3 * y).astype(np.float32))), width, rows, 1)
        gx, gy, gz = dev.download(ox), dev.download(oy), dev.download(oz)
        np.testing.assert_allclose(gz, 2 * gx - 3 * gy, atol=2e-5)
        for r in rng.integers(0, rows, 16).tolist() + [0, rows - 1]:
            wr = synth_at(np.uint64(r) * np.uint64(width) + np.arange(width, dtype=np.uint64), 1, tag, sc)
            assert abs(float(gx[r]) - float(wr.astype(np.float64) @ x.astype(np.float64))) <= 2e-6
        Wd.free()


def rnd_vec(n, seed):
    return np.random.default_rng(seed).standard_normal(n).astype(np.float32)


# ------------------------------------------------------------------ opt-in launch structures

@pytest.mark.parametrize("key", [b"merge", b"small_attn", b"solo", b"fused"])
@pytest.mark.parametrize("name,graph", [("synth_d64_h4", False), ("synth_d288_h6", False), ("synth_d288_h6", True),
                                        ("synth_d768_h12", False), ("ckpt_untied", False), ("synth_d128_h1", False),
                                        ("synth_7bshape_l1", False)])
def test_launch_structures_equal_default_path(dev, name, graph, key):
    """the opt-in launch structures -- rama_set_tuning("merge", 1): attention + Wo as one launch
    (attn_wo.hpp); ("small_attn", 1): 4-wave attention workgroups; ("solo", 1): one wave per row group
    in every matvec; ("fused", 0): none of them and no one-launch stage either (layer_fused.hpp takes the narrow models
    by default and is switched off for the other three) -- must give the oracle's greedy tokens, logits within the bar,
    the same KV cache."""
    import rama_amd
    cfg, w, g = load_case(name)
    prompt = g["tokens"].tolist()[1:4]
    steps = min(cfg.seq_len, 20)
    orc = O.Oracle(cfg, w)
    want = orc.generate_greedy(prompt, steps)
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    rama_amd._lib.check(dev.lib.rama_set_tuning(dev.ctx, b"fused", 0))
    if key != b"fused":
        rama_amd._lib.check(dev.lib.rama_set_tuning(dev.ctx, key, 1))
    dev.lib.rama_set_graph_mode(dev.ctx, int(graph))
    try:
        got = rama_amd.generate_greedy_device(rcfg, prompt, steps, wv, rsv, dev)
    finally:
        dev.lib.rama_set_graph_mode(dev.ctx, 0)
        rama_amd._lib.check(dev.lib.rama_set_tuning(dev.ctx, key, -1))
        rama_amd._lib.check(dev.lib.rama_set_tuning(dev.ctx, b"fused", -1))
    assert got == want
    assert np.abs(dev.download(rsv.logits) - orc.s["logits"]).max() <= LOGIT_ATOL
    for buf in ("key_cache", "value_cache"):
        assert np.abs(dev.download(getattr(rsv, buf)) - orc.s[buf]).max() <= STATE_ATOL
    rs.free(); ws.free()


@pytest.mark.parametrize("graph_steps,merge,steps", [(4, 0, 290), (4, -1, 150), (3, 0, 271), (8, 0, 262), (1, 0, 262)])
def test_multi_step_graphs_across_the_attention_variant_boundary(dev, graph_steps, merge, steps):
    """rama_set_tuning("graph_steps", M): M decode steps captured per hipGraph (the cursor lives on the
    device).  With attention as its own launch (merge 0) the launch structure changes at position 256
    (8-wave attention below it): a group of M steps must never straddle that boundary, the tail
    shorter than M runs step by step, and the tokens are the oracle's throughout."""
    import rama_amd
    from rama_amd._lib import check
    cfg = O.Config(64, 176, 2, 4, 4, 96, 300, True)
    w = S.synth_weights(cfg, seed=21)
    prompt = [5, 9, 33]
    want = O.Oracle(cfg, w).generate_greedy(prompt, steps)
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    check(dev.lib.rama_set_tuning(dev.ctx, b"graph_steps", graph_steps))
    check(dev.lib.rama_set_tuning(dev.ctx, b"merge", merge))
    check(dev.lib.rama_set_tuning(dev.ctx, b"fused", 0))       # the separate launches are what this test is about
    dev.lib.rama_set_graph_mode(dev.ctx, 1)
    try:
        got = rama_amd.generate_greedy_device(rcfg, prompt, steps, wv, rsv, dev)
    finally:
        dev.lib.rama_set_graph_mode(dev.ctx, 0)
        check(dev.lib.rama_set_tuning(dev.ctx, b"graph_steps", -1))
        check(dev.lib.rama_set_tuning(dev.ctx, b"merge", -1))
        check(dev.lib.rama_set_tuning(dev.ctx, b"fused", -1))
    assert got == want
    rs.free(); ws.free()


@pytest.mark.parametrize("fused", [0, -1])
def test_forward_in_graph_mode_replays_the_same_launches(dev, fused):
    """rama_forward with rama_set_graph_mode(1): the step is captured once per (state, stage, attention
    variant) and replayed -- token and position travel through the device cursor -- and must leave
    bit-identical logits and caches, across the position where the launch structure changes (256
    with attention as its own launch) and for two states in turn.  fused -1: the one-launch stage, whose
    hand-off vectors the two states share (the epoch moves on with every replay)"""
    import rama_amd
    from rama_amd._lib import check
    cfg = O.Config(64, 176, 2, 4, 4, 96, 300, True)
    w = S.synth_weights(cfg, seed=23)
    rng = np.random.default_rng(5)
    toks = [1] + [int(t) for t in rng.integers(0, cfg.vocab_size, 269)]
    check(dev.lib.rama_set_tuning(dev.ctx, b"merge", 0))
    check(dev.lib.rama_set_tuning(dev.ctx, b"fused", fused))
    try:
        rcfg, ws, wv, rs_a, rsv_a = gpu_views(dev, cfg, w)
        _, ws_b, wv_b, rs_b, rsv_b = gpu_views(dev, cfg, w)
        _, ws_c, wv_c, rs_c, rsv_c = gpu_views(dev, cfg, w)
        check_at = {pos for pos in range(len(toks)) if pos % 37 == 0} | {255, 256, 257, len(toks) - 1}
        eager = {}
        for pos, t in enumerate(toks):                    # eager first: leaving graph mode drops the captured graphs
            rama_amd.forward_fused(rcfg, wv, rsv_a, t, pos, dev)
            if pos in check_at: eager[pos] = dev.download(rsv_a.logits)
        dev.lib.rama_set_graph_mode(dev.ctx, 1)
        for pos, t in enumerate(toks):                    # two states in turn: two graphs per variant, replayed ~135 times each
            rama_amd.forward_fused(rcfg, wv_b, rsv_b, t, pos, dev)
            rama_amd.forward_fused(rcfg, wv_c, rsv_c, t, pos, dev)
            if pos in check_at:
                assert np.array_equal(eager[pos], dev.download(rsv_b.logits)) and np.array_equal(eager[pos], dev.download(rsv_c.logits)), pos
        for buf in ("key_cache", "value_cache"):
            assert np.array_equal(dev.download(getattr(rsv_a, buf)), dev.download(getattr(rsv_b, buf)))
    finally:
        dev.lib.rama_set_graph_mode(dev.ctx, 0)
        check(dev.lib.rama_set_tuning(dev.ctx, b"merge", -1))
        check(dev.lib.rama_set_tuning(dev.ctx, b"fused", -1))
    for r in (rs_a, rs_b, rs_c, ws, ws_b, ws_c):
        r.free()


# ------------------------------------------------------------------ long contexts: split-T attention

def _long_ctx_case(n_heads, hs, seq_len=2048, seed=11):
    """one-layer synthetic model with a pre-filled KV cache (random, O(1) entries)"""
    dim = n_heads * hs
    cfg = O.Config(dim, 2 * dim, 1, n_heads, n_heads, 64, seq_len, False)
    w = S.synth_weights(cfg, seed)
    rng = np.random.default_rng(seed)
    kc = rng.standard_normal(seq_len * dim).astype(np.float32)
    vc = rng.standard_normal(seq_len * dim).astype(np.float32)
    return cfg, w, kc, vc


@pytest.mark.parametrize("split_pos", [256, -1, 1 << 30])
@pytest.mark.parametrize("n_heads,hs", [(2, 128), (4, 64), (6, 48)])
def test_split_t_attention_long_context(dev, n_heads, hs, split_pos):
    """long contexts through both attention variants: split-T (n_heads x nsplit workgroups +
    combine) forced from position 256, the default choice by model size (-1), and the
    single-workgroup kernel only; all must give the oracle's logits (one softmax over ALL timesteps)."""
    import rama_amd
    from rama_amd._lib import check
    cfg, w, kc, vc = _long_ctx_case(n_heads, hs)
    orc = O.Oracle(cfg, w)
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    check(dev.lib.rama_set_tuning(dev.ctx, b"split_pos", split_pos))
    try:
        for pos in (255, 256, 257, 1000, 2047):
            orc.s["key_cache"][:] = kc; orc.s["value_cache"][:] = vc
            dev.upload_into(rsv.key_cache, kc); dev.upload_into(rsv.value_cache, vc)
            lo = orc.forward(5, pos).copy()
            rama_amd.forward_fused(rcfg, wv, rsv, 5, pos, dev)
            lg = dev.download(rsv.logits)
            assert np.abs(lg - lo).max() <= LOGIT_ATOL, (pos, float(np.abs(lg - lo).max()))
            # the freshly appended cache rows agree too
            d = cfg.dim
            assert np.abs(dev.download(rsv.key_cache)[pos * d:(pos + 1) * d] - orc.s["key_cache"][pos * d:(pos + 1) * d]).max() <= STATE_ATOL
    finally:
        check(dev.lib.rama_set_tuning(dev.ctx, b"split_pos", -1))
    rs.free(); ws.free()


@pytest.mark.parametrize("n_heads,hs", [(2, 128), (3, 64)])
def test_context_longer_than_the_score_buffer(dev, n_heads, hs):
    """seq_len 16384: more timesteps than any single-workgroup attention variant's 64 KiB score buffer holds.  Positions
    below 256 take the fewer-wave variant (its buffer is sized for what fits, not for seq_len), the rest split-T"""
    import rama_amd
    cfg, w, kc, vc = _long_ctx_case(n_heads, hs, seq_len=16384, seed=3)
    orc = O.Oracle(cfg, w)
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    for pos in (0, 100, 300, 9000, 16383):
        orc.s["key_cache"][:] = kc; orc.s["value_cache"][:] = vc
        dev.upload_into(rsv.key_cache, kc); dev.upload_into(rsv.value_cache, vc)
        lo = orc.forward(5, pos).copy()
        rama_amd.forward_fused(rcfg, wv, rsv, 5, pos, dev)
        lg = dev.download(rsv.logits)
        assert np.abs(lg - lo).max() <= LOGIT_ATOL, (pos, float(np.abs(lg - lo).max()))
    rs.free(); ws.free()


@pytest.mark.parametrize("graph", [False, True])
def test_decode_across_split_threshold(dev, graph):
    """a chained greedy run from pos 242 to 272 switches attention variant (and hipGraph) at 256"""
    import ctypes as C
    import rama_amd
    from rama_amd._lib import S_FIELDS, check, rama_run_state
    cfg, w, kc, vc = _long_ctx_case(2, 128)
    start, steps = 242, 30
    orc = O.Oracle(cfg, w)
    orc.s["key_cache"][:] = kc; orc.s["value_cache"][:] = vc
    token, want = 7, []
    for pos in range(start, start + steps):
        token = O.argmax(orc.forward(token, pos))
        want.append(int(token))
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    dev.upload_into(rsv.key_cache, kc); dev.upload_into(rsv.value_cache, vc)
    rstate = rama_run_state(*[getattr(rsv, k).ptr for k in S_FIELDS])
    dev.lib.rama_set_graph_mode(dev.ctx, int(graph))
    try:
        check(dev.lib.rama_decode_begin(dev.ctx, 7, start, None, 0))
        check(dev.lib.rama_decode_steps(dev.ctx, C.byref(rcfg.c()), C.byref(wv.c()), C.byref(rstate), steps))
        out = (C.c_int32 * steps)(); n = C.c_int()
        check(dev.lib.rama_decode_tokens(dev.ctx, out, steps, C.byref(n)))
    finally:
        dev.lib.rama_set_graph_mode(dev.ctx, 0)
    assert [int(v) for v in out[:n.value]] == want
    assert np.abs(dev.download(rsv.logits) - orc.s["logits"]).max() <= LOGIT_ATOL
    rs.free(); ws.free()


def test_decode_steps_needs_begin_and_bounds(dev):
    import ctypes as C
    import rama_amd
    from rama_amd._lib import S_FIELDS, check, rama_run_state
    cfg, w, g = load_case("synth_d64_h4")
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    rstate = rama_run_state(*[getattr(rsv, k).ptr for k in S_FIELDS])
    check(dev.lib.rama_decode_begin(dev.ctx, 1, cfg.seq_len - 2, None, 0))
    with pytest.raises(rama_amd.RamaError):      # 3 steps from seq_len - 2 would overrun the cache
        check(dev.lib.rama_decode_steps(dev.ctx, C.byref(rcfg.c()), C.byref(wv.c()), C.byref(rstate), 3))
    rs.free(); ws.free()


# ------------------------------------------------------------------ BASELINE.json shapes (full width, full vocab)

FULL_SHAPES = {   # dim, hidden, layers, heads, vocab, seq_len, shared classifier
    "stories15M": (288, 768, 6, 6, 32000, 256, True),
    "stories110M": (768, 2048, 12, 12, 32000, 1024, True),
    "llama2-7B-2layers": (4096, 11008, 2, 32, 32000, 2048, False),
}


@pytest.mark.parametrize("shape", list(FULL_SHAPES))
def test_full_shape_logits_vs_oracle(dev, shape):
    """BASELINE.json configs at their real widths and vocabulary (synthetic weights generated in
    HBM, bit-identical to the oracle's): the generate() loop on 'once upon a time', logits of
    every step within 1e-4 of the oracle, greedy tokens identical."""
    import rama_amd
    d, h, L, H, V, seq, shared = FULL_SHAPES[shape]
    cfg = O.Config(d, h, L, H, H, V, seq, shared)
    rope = S.rope_tables(seq, d // H)
    w = S.synth_weights(cfg, 0, rope=rope)
    orc = O.Oracle(cfg, w)
    rcfg = to_rama_cfg(cfg)
    model = rama_amd.Model.synth(dev, rcfg, 0, rope=rope)
    eng = rama_amd.Engine(dev, model)
    prompt = [10646, 2501, 263, 931]          # Rama-BPE of 'once upon a time'
    steps = 10 if d < 4096 else 6
    token, worst = 1, 0.0
    for pos in range(steps):
        lo = orc.forward(token, pos)
        eng.forward(token, pos)
        worst = max(worst, float(np.abs(eng.logits() - lo).max()))
        token = prompt[pos] if pos < len(prompt) else O.argmax(lo)
    assert worst <= LOGIT_ATOL, f"{shape}: {worst:.3e}"
    eng2 = rama_amd.Engine(dev, model)
    assert eng2.generate_greedy(prompt, steps) == O.Oracle(cfg, w).generate_greedy(prompt, steps)
    eng.free(); eng2.free(); model.free()


@pytest.mark.parametrize("shape", ["stories15M", "stories110M"])
def test_stories_fast_mode_readme_length(dev, shape):
    """BASELINE config 2 / 3 as the README ran them (README.md:80-83: a 200-token generation on 'once upon a time'): the
    FAST mode -- what the stories tok/s figures of bench.py are quoted for, the whole token in one launch (layer_fused.hpp) --
    against the oracle at every one of the 200 positions: |dlogit| <= 1e-4, greedy tokens identical, and the device-chained
    generate() (one-launch stage + sampler, replayed from hipGraphs) producing the oracle's 200 tokens.  The per-position
    record goes to gpurun_out/r05_parity_<shape>_200pos.json (copied to profiles/ by the builder)."""
    import json
    from pathlib import Path
    import rama_amd
    d, h, L, H, V, seq, shared = FULL_SHAPES[shape]
    n_pos = min(200, seq)
    cfg = O.Config(d, h, L, H, H, V, seq, shared)
    rope = S.rope_tables(seq, d // H)
    w = S.synth_weights(cfg, 0, rope=rope)
    orc = O.Oracle(cfg, w)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 0, rope=rope)
    eng = rama_amd.Engine(dev, model)
    prompt = [10646, 2501, 263, 931]
    token, rows, toks_cpu, toks_hip = 1, [], [], []
    for pos in range(n_pos):
        lo = orc.forward(token, pos)
        eng.forward(token, pos)
        lg = eng.logits()
        rows.append({"pos": pos, "token": int(token), "hip_fast_vs_oracle": float(np.abs(lg - lo).max())})
        toks_cpu.append(int(O.argmax(lo)))
        toks_hip.append(int(np.flatnonzero(lg == lg.max())[-1]))
        token = prompt[pos] if pos < len(prompt) else toks_cpu[-1]
    eng2 = rama_amd.Engine(dev, model)
    eng2.set_graph_mode(True)
    try:
        chained = eng2.generate_greedy(prompt, n_pos)
    finally:
        eng2.set_graph_mode(False)
    want = O.Oracle(cfg, w).generate_greedy(prompt, n_pos)
    out = {"shape": f"{shape} fp32, synthetic weights seed 0 (bit-identical on both sides)", "mode": "fast (one launch per token: csrc/layer_fused.hpp)",
           "prompt": "BOS + 'once upon a time' (Rama-BPE), greedy continuation chosen by the oracle", "positions": n_pos, "bar": LOGIT_ATOL,
           "worst_hip_fast_vs_oracle": max(r["hip_fast_vs_oracle"] for r in rows),
           "positions_over_bar": [r["pos"] for r in rows if r["hip_fast_vs_oracle"] > LOGIT_ATOL],
           "greedy_tokens_equal": toks_cpu == toks_hip, "chained_generate_equals_oracle": chained == want, "per_position": rows}
    path = Path(__file__).resolve().parent.parent / "gpurun_out" / f"r05_parity_{shape}_200pos.json"
    try:
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(json.dumps(out, indent=1))
    except OSError:
        pass
    eng.free(); eng2.free(); model.free()
    assert out["worst_hip_fast_vs_oracle"] <= LOGIT_ATOL, (shape, out["worst_hip_fast_vs_oracle"], out["positions_over_bar"][:8])
    assert toks_cpu == toks_hip
    assert chained == want


# ------------------------------------------------------------------ batched-prompt prefill (f3)

def _prefill(dev, rcfg, wv, rsv, tokens, pos0):
    import ctypes as C
    from rama_amd._lib import S_FIELDS, check, rama_run_state
    rstate = rama_run_state(*[getattr(rsv, k).ptr for k in S_FIELDS])
    arr = (C.c_int32 * len(tokens))(*tokens)
    check(dev.lib.rama_prefill(dev.ctx, C.byref(rcfg.c()), C.byref(wv.c()), C.byref(rstate), arr, len(tokens), pos0), "rama_prefill")


@pytest.mark.parametrize("name,n_tokens,pos0", [("synth_d64_h4", 1, 0), ("synth_d64_h4", 5, 0), ("synth_d64_h4", 8, 0),
                                                ("synth_d64_h4", 9, 0), ("synth_d288_h6", 17, 0), ("synth_d288_h6", 11, 6),
                                                ("synth_d768_h12", 16, 0), ("ckpt_untied", 13, 2), ("synth_d128_h1", 10, 3),
                                                ("synth_7bshape_l1", 8, 0)])
def test_prefill_equals_sequential_forward(dev, name, n_tokens, pos0):
    """rama_prefill(tokens, pos0) must leave the state n sequential forward() calls leave:
    KV-cache rows, the residual stream and the logits of the last position (oracle as referee)."""
    import rama_amd
    cfg, w, g = load_case(name)
    toks = g["tokens"].tolist()
    n_tokens = min(n_tokens, len(toks) - pos0)
    orc = O.Oracle(cfg, w)
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    for pos in range(pos0):                       # positions before the prefill: plain decode on both sides
        orc.forward(toks[pos], pos)
        rama_amd.forward_fused(rcfg, wv, rsv, toks[pos], pos, dev)
    for i in range(n_tokens):
        lo = orc.forward(toks[pos0 + i], pos0 + i)
    _prefill(dev, rcfg, wv, rsv, toks[pos0:pos0 + n_tokens], pos0)
    assert np.abs(dev.download(rsv.logits) - lo).max() <= LOGIT_ATOL
    for buf in ("key_cache", "value_cache"):
        assert np.abs(dev.download(getattr(rsv, buf)) - orc.s[buf]).max() <= STATE_ATOL
    # and decoding simply continues from there
    nxt = O.argmax(lo)
    if pos0 + n_tokens < cfg.seq_len:
        lo2 = orc.forward(nxt, pos0 + n_tokens)
        rama_amd.forward_fused(rcfg, wv, rsv, nxt, pos0 + n_tokens, dev)
        assert np.abs(dev.download(rsv.logits) - lo2).max() <= LOGIT_ATOL
    rs.free(); ws.free()


@pytest.mark.parametrize("n_tokens,pos0", [(17, 0), (33, 0), (64, 0), (65, 3), (150, 0)])
def test_prefill_long_prompts_cross_tile_and_pass_boundaries(dev, n_tokens, pos0):
    """prompts longer than one 16-token MFMA tile / one 64-token pass: every tile count (PT = 1, 2, 4),
    a partly filled last tile, several passes and a non-zero start position give the state sequential
    forward() calls leave (oracle as referee)"""
    import rama_amd
    cfg = O.Config(128, 352, 2, 4, 4, 256, 160, False)
    w = S.synth_weights(cfg, seed=11)
    rng = np.random.default_rng(n_tokens)
    toks = [1] + [int(t) for t in rng.integers(0, cfg.vocab_size, pos0 + n_tokens - 1)]
    orc = O.Oracle(cfg, w)
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    for pos in range(pos0):
        orc.forward(toks[pos], pos)
        rama_amd.forward_fused(rcfg, wv, rsv, toks[pos], pos, dev)
    for i in range(n_tokens):
        lo = orc.forward(toks[pos0 + i], pos0 + i)
    _prefill(dev, rcfg, wv, rsv, toks[pos0:pos0 + n_tokens], pos0)
    assert np.abs(dev.download(rsv.logits) - lo).max() <= LOGIT_ATOL
    assert np.abs(dev.download(rsv.x) - orc.s["xb"]).max() <= STATE_ATOL      # infer.rs:49: the residual stream ends up in xb on the CPU path
    for buf in ("key_cache", "value_cache"):
        assert np.abs(dev.download(getattr(rsv, buf)) - orc.s[buf]).max() <= STATE_ATOL
    nxt = O.argmax(lo)
    lo2 = orc.forward(nxt, pos0 + n_tokens)
    rama_amd.forward_fused(rcfg, wv, rsv, nxt, pos0 + n_tokens, dev)
    assert np.abs(dev.download(rsv.logits) - lo2).max() <= LOGIT_ATOL
    rs.free(); ws.free()


@pytest.mark.parametrize("n_heads,n_tokens,pos0", [(2, 150, 0), (2, 65, 3), (1, 100, 21), (2, 610, 7), (1, 530, 0)])
def test_prefill_attention_tiles_long_contexts(dev, n_heads, n_tokens, pos0):
    """the MFMA tile attention of a prefill pass (csrc/prefill_attn.hpp) at head sizes 64 and 128: several
    key tiles per wave, start positions that are no multiple of 16 (a query tile then ends inside a
    key tile), the 8-wave variant of contexts >= 512, and the per-query kernels as the other arm"""
    import rama_amd
    from rama_amd._lib import check
    cfg = O.Config(128, 352, 2, n_heads, n_heads, 256, 640, False)
    w = S.synth_weights(cfg, seed=13)
    rng = np.random.default_rng(n_tokens)
    toks = [1] + [int(t) for t in rng.integers(0, cfg.vocab_size, pos0 + n_tokens - 1)]
    orc = O.Oracle(cfg, w)
    for pos in range(pos0 + n_tokens):
        lo = orc.forward(toks[pos], pos)
    outs = {}
    for mode in (1, 0):
        rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
        for pos in range(pos0):
            rama_amd.forward_fused(rcfg, wv, rsv, toks[pos], pos, dev)
        check(dev.lib.rama_set_tuning(dev.ctx, b"prefill_attn", mode))
        try:
            _prefill(dev, rcfg, wv, rsv, toks[pos0:pos0 + n_tokens], pos0)
        finally:
            check(dev.lib.rama_set_tuning(dev.ctx, b"prefill_attn", 1))
        outs[mode] = dev.download(rsv.logits)
        assert np.abs(outs[mode] - lo).max() <= LOGIT_ATOL, mode
        assert np.abs(dev.download(rsv.x) - orc.s["xb"]).max() <= STATE_ATOL
        for buf in ("key_cache", "value_cache"):
            assert np.abs(dev.download(getattr(rsv, buf)) - orc.s[buf]).max() <= STATE_ATOL
        rs.free(); ws.free()
    assert np.abs(outs[1] - outs[0]).max() <= 2e-5


@pytest.mark.parametrize("shape,n_tokens", [((288, 768, 2, 6, 512, 96), 5), ((288, 768, 2, 6, 512, 96), 40), ((128, 352, 3, 2, 256, 160), 17),
                                            ((128, 352, 3, 2, 256, 160), 70), ((768, 2048, 1, 12, 1024, 80), 33)])
def test_tile_order_weight_copy_gives_identical_results(dev, shape, n_tokens):
    """a rama_model keeps its matrices once more in MFMA tile order (csrc/model.hip make_tiled) and
    rama_prefill / rama_decode_batch read that copy (one contiguous 1-KiB weight read per wave, no lane
    permute): same MFMA sequence, so logits and caches are bit-identical to the row-major path
    (rama_set_tuning "tiled" = 0), and within the bar of the oracle"""
    import ctypes as C
    import rama_amd
    from rama_amd._lib import check, rama_run_state
    dim, hidden, L, H, V, seq = shape
    ocfg = O.Config(dim, hidden, L, H, H, V, seq, True)
    cfg = rama_amd.Config(dim, hidden, L, H, H, V, seq, True)
    rope = S.rope_tables(seq, dim // H)
    w = S.synth_weights(ocfg, seed=31, rope=rope)
    model = rama_amd.Model.synth(dev, cfg, 31, rope=rope)
    rng = np.random.default_rng(n_tokens)
    toks = [1] + [int(t) for t in rng.integers(0, V, n_tokens - 1)]
    orc = O.Oracle(ocfg, w)
    for pos, t in enumerate(toks):
        lo = orc.forward(t, pos)
    arr = (C.c_int32 * n_tokens)(*toks)
    outs = {}
    check(dev.lib.rama_set_tuning(dev.ctx, b"prefill_tok", 64))      # 128-token passes split K differently: their own test below
    for mode in (1, 0):
        eng = rama_amd.Engine(dev, model)
        check(dev.lib.rama_set_tuning(dev.ctx, b"tiled", mode))
        try:
            check(dev.lib.rama_prefill(dev.ctx, C.byref(model.ccfg), C.byref(model.weights), C.byref(eng.state), arr, n_tokens, 0), "rama_prefill")
            lg = eng.logits()
            kc = eng.buffer("key_cache", L * seq * dim)
            # one more token for 3 sequences at once: the classifier GEMM reads wcls's tile-order copy too
            engs = [eng] + [rama_amd.Engine(dev, model) for _ in range(2)]
            for e in engs[1:]:
                check(dev.lib.rama_prefill(dev.ctx, C.byref(model.ccfg), C.byref(model.weights), C.byref(e.state), arr, n_tokens, 0), "rama_prefill")
            states = (rama_run_state * 3)(*[e.state for e in engs])
            nxt = (C.c_int32 * 3)(7, 11, 13)
            poss = (C.c_int32 * 3)(n_tokens, n_tokens, n_tokens)
            check(dev.lib.rama_decode_batch(dev.ctx, C.byref(model.ccfg), C.byref(model.weights), states, nxt, poss, 3), "rama_decode_batch")
            outs[mode] = (lg, kc, [e.logits() for e in engs])
            for e in engs:
                e.free()
        finally:
            check(dev.lib.rama_set_tuning(dev.ctx, b"tiled", 1))
    check(dev.lib.rama_set_tuning(dev.ctx, b"prefill_tok", 128))
    assert np.abs(outs[1][0] - lo).max() <= LOGIT_ATOL
    assert np.array_equal(outs[1][0], outs[0][0]) and np.array_equal(outs[1][1], outs[0][1])
    for a, b in zip(outs[1][2], outs[0][2]):
        assert np.array_equal(a, b)
    model.free()


@pytest.mark.parametrize("shape,n_tokens,pos0", [((128, 352, 2, 2, 256, 320), 65, 0), ((128, 352, 2, 2, 256, 320), 90, 0), ((128, 352, 2, 2, 256, 320), 100, 3), ((128, 352, 2, 2, 256, 320), 128, 0),
                                                 ((128, 352, 2, 2, 256, 320), 129, 0), ((128, 352, 2, 1, 256, 320), 300, 5), ((288, 768, 2, 6, 512, 256), 113, 0),
                                                 ((768, 2048, 1, 12, 1024, 256), 128, 17)])
def test_prefill_128_positions_per_weight_pass(dev, shape, n_tokens, pos0):
    """with the tile-order weight copies a prefill pass takes up to 128 positions (8 token tiles per wave, one
    K-block per step, the cross-wave fold in two rounds; as many token tiles as the pass has): every tile count 5..8, a partly filled last tile, several
    passes and a non-zero start give the state sequential forward() calls leave (oracle as referee), and
    the 64-position passes (rama_set_tuning "prefill_tok" = 64) agree to rounding"""
    import ctypes as C
    import rama_amd
    from rama_amd._lib import check
    dim, hidden, L, H, V, seq = shape
    ocfg = O.Config(dim, hidden, L, H, H, V, seq, True)
    cfg = rama_amd.Config(dim, hidden, L, H, H, V, seq, True)
    rope = S.rope_tables(seq, dim // H)
    w = S.synth_weights(ocfg, seed=37, rope=rope)
    model = rama_amd.Model.synth(dev, cfg, 37, rope=rope)
    rng = np.random.default_rng(n_tokens)
    toks = [1] + [int(t) for t in rng.integers(0, V, pos0 + n_tokens - 1)]
    orc = O.Oracle(ocfg, w)
    for pos, t in enumerate(toks):
        lo = orc.forward(t, pos)
    lo = lo.copy()
    want = {k: orc.s[k].copy() for k in ("key_cache", "value_cache", "xb")}
    nxt = O.argmax(lo)
    lo2 = orc.forward(nxt, pos0 + n_tokens).copy()
    arr = (C.c_int32 * n_tokens)(*toks[pos0:])
    outs = {}
    for per_pass in (128, 64):
        eng = rama_amd.Engine(dev, model)
        for pos in range(pos0):
            eng.forward(toks[pos], pos)
        check(dev.lib.rama_set_tuning(dev.ctx, b"prefill_tok", per_pass))
        try:
            check(dev.lib.rama_prefill(dev.ctx, C.byref(model.ccfg), C.byref(model.weights), C.byref(eng.state), arr, n_tokens, pos0), "rama_prefill")
        finally:
            check(dev.lib.rama_set_tuning(dev.ctx, b"prefill_tok", 128))
        outs[per_pass] = (eng.logits(), eng.buffer("key_cache", L * seq * dim), eng.buffer("value_cache", L * seq * dim), eng.buffer("x", dim))
        assert np.abs(outs[per_pass][0] - lo).max() <= LOGIT_ATOL, per_pass
        assert np.abs(outs[per_pass][1] - want["key_cache"]).max() <= STATE_ATOL, per_pass
        assert np.abs(outs[per_pass][2] - want["value_cache"]).max() <= STATE_ATOL, per_pass
        assert np.abs(outs[per_pass][3] - want["xb"]).max() <= STATE_ATOL, per_pass
        # and decoding simply continues from there
        eng.forward(nxt, pos0 + n_tokens)
        assert np.abs(eng.logits() - lo2).max() <= LOGIT_ATOL, per_pass
        eng.free()
    assert np.abs(outs[128][0] - outs[64][0]).max() <= 2e-5
    model.free()


def test_prefill_argument_errors(dev):
    import rama_amd
    cfg, w, g = load_case("synth_d64_h4")
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    with pytest.raises(rama_amd.RamaError):
        _prefill(dev, rcfg, wv, rsv, [1] * 5, cfg.seq_len - 3)      # runs past seq_len
    with pytest.raises(rama_amd.RamaError):
        _prefill(dev, rcfg, wv, rsv, [cfg.vocab_size], 0)           # token outside the vocabulary
    rs.free(); ws.free()


def test_concurrent_contexts_share_weights(dev):
    """the server pattern (Threading): several host threads, one context (= one HIP
    stream) and one run state each, the SAME read-only weights; every thread must produce the
    tokens a lone run produces"""
    import threading
    import rama_amd
    cfg, w, g = load_case("synth_d288_h6")
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), int(g["seed"]), rope=(g["freq_cis_real"], g["freq_cis_imag"]))
    prompts = [[5, 9, 2], [17], [], [3, 3, 3, 3, 40, 41], [100, 7]]
    steps = 24
    solo = rama_amd.Engine(dev, model)
    want = [solo.generate_greedy(p, steps) for p in prompts]
    solo.free()
    ctxs = [rama_amd.Hip(0) for _ in prompts]
    engines = [rama_amd.Engine(c, model) for c in ctxs]
    for i, e in enumerate(engines):
        e.set_graph_mode(i % 2 == 0)          # graph replay and eager launches side by side
    got, errs = [None] * len(prompts), []
    start = threading.Barrier(len(prompts))

    def run(i):
        try:
            start.wait()
            for _ in range(3):
                got[i] = engines[i].generate_greedy(prompts[i], steps)
        except Exception as e:        # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=run, args=(i,)) for i in range(len(prompts))]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    assert got == want
    for e in engines: e.free()
    model.free()


# ------------------------------------------------------------------ randomized shapes

def _random_shapes(n, seed=2024):
    """(dim, hidden, n_layers, n_heads, vocab, seq_len, shared): head sizes 4..256 (multiples of 4),
    row widths that are NOT multiples of the 256-float chunk or of the 4-row workgroup, odd vocab"""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        hs = int(rng.choice([4, 8, 12, 20, 32, 48, 64, 80, 96, 128, 160, 256]))
        n_heads = int(rng.integers(1, 9))
        dim = hs * n_heads
        if dim > 1536:
            continue
        hidden = 4 * int(rng.integers(1, 3 * dim // 4 + 2))
        out.append((dim, hidden, int(rng.integers(1, 4)), n_heads, int(rng.integers(5, 700)),
                    int(rng.integers(12, 80)), bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("shape", _random_shapes(14), ids=lambda s: "d%d_h%d_L%d_H%d_V%d_S%d_%s" % (s[:6] + ("tied" if s[6] else "untied",)))
def test_random_shapes_fused_prefill_and_ops_vs_oracle(dev, shape):
    """shapes nobody tuned for: the fused path, the batched prefill and the 1:1 trait-op path
    must all give the oracle's logits and caches"""
    import rama_amd
    dim, hidden, L, H, V, seq, shared = shape
    cfg = O.Config(dim, hidden, L, H, H, V, seq, shared)
    w = S.synth_weights(cfg, seed=dim + hidden)
    rng = np.random.default_rng(dim * 31 + V)
    toks = [1 % V] + [int(t) for t in rng.integers(0, V, 10)]
    orc = O.Oracle(cfg, w)
    want = [orc.forward(t, p).copy() for p, t in enumerate(toks)]
    # fused, token by token
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    for p, t in enumerate(toks):
        rama_amd.forward_fused(rcfg, wv, rsv, t, p, dev)
        assert np.abs(dev.download(rsv.logits) - want[p]).max() <= LOGIT_ATOL, ("fused", p)
    for buf in ("key_cache", "value_cache"):
        assert np.abs(dev.download(getattr(rsv, buf)) - orc.s[buf]).max() <= STATE_ATOL, buf
    rs.free()
    # batched prefill of the same positions into a fresh state
    rs = rama_amd.RunState.from_config(rcfg, dev); rsv = rama_amd.RunStateView.from_rs(rs)
    _prefill(dev, rcfg, wv, rsv, toks, 0)
    assert np.abs(dev.download(rsv.logits) - want[-1]).max() <= LOGIT_ATOL, "prefill"
    for buf in ("key_cache", "value_cache"):
        assert np.abs(dev.download(getattr(rsv, buf)) - orc.s[buf]).max() <= STATE_ATOL, ("prefill", buf)
    rs.free()
    # the reference's own op sequence through the Device trait mirror (first 3 positions)
    rs = rama_amd.RunState.from_config(rcfg, dev); rsv = rama_amd.RunStateView.from_rs(rs)
    for p, t in enumerate(toks[:3]):
        rama_amd.forward(rcfg, wv, rsv, t, p, dev)
        assert np.abs(dev.download(rsv.logits) - want[p]).max() <= LOGIT_ATOL, ("ops", p)
    rs.free(); ws.free()


# ------------------------------------------------------------------ batched independent sequences

@pytest.mark.parametrize("name,n_seq", [("synth_d64_h4", 3), ("synth_d288_h6", 8), ("synth_d768_h12", 5), ("ckpt_untied", 2), ("synth_7bshape_l1", 8)])
def test_decode_batch_equals_independent_forwards(dev, name, n_seq):
    """rama_decode_batch: n sequences with different histories and positions advance together, one
    weight pass; every sequence must end up with the logits and cache rows of its own forward()"""
    import rama_amd
    cfg, w, g = load_case(name)
    toks = g["tokens"].tolist()
    if name.startswith("ckpt"):
        m = rama_amd.Model.load(dev, GOLDEN / f"{name}.bin")
    else:
        m = rama_amd.Model.synth(dev, to_rama_cfg(cfg), int(g["seed"]), rope=(g["freq_cis_real"], g["freq_cis_imag"]))
    batch = [rama_amd.Engine(dev, m) for _ in range(n_seq)]
    orcs = [O.Oracle(cfg, w) for _ in range(n_seq)]
    # sequence i starts with i tokens of its own history (different positions inside one batch)
    rng = np.random.default_rng(5)
    hist = [[1] + [int(t) for t in rng.integers(0, cfg.vocab_size, i % 4)] for i in range(n_seq)]
    for i in range(n_seq):
        for p_, t in enumerate(hist[i][:-1]):
            batch[i].forward(t, p_); orcs[i].forward(t, p_)
    cur = [h[-1] for h in hist]
    pos = [len(h) - 1 for h in hist]
    steps = min(4, cfg.seq_len - max(pos) - 1)
    for _ in range(steps):
        rama_amd.decode_batch(batch, cur, pos)
        for i in range(n_seq):
            lo = orcs[i].forward(cur[i], pos[i])
            assert np.abs(batch[i].logits() - lo).max() <= LOGIT_ATOL, (i, pos[i])
            cur[i] = O.argmax(lo); pos[i] += 1
    for i in range(n_seq):
        for buf in ("key_cache", "value_cache"):
            assert np.abs(batch[i].buffer(buf, orcs[i].s[buf].size) - orcs[i].s[buf]).max() <= STATE_ATOL, (i, buf)
    for e in batch: e.free()
    m.free()


@pytest.mark.parametrize("n_seq", [17, 40, 64, 65, 100, 128])
def test_decode_batch_many_sequences(dev, n_seq):
    """more sequences than one 16-token MFMA tile: up to 128 independent sequences per weight pass (beyond 64: the
    8-token-tile kernels on the model's tile-order copies)"""
    import rama_amd
    cfg = O.Config(128, 352, 2, 4, 4, 256, 24, True)
    rope = S.rope_tables(cfg.seq_len, cfg.head_size)
    w = S.synth_weights(cfg, seed=3, rope=rope)
    m = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 3, rope=rope)
    batch = [rama_amd.Engine(dev, m) for _ in range(n_seq)]
    orcs = [O.Oracle(cfg, w) for _ in range(n_seq)]
    rng = np.random.default_rng(n_seq)
    cur = [int(t) for t in rng.integers(0, cfg.vocab_size, n_seq)]
    pos = [0] * n_seq
    # stagger the sequences: sequence i is advanced alone i % 3 times first
    for i in range(n_seq):
        for _ in range(i % 3):
            lo = orcs[i].forward(cur[i], pos[i]); batch[i].forward(cur[i], pos[i])
            cur[i] = O.argmax(lo); pos[i] += 1
    for _ in range(3):
        rama_amd.decode_batch(batch, cur, pos)
        for i in range(n_seq):
            lo = orcs[i].forward(cur[i], pos[i])
            assert np.abs(batch[i].logits() - lo).max() <= LOGIT_ATOL, (i, pos[i])
            cur[i] = O.argmax(lo); pos[i] += 1
    for i in (0, n_seq // 2, n_seq - 1):
        for buf in ("key_cache", "value_cache"):
            assert np.abs(batch[i].buffer(buf, orcs[i].s[buf].size) - orcs[i].s[buf]).max() <= STATE_ATOL, (i, buf)
    for e in batch: e.free()
    m.free()


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("n_seq", [1, 5, 19, 64, 97])
def test_decode_batch_chained_equals_independent_generations(dev, n_seq, graph):
    """rama_decode_batch_begin / _steps / _tokens: the cursors of every sequence live on the device, a step ends with
    one argmax per sequence and (graph mode) is one hipGraph replay.  Every sequence must produce the greedy tokens
    its own oracle generation produces, from its own start position, and end with that generation's caches."""
    import rama_amd
    cfg = O.Config(128, 352, 2, 4, 4, 256, 40, True)
    rope = S.rope_tables(cfg.seq_len, cfg.head_size)
    w = S.synth_weights(cfg, seed=4, rope=rope)
    m = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 4, rope=rope)
    batch = [rama_amd.Engine(dev, m) for _ in range(n_seq)]
    orcs = [O.Oracle(cfg, w) for _ in range(n_seq)]
    rng = np.random.default_rng(100 + n_seq)
    cur = [int(t) for t in rng.integers(0, cfg.vocab_size, n_seq)]
    pos = [0] * n_seq
    for i in range(n_seq):                        # stagger: sequence i is advanced alone i % 4 times first
        for _ in range(i % 4):
            lo = orcs[i].forward(cur[i], pos[i]); batch[i].forward(cur[i], pos[i])
            cur[i] = O.argmax(lo); pos[i] += 1
    steps = 9
    batch[0].set_graph_mode(graph)
    try:
        got = rama_amd.decode_batch_chained(batch, cur, pos, steps)
    finally:
        batch[0].set_graph_mode(False)
    for i in range(n_seq):
        want, t, p_ = [], cur[i], pos[i]
        for _ in range(steps):
            t = O.argmax(orcs[i].forward(t, p_)); p_ += 1
            want.append(t)
        assert got[i] == want, (i, got[i], want)
    for i in (0, n_seq // 2, n_seq - 1):
        for buf in ("key_cache", "value_cache"):
            assert np.abs(batch[i].buffer(buf, orcs[i].s[buf].size) - orcs[i].s[buf]).max() <= STATE_ATOL, (i, buf)
    for e in batch: e.free()
    m.free()


def test_decode_batch_chained_across_score_buffer_buckets(dev):
    """the attention score buffers are sized per bucket of 256 timesteps: a chained run that crosses a bucket boundary
    re-captures its graph and keeps producing the single-sequence tokens"""
    import rama_amd
    cfg = O.Config(64, 176, 1, 4, 4, 96, 300, False)
    rope = S.rope_tables(cfg.seq_len, cfg.head_size)
    w = S.synth_weights(cfg, seed=8, rope=rope)
    m = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 8, rope=rope)
    a, b = rama_amd.Engine(dev, m), rama_amd.Engine(dev, m)
    a.set_graph_mode(True)
    try:
        got = rama_amd.decode_batch_chained([a, b], [3, 7], [0, 0], 270)
    finally:
        a.set_graph_mode(False)
    for i, t0 in enumerate((3, 7)):
        e = rama_amd.Engine(dev, m)
        assert got[i] == e.generate_greedy([], 271)[1:] if False else True     # (generate() starts from BOS; compared below instead)
        orc = O.Oracle(cfg, w)
        t, want = t0, []
        for p_ in range(270):
            t = O.argmax(orc.forward(t, p_)); want.append(t)
        assert got[i] == want, (i, next(k for k in range(270) if got[i][k] != want[k]))
        e.free()
    a.free(); b.free(); m.free()


def test_decode_batch_argument_errors(dev):
    import rama_amd
    cfg, w, g = load_case("synth_d64_h4")
    m = rama_amd.Model.synth(dev, to_rama_cfg(cfg), int(g["seed"]), rope=(g["freq_cis_real"], g["freq_cis_imag"]))
    a, b = rama_amd.Engine(dev, m), rama_amd.Engine(dev, m)
    with pytest.raises(rama_amd.RamaError):
        rama_amd.decode_batch([a, a], [1, 1], [0, 0])                 # one state twice
    with pytest.raises(rama_amd.RamaError):
        rama_amd.decode_batch([a, b], [1, cfg.vocab_size], [0, 0])    # token outside the vocabulary
    with pytest.raises(rama_amd.RamaError):
        rama_amd.decode_batch([a, b], [1, 1], [0, cfg.seq_len])       # position outside the context
    a.free(); b.free(); m.free()


def test_decode_batch_beyond_64_needs_the_tile_order_copies(dev):
    """65+ sequences per pass run the 8-token-tile kernels, which read the model's tile-order weight copies: with the
    copies switched off ("tiled" = 0) the call is refused, not silently split"""
    import rama_amd
    from rama_amd._lib import check
    cfg = O.Config(64, 176, 1, 4, 4, 96, 8, True)
    rope = S.rope_tables(cfg.seq_len, cfg.head_size)
    m = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 2, rope=rope)
    batch = [rama_amd.Engine(dev, m) for _ in range(65)]
    check(dev.lib.rama_set_tuning(dev.ctx, b"tiled", 0))
    try:
        with pytest.raises(rama_amd.RamaError):
            rama_amd.decode_batch(batch, [1] * 65, [0] * 65)
        rama_amd.decode_batch(batch[:64], [1] * 64, [0] * 64)
    finally:
        check(dev.lib.rama_set_tuning(dev.ctx, b"tiled", 1))
    rama_amd.decode_batch(batch, [1] * 65, [0] * 65)
    for e in batch: e.free()
    m.free()


def test_interleaved_w13_copy_equals_two_tensor_path(dev, tmp_path):
    """a rama_model keeps W1 | W3 row-interleaved for the fused SwiGLU launch (model.hip); logits with
    the copy in use ("w13i" = 1, default) and with the checkpoint's two tensors ("w13i" = 0) are the
    same to the last bit -- the same four rows per workgroup, the same order of operations -- and the
    saved file still holds the checkpoint layout"""
    import rama_amd
    name = "synth_d288_h6"
    cfg, w, g = load_case(name)
    m = rama_amd.Model.synth(dev, to_rama_cfg(cfg), int(g["seed"]), rope=(g["freq_cis_real"], g["freq_cis_imag"]))
    eng = rama_amd.Engine(dev, m)
    toks = g["tokens"].tolist()[:6]
    runs = {}
    for v in (1, 0):
        eng.set_tuning("w13i", v)
        e2 = rama_amd.Engine(dev, m)
        out = []
        for pos, t in enumerate(toks):
            e2.forward(t, pos)
            out.append(e2.logits())
        runs[v] = out
        e2.free()
    eng.set_tuning("w13i", 1)
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    orc = O.Oracle(cfg, w)
    for pos, t in enumerate(toks):
        assert np.abs(runs[1][pos] - orc.forward(t, pos)).max() <= LOGIT_ATOL
    p = tmp_path / "saved.bin"
    m.save(p)
    cfg2, w2 = O.read_checkpoint(p)
    assert cfg2 == cfg
    for k in ("w1", "w3", "w2"):
        assert np.array_equal(w2[k].reshape(-1), w[k].reshape(-1)), k
    eng.free(); m.free()


def test_v1_ak42_checkpoint_is_rejected(dev):
    """a llama2.c v1 file written by the reference's own version1_export (export.py:132-180, magic
    "ak42" + 256-byte header): the engine reads v0 only (mod.rs:141-166), the loader must say so
    instead of taking the magic for `dim`"""
    import rama_amd
    with pytest.raises(rama_amd.RamaError) as e:
        rama_amd.Model.load(dev, GOLDEN / "ckpt_v1_ak42.bin")
    assert "(-2)" in str(e.value)           # RAMA_EUNSUP
    with pytest.raises(Exception):
        O.read_checkpoint(GOLDEN / "ckpt_v1_ak42.bin")


# ------------------------------------------------------------------ generate_stream (mod.rs:209-248): tokens as they are produced

@pytest.mark.parametrize("temperature,prompt", [(0.0, []), (0.0, [5, 9, 3, 7, 11]), (1.0, [5, 9, 3]), (0.0, list(range(2, 40)))])
def test_generate_stream_hands_over_every_token_in_order(dev, temperature, prompt):
    """rama_generate_stream: the chained loop writes every token to a host-visible ring as well; the callback sees
    index 0, 1, .. with exactly the tokens rama_generate returns (and the oracle's at temperature 0), prompt positions
    included, whether the prompt went through the per-token loop or through rama_prefill"""
    import rama_amd
    cfg = O.Config(128, 352, 2, 4, 4, 256, 96, True)
    rope = S.rope_tables(cfg.seq_len, cfg.head_size)
    w = S.synth_weights(cfg, seed=21, rope=rope)
    m = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 21, rope=rope)
    steps = 80
    u = 0.2721174359321594
    a, b = rama_amd.Engine(dev, m), rama_amd.Engine(dev, m)
    want = a.generate(prompt, steps, temperature, 0.9, u)
    seen = []
    for graph in (False, True):
        b.set_graph_mode(graph)
        seen.clear()
        got = b.generate_stream(prompt, steps, lambda i, t: seen.append((i, t)), temperature, 0.9, u)
        assert got == want
        assert [i for i, _ in seen] == list(range(steps))
        assert [t for _, t in seen] == want
    b.set_graph_mode(False)
    if temperature == 0.0:
        assert want == O.Oracle(cfg, w).generate_greedy(prompt, steps)
    a.free(); b.free(); m.free()


def test_decode_batch_stream_poll_per_sequence(dev):
    """the chained batch writes each sequence's tokens to a host-visible ring of its own: polled while the steps run,
    every sequence's stream is, in order, what rama_decode_batch_tokens returns at the end"""
    import rama_amd
    cfg = O.Config(128, 352, 2, 4, 4, 256, 64, True)
    rope = S.rope_tables(cfg.seq_len, cfg.head_size)
    m = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 4, rope=rope)
    n_seq, steps = 5, 40
    batch = [rama_amd.Engine(dev, m) for _ in range(n_seq)]
    for rnd in range(2):                          # the second round re-uses (and must have cleared) the rings
        seen = [[] for _ in range(n_seq)]
        got = rama_amd.decode_batch_chained(batch, [3 + rnd + i for i in range(n_seq)], [0] * n_seq, steps,
                                            on_token=lambda s_, i, t: seen[s_].append((i, t)))
        for s_ in range(n_seq):
            assert [i for i, _ in seen[s_]] == list(range(steps))
            assert [t for _, t in seen[s_]] == got[s_]
    for e in batch: e.free()
    m.free()


def test_decode_stream_poll_sees_the_loop_progress(dev):
    """rama_decode_stream_poll never blocks: polled while rama_decode_steps' launches are still running it returns the
    tokens produced so far (possibly none), and in the end exactly rama_decode_tokens' list; a new rama_decode_begin
    clears the ring"""
    import rama_amd
    cfg = O.Config(288, 768, 4, 6, 6, 512, 256, True)
    rope = S.rope_tables(cfg.seq_len, cfg.head_size)
    m = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 5, rope=rope)
    e = rama_amd.Engine(dev, m)
    for rnd in range(2):
        e.decode_begin(1 + rnd, 0, [7, 8, 9])
        assert e.decode_stream_poll(0) == []                     # nothing produced yet (and the previous round's tokens are gone)
        e.decode_steps(200)
        got, polls = [], 0
        while len(got) < 200:
            got += e.decode_stream_poll(len(got), 64)
            polls += 1
            assert polls < 10_000_000
        assert got == e.decode_tokens()
        assert got[:3] == [7, 8, 9]
    e.free(); m.free()


@pytest.mark.parametrize("steps,prompt", [(0, [4, 5]), (1, []), (2, [4, 5, 6]), (3, [4, 5, 6]), (4, [4, 5, 6]), (5, [4, 5, 6])])
def test_generate_stream_edge_lengths(dev, steps, prompt):
    """generations shorter than, equal to and just beyond the prompt: the stream hands over exactly rama_generate's list"""
    import rama_amd
    cfg = O.Config(64, 176, 2, 4, 4, 96, 16, True)
    rope = S.rope_tables(cfg.seq_len, cfg.head_size)
    m = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 3, rope=rope)
    a, b = rama_amd.Engine(dev, m), rama_amd.Engine(dev, m)
    want = a.generate(prompt, steps) if steps else []
    seen = []
    got = b.generate_stream(prompt, steps, lambda i, t: seen.append((i, t)))
    assert got == want and [t for _, t in seen] == want and [i for i, _ in seen] == list(range(steps))
    a.free(); b.free(); m.free()
