"""-m gpu: the reference-order mode (rama_set_tuning "ref_order" = 1, csrc/ref_order.hpp) must
reproduce the CPU oracle BIT FOR BIT -- every Device<T> op, every RunState buffer of forward()
(engine/src/transformer/infer.rs:8-53 over engine/src/device/cpu.rs), the generate() loop.

This is the parity mode: the fast path sums dot products in a different order than the reference
(a few 1e-5 .. 1.5e-4 apart at llama2-7B depth, all of it the reference's own rounding error, see
tests/test_hip_parity_7b.py); this mode removes the order difference, so what remains is zero."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from oracle import synth as S

from .helpers import CKPT_CASES, SYNTH_CASES, gpu_views, load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import rama_amd
    from rama_amd._lib import check
    d = rama_amd.Hip(0)
    check(d.lib.rama_set_tuning(d.ctx, b"ref_order", 1))
    yield d
    check(d.lib.rama_set_tuning(d.ctx, b"ref_order", 0))
    d.close()


def up(dev, a):
    import rama_amd
    return rama_amd.MutView(dev.allocate(a))


def rnd(n, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(n) * scale).astype(np.float32)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bits_equal(got, want, what=""):
    g, w = bits(got), bits(want)
    if not np.array_equal(g, w):
        bad = np.flatnonzero(g != w)
        raise AssertionError(f"{what}: {bad.size} of {g.size} values differ, first at {bad[0]}: "
                             f"{got.reshape(-1)[bad[0]]!r} vs {want.reshape(-1)[bad[0]]!r}")


# ------------------------------------------------------------------ expf

def test_expf_matches_libm_bit_for_bit(dev):
    """glibc's expf restated in double on the device (ref_order.hpp) against the host libm the
    oracle -- and Rust's f32::exp -- call: every input class the decode path produces (softmax
    arguments <= 0 down to underflow, SiLU arguments of either sign) plus the edges"""
    from rama_amd._lib import check
    rng = np.random.default_rng(0)
    parts = [
        rng.uniform(-110.0, 0.0, 2_000_000), rng.uniform(-20.0, 20.0, 2_000_000), rng.uniform(-1e-3, 1e-3, 200_000),
        rng.uniform(80.0, 95.0, 100_000), rng.uniform(-105.0, -85.0, 200_000),
        np.array([0.0, -0.0, 1.0, -1.0, 88.72283, 88.72284, 88.7229, -87.33654, -87.33655, -103.97207, -103.97208, -103.9721,
                  np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e38, -1e38, 88.0, -88.0, 87.99999, -87.99999]),
        np.linspace(-104.0, 89.0, 1_000_003),
    ]
    x = np.concatenate(parts).astype(np.float32)
    want = O.expf(x)
    tx = up(dev, x); to = up(dev, np.zeros_like(x))
    check(dev.lib.rama_ref_expf(dev.ctx, to.ptr, tx.ptr, x.size))
    got = dev.download(to)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert_bits_equal(got[~nan], want[~nan], "expf")


# ------------------------------------------------------------------ ops

@pytest.mark.parametrize("rows,width", [(1, 4), (7, 8), (288, 288), (768, 288), (333, 772), (4096, 4096), (64, 11008), (1000, 4100)])
def test_matmul_bit_exact(dev, rows, width):
    w, x = rnd(rows * width, rows + width, 0.05), rnd(width, 5, 1.5)
    want = np.empty(rows, np.float32)
    O.matmul(want, w, x, width, rows)
    tw = up(dev, w); tx = up(dev, x); to = up(dev, np.zeros(rows, np.float32))
    dev.matmul(to, tw.as_view(), tx.as_view(), width, rows, 1)
    assert_bits_equal(dev.download(to), want, "matmul")


@pytest.mark.parametrize("order", ["strided", "sequential"])
def test_lane_reduce_orders_bit_exact(dev, order):
    """[r6] cpu.rs:148 `v.reduce_add()`: the order of the final 4-lane sum belongs to the `wide` crate and the build's target features, not to the
    reference.  "lane_reduce" = 1 (strided, (l0+l2)+(l1+l3)) | 2 (sequential, ((l0+l1)+l2)+l3) against the oracle switched the same way
    (oracle_set_lane_reduce): Device::matmul on a model-less matrix (the chain-order view, one and two waves per row group, half groups, the
    one-thread-per-row kernel of an unaligned view, o_cols = 2), forward() through the fused entry and through the 1:1 ops with every RunState
    buffer, a prompt through the chain-order token-batch kernels -- bit for bit; and the orders do differ from the default on the same inputs"""
    import rama_amd
    from rama_amd._lib import check
    from .helpers import to_rama_cfg
    code = O.LANES[order]
    check(dev.lib.rama_set_tuning(dev.ctx, b"lane_reduce", code))
    try:
        differs = 0
        with O.orders(lane_reduce=order):
            for rows, width in [(7, 8), (288, 288), (333, 772), (4096, 4096), (64, 11008), (22016, 4096), (1000, 4100)]:
                w, x = rnd(rows * width, rows + width, 0.05), rnd(width, 5, 1.5)
                want = np.empty(rows, np.float32)
                O.matmul(want, w, x, width, rows)
                tw = up(dev, w); tx = up(dev, x); to = up(dev, np.zeros(rows, np.float32))
                dev.matmul(to, tw.as_view(), tx.as_view(), width, rows, 1)
                assert_bits_equal(dev.download(to), want, f"matmul {rows}x{width} lanes {order}")
                with O.orders(lane_reduce="pairwise"):
                    base = np.empty(rows, np.float32); O.matmul(base, w, x, width, rows)
                differs += int((bits(base) != bits(want)).sum())
            assert differs > 0, "the lane orders never differed from the default: nothing was tested"
            # an unaligned view (the one-thread-per-row kernel) and the trait's o_cols = 2 shape
            rows, width = 50, 64
            w, x = rnd(rows * width + 3, 1, 0.1), rnd(width + 1, 2)
            want = np.empty(rows, np.float32)
            O.matmul(want, np.ascontiguousarray(w[3:]), np.ascontiguousarray(x[1:]), width, rows)
            sw = dev.allocate(w); sx = dev.allocate(x); to = up(dev, np.zeros(rows, np.float32))
            dev.matmul(to, rama_amd.View(sw).slice(3), rama_amd.View(sx).slice(1), width, rows, 1)
            assert_bits_equal(dev.download(to), want, f"matmul (unaligned view) lanes {order}")
            b2 = rnd(width * 2, 9)
            want2 = np.empty(rows * 2, np.float32)
            O.matmul(want2, np.ascontiguousarray(w[3:]), b2, width, rows, 2)
            tb2 = up(dev, b2); to2 = up(dev, np.zeros(rows * 2, np.float32))
            dev.matmul(to2, rama_amd.View(sw).slice(3), tb2.as_view(), width, rows, 2)
            assert_bits_equal(dev.download(to2), want2, f"matmul o_cols=2 lanes {order}")
            # forward(): fused entry + 1:1 ops on a small golden case, and llama2-7B's width (leader norms, two waves per row group, W1|W3's half groups)
            cfg, wts, g = load_case("synth_d288_h6")
            toks = g["tokens"].tolist()[:6]
            orc = O.Oracle(cfg, wts)
            rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, wts)
            model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), int(g["seed"]), rope=(g["freq_cis_real"], g["freq_cis_imag"]))
            eng = rama_amd.Engine(dev, model)
            for pos, t in enumerate(toks):
                lo = orc.forward(t, pos)
                eng.forward(t, pos)
                rama_amd.forward(rcfg, wv, rsv, t, pos, dev)
                assert_bits_equal(eng.logits(), lo, f"fused entry lanes {order} pos {pos}")
                for buf in ("logits", "x", "xb", "xb2", "hb", "q", "key_cache"):
                    assert_bits_equal(dev.download(getattr(rsv, buf)), orc.s[buf], f"1:1 ops lanes {order} pos {pos} {buf}")
            eng.free(); model.free(); rs.free(); ws.free()
            d, h, L, H, V, seq = 4096, 11008, 1, 32, 320, 64
            cfg7 = O.Config(d, h, L, H, H, V, seq, False)
            rope = S.rope_tables(seq, d // H)
            w7 = S.synth_weights(cfg7, 17, rope=rope)
            orc7 = O.Oracle(cfg7, w7)
            model7 = rama_amd.Model.synth(dev, to_rama_cfg(cfg7), 17, rope=rope)
            eng7 = rama_amd.Engine(dev, model7)
            token = 1
            for pos in range(4):
                lo = orc7.forward(token, pos)
                eng7.forward(token, pos)
                assert_bits_equal(eng7.logits(), lo, f"7B width lanes {order} pos {pos}")
                for buf, n in (("x", d), ("hb", h), ("q", d)):
                    assert_bits_equal(eng7.buffer(buf, n), orc7.s[buf], f"7B width lanes {order} pos {pos} {buf}")
                token = O.argmax(lo)
            # a prompt through the chain-order token-batch kernels (gemm_chain_kernel)
            toks = [1] + [int(t) for t in np.random.default_rng(3).integers(0, V, 20)]
            orc8 = O.Oracle(cfg7, w7)
            for pos, t in enumerate(toks):
                lo = orc8.forward(t, pos)
            eng8 = rama_amd.Engine(dev, model7)
            arr = (C.c_int32 * len(toks))(*toks)
            check(dev.lib.rama_prefill(dev.ctx, C.byref(model7.ccfg), C.byref(model7.weights), C.byref(eng8.state), arr, len(toks), 0), "rama_prefill")
            assert_bits_equal(eng8.logits(), lo, f"prefill lanes {order}: logits")
            assert_bits_equal(eng8.buffer("key_cache", L * seq * d), orc8.s["key_cache"], f"prefill lanes {order}: key_cache")
            eng7.free(); eng8.free(); model7.free()
    finally:
        check(dev.lib.rama_set_tuning(dev.ctx, b"lane_reduce", 0))
    with pytest.raises(rama_amd.RamaError):
        check(dev.lib.rama_set_tuning(dev.ctx, b"lane_reduce", 3))


def test_rmsnorm_then_matmul_with_columns_reads_the_norm(dev):
    """[r6] (round 5's advisor) a parity-mode Device::rmsnorm is RECORDED (it may ride in the launch of the matmuls on its output); a Device::matmul with
    o_cols = 2 reading that output records nothing and used to run before the norm had been issued.  Now everything pending is issued first, and
    the o_cols > 1 product follows the reference's order too (cpu.rs:137-151)"""
    n, rows = 256, 24
    x, g = rnd(n, 1, 2.0), rnd(n, 2)
    a = rnd(rows * (n // 2), 3, 0.1)
    normed = np.empty(n, np.float32)
    O.rmsnorm(normed, x, g, n)
    want = np.empty(rows * 2, np.float32)
    O.matmul(want, a, normed, n // 2, rows, 2)          # b = the normalised vector read as [n / 2, 2]
    tx = up(dev, x); tg = up(dev, g); tn = up(dev, np.full(n, 7.0, np.float32)); ta = up(dev, a); to = up(dev, np.zeros(rows * 2, np.float32))
    dev.rmsnorm(tn, tx.as_view(), tg.as_view(), n)
    dev.matmul(to, ta.as_view(), tn.as_view(), n // 2, rows, 2)
    assert_bits_equal(dev.download(to), want, "matmul (o_cols = 2) behind a recorded rmsnorm")
    assert_bits_equal(dev.download(tn), normed, "the recorded rmsnorm's output")


def test_device_side_writes_dissolve_derived_copies(dev):
    """[r6] (round 5's advisor) chain-order copies derived from tensors the caller uploaded are dropped by rama_free / rama_copy_h2d_f32 -- and now by every
    entry that WRITES device memory: TransformerWeights re-seeded in place with rama_fill_synth, an op's output landing in a matrix.  The next
    parity-mode call derives them again from the live tensor: logits bit for bit the oracle's for the NEW weights"""
    import rama_amd
    from rama_amd._lib import check
    from .helpers import to_rama_cfg
    cfg = O.Config(128, 352, 2, 2, 2, 96, 32, False)
    rcfg = to_rama_cfg(cfg)
    ws = rama_amd.TransformerWeights.synth(rcfg, 31, dev)
    wv = rama_amd.TransformerWeightsView.from_gpu_ws(ws)
    rs = rama_amd.RunState.from_config(rcfg, dev); rsv = rama_amd.RunStateView.from_rs(rs)
    fr = dev.download(wv.freq_cis_real).reshape(cfg.seq_len, -1); fi = dev.download(wv.freq_cis_imag).reshape(cfg.seq_len, -1)

    def check_seed(seed, what):
        w = S.synth_weights(cfg, seed, rope=(fr, fi))
        orc = O.Oracle(cfg, w)
        token = 1
        for pos in range(3):
            lo = orc.forward(token, pos)
            rama_amd.forward_fused(rcfg, wv, rsv, token, pos, dev)
            assert_bits_equal(dev.download(rsv.logits), lo, f"{what}: fused entry pos {pos}")
            rama_amd.forward(rcfg, wv, rsv, token, pos, dev)
            assert_bits_equal(dev.download(rsv.logits), lo, f"{what}: 1:1 ops pos {pos}")
            token = O.argmax(lo)

    check_seed(31, "first upload")
    assert _chain_lookup(dev, wv.wq.ptr, cfg.dim, cfg.dim), "no chain-order copy was derived: nothing is tested"
    # the same tensors re-seeded ON THE DEVICE (what TransformerWeights.synth does per tensor): the copies must go
    spec = S.synth_spec(cfg)
    for name in ("wq", "wk", "wv", "wo", "w1", "w2", "w3", "token_embedding_table", "wcls", "rms_att_weight", "rms_ffn_weight", "rms_final_weight"):
        view = getattr(wv, name)
        tag, scale, bias = spec[name]
        n = int(np.prod(dict(O.weight_shapes(cfg))[name]))
        check(dev.lib.rama_fill_synth(dev.ctx, view.ptr, n, 32, tag, 0, float(scale), float(bias)), "rama_fill_synth")
    assert not _chain_lookup(dev, wv.wq.ptr, cfg.dim, cfg.dim), "a chain-order copy survived a device-side write into its tensor"
    check_seed(32, "re-seeded in place")
    # an op's output landing in a matrix (array_add of the matrix onto itself: every element doubles)
    assert _chain_lookup(dev, wv.wo.ptr, cfg.dim, cfg.dim)
    dev.array_add(rama_amd.MutView(ws.wo), wv.wo, cfg.n_layers * cfg.dim * cfg.dim)
    assert not _chain_lookup(dev, wv.wo.ptr, cfg.dim, cfg.dim), "a chain-order copy survived an op that wrote its tensor"
    w = S.synth_weights(cfg, 32, rope=(fr, fi))
    w["wo"] = (w["wo"] + w["wo"]).astype(np.float32)
    orc = O.Oracle(cfg, w)
    lo = orc.forward(5, 0)
    rama_amd.forward_fused(rcfg, wv, rsv, 5, 0, dev)
    assert_bits_equal(dev.download(rsv.logits), lo, "after an op wrote wo")
    rs.free(); ws.free()


def test_matmul_unaligned_view_bit_exact(dev):
    import rama_amd
    rows, width = 50, 64
    w, x = rnd(rows * width + 3, 1, 0.1), rnd(width + 1, 2)
    want = np.empty(rows, np.float32)
    O.matmul(want, np.ascontiguousarray(w[3:]), np.ascontiguousarray(x[1:]), width, rows)
    sw = dev.allocate(w); sx = dev.allocate(x); to = up(dev, np.zeros(rows, np.float32))
    dev.matmul(to, rama_amd.View(sw).slice(3), rama_amd.View(sx).slice(1), width, rows, 1)
    assert_bits_equal(dev.download(to), want, "matmul (unaligned view)")


@pytest.mark.parametrize("n", [4, 288, 768, 4096])
def test_rmsnorm_sinu_softmax_bit_exact(dev, n):
    x, w = rnd(n, n, 2.0), rnd(n, n + 1)
    want = np.empty(n, np.float32)
    O.rmsnorm(want, x, w, n)
    tx = up(dev, x); tw = up(dev, w); to = up(dev, np.zeros(n, np.float32))
    dev.rmsnorm(to, tx.as_view(), tw.as_view(), n)
    assert_bits_equal(dev.download(to), want, "rmsnorm")
    a = rnd(n, n + 2, 4.0)
    ta = up(dev, a)
    dev.sinu(ta, n)
    O.sinu(a, n)
    assert_bits_equal(dev.download(ta), a, "sinu")
    s = rnd(n, n + 3, 6.0)
    ts = up(dev, s)
    dev.softmax(ts, n)
    O.softmax(s, n)
    assert_bits_equal(dev.download(ts), s, "softmax")


@pytest.mark.parametrize("hs", [4, 48, 64, 128])
def test_apply_position_bit_exact(dev, hs):
    q, k = rnd(hs, 1), rnd(hs, 2)
    pr, pi = rnd(hs // 2, 3), rnd(hs // 2, 4)
    tq = up(dev, q); tk = up(dev, k); tr = up(dev, pr); ti = up(dev, pi)
    dev.apply_position(tq, tk, tr.as_view(), ti.as_view(), hs)
    O.apply_position(q, k, pr, pi, hs)
    assert_bits_equal(dev.download(tq), q, "rope q")
    assert_bits_equal(dev.download(tk), k, "rope k")


@pytest.mark.parametrize("n_heads,hs,seq_len,positions", [(4, 16, 32, [0, 1, 31]), (6, 48, 256, [0, 17, 255]), (2, 128, 300, [0, 64, 299])])
def test_multi_head_attention_bit_exact(dev, n_heads, hs, seq_len, positions):
    import rama_amd
    dim, L = n_heads * hs, 2
    cfg = O.Config(dim, 4 * dim, L, n_heads, n_heads, 8, seq_len, True)
    rcfg = rama_amd.Config(dim, 4 * dim, L, n_heads, n_heads, 8, seq_len, True)
    z1 = np.zeros(1, np.float32)
    orc = O.Oracle(cfg, dict(token_embedding_table=np.zeros((8, dim), np.float32), rms_att_weight=np.zeros((L, dim), np.float32),
                             rms_ffn_weight=np.zeros((L, dim), np.float32), wq=z1, wk=z1, wv=z1, wo=z1, w1=z1, w2=z1, w3=z1,
                             rms_final_weight=np.zeros(dim, np.float32), freq_cis_real=z1, freq_cis_imag=z1))
    kc = rnd(L * seq_len * dim, 20); vc = rnd(L * seq_len * dim, 21); q = rnd(dim, 22, 2.0)
    rs = rama_amd.RunState.from_config(rcfg, dev); rsv = rama_amd.RunStateView.from_rs(rs)
    dev.upload_into(rsv.key_cache, kc); dev.upload_into(rsv.value_cache, vc); dev.upload_into(rsv.q, q)
    for layer in (0, 1):
        for pos in positions:
            orc.s["key_cache"][:] = kc; orc.s["value_cache"][:] = vc; orc.s["q"][:] = q
            orc.multi_head_attention(layer, pos)
            dev.multi_head_attention(rsv, rcfg, layer, pos)
            assert_bits_equal(dev.download(rsv.xb), orc.s["xb"], f"xb layer {layer} pos {pos}")
            att = dev.download(rsv.att).reshape(n_heads, seq_len)[:, :pos + 1]
            assert_bits_equal(att, orc.s["att"].reshape(n_heads, seq_len)[:, :pos + 1], f"att layer {layer} pos {pos}")
    rs.free()


# ------------------------------------------------------------------ forward(): every RunState buffer

STATE_BUFS = ("x", "xb", "xb2", "hb", "hb2", "q", "k", "v", "logits", "key_cache", "value_cache")


@pytest.mark.parametrize("name", CKPT_CASES + SYNTH_CASES)
def test_forward_every_buffer_bit_exact(dev, name):
    """the fused entry (rama_forward) and the 1:1 trait-op composition, both in reference order,
    leave the oracle's RunState bit for bit at every position of the fixture"""
    import rama_amd
    cfg, w, g = load_case(name)
    toks = g["tokens"].tolist()[:10]
    orc = O.Oracle(cfg, w)
    rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
    rs2 = rama_amd.RunState.from_config(rcfg, dev); rsv2 = rama_amd.RunStateView.from_rs(rs2)
    for pos, t in enumerate(toks):
        orc.forward(t, pos)
        rama_amd.forward_fused(rcfg, wv, rsv, t, pos, dev)
        rama_amd.forward(rcfg, wv, rsv2, t, pos, dev)
        for buf in STATE_BUFS:
            assert_bits_equal(dev.download(getattr(rsv, buf)), orc.s[buf], f"{name} fused pos {pos} {buf}")
            assert_bits_equal(dev.download(getattr(rsv2, buf)), orc.s[buf], f"{name} ops pos {pos} {buf}")
        att = dev.download(rsv.att).reshape(cfg.n_heads, cfg.seq_len)[:, :pos + 1]
        assert_bits_equal(att, orc.s["att"].reshape(cfg.n_heads, cfg.seq_len)[:, :pos + 1], f"{name} pos {pos} att")
    rs.free(); rs2.free(); ws.free()


@pytest.mark.parametrize("shape", ["stories15M", "stories110M", "llama2-7B-2layers"])
def test_full_shape_logits_bit_exact(dev, shape):
    """BASELINE shapes at full width and vocabulary: logits of the generate() loop on 'once upon a
    time' identical to the oracle's, bit for bit, device-chained greedy tokens identical"""
    import rama_amd
    from .helpers import to_rama_cfg
    shapes = {"stories15M": (288, 768, 6, 6, 32000, 256, True), "stories110M": (768, 2048, 12, 12, 32000, 1024, True),
              "llama2-7B-2layers": (4096, 11008, 2, 32, 32000, 2048, False)}
    d, h, L, H, V, seq, shared = shapes[shape]
    cfg = O.Config(d, h, L, H, H, V, seq, shared)
    rope = S.rope_tables(seq, d // H)
    w = S.synth_weights(cfg, 0, rope=rope)
    orc = O.Oracle(cfg, w)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 0, rope=rope)
    eng = rama_amd.Engine(dev, model)
    prompt = [10646, 2501, 263, 931]
    steps = 12 if d < 4096 else 6
    token = 1
    for pos in range(steps):
        lo = orc.forward(token, pos)
        eng.forward(token, pos)
        assert_bits_equal(eng.logits(), lo, f"{shape} pos {pos}")
        token = prompt[pos] if pos < len(prompt) else O.argmax(lo)
    eng2 = rama_amd.Engine(dev, model)
    assert eng2.generate_greedy(prompt, steps) == O.Oracle(cfg, w).generate_greedy(prompt, steps)
    eng.free(); eng2.free(); model.free()


# ------------------------------------------------------------------ parity mode on a RESIDENT MODEL: chain-order weights (csrc/chain.hpp)

def _engine_buf(eng, name, n):
    return eng.buffer(name, n)


@pytest.mark.parametrize("name", CKPT_CASES + SYNTH_CASES)
def test_model_forward_every_buffer_bit_exact(dev, name, tmp_path):
    """a resident model (rama_model_load / rama_model_synth) runs parity mode on its chain-order weight
    copy -- one lane per (row, k mod 4) chain, the exact scan for the sequential sums: every RunState
    buffer must still be the oracle's, bit for bit, and identical to the one-thread-per-row kernels'"""
    import rama_amd
    from .helpers import GOLDEN, to_rama_cfg
    cfg, w, g = load_case(name)
    toks = g["tokens"].tolist()[:10]
    orc = O.Oracle(cfg, w)
    if name.startswith("ckpt_"):
        model = rama_amd.Model.load(dev, GOLDEN / f"{name}.bin")
    else:
        model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), int(g["seed"]), rope=(g["freq_cis_real"], g["freq_cis_imag"]))
    eng = rama_amd.Engine(dev, model)
    sizes = dict(x=cfg.dim, xb=cfg.dim, xb2=cfg.dim, hb=cfg.hidden_dim, hb2=cfg.hidden_dim, q=cfg.dim, k=cfg.dim, v=cfg.dim,
                 logits=cfg.vocab_size, key_cache=cfg.n_layers * cfg.seq_len * cfg.dim, value_cache=cfg.n_layers * cfg.seq_len * cfg.dim)
    for pos, t in enumerate(toks):
        orc.forward(t, pos)
        eng.forward(t, pos)
        for buf, n in sizes.items():
            assert_bits_equal(eng.buffer(buf, n), orc.s[buf], f"{name} model pos {pos} {buf}")
        att = eng.buffer("att", cfg.n_heads * cfg.seq_len).reshape(cfg.n_heads, cfg.seq_len)[:, :pos + 1]
        assert_bits_equal(att, orc.s["att"].reshape(cfg.n_heads, cfg.seq_len)[:, :pos + 1], f"{name} model pos {pos} att")
    # the same positions without the chain-order copy (ref_order.hpp's kernels) give the same bits
    eng.set_tuning("chain", 0)
    try:
        eng2 = rama_amd.Engine(dev, model)
        for pos, t in enumerate(toks):
            eng2.forward(t, pos)
        for buf, n in sizes.items():
            assert_bits_equal(eng2.buffer(buf, n), orc.s[buf], f"{name} model (chain off) {buf}")
        eng2.free()
    finally:
        eng.set_tuning("chain", 1)
    eng.free(); model.free()


@pytest.mark.parametrize("dim,hidden,heads,vocab,seq", [(48, 80, 3, 50, 40), (144, 400, 3, 100, 70), (16, 16, 1, 17, 8), (272, 720, 17, 333, 64)])
def test_model_ragged_shapes_bit_exact(dev, dim, hidden, heads, vocab, seq):
    """row counts that are not multiples of 16 (padded chain groups), widths of a few 16-float blocks
    (shorter than the load ring), odd head counts: logits and caches bit for bit over a short generation"""
    import rama_amd
    from .helpers import to_rama_cfg
    cfg = O.Config(dim, hidden, 2, heads, heads, vocab, seq, False)
    rope = S.rope_tables(seq, dim // heads)
    w = S.synth_weights(cfg, 3, rope=rope)
    orc = O.Oracle(cfg, w)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 3, rope=rope)
    eng = rama_amd.Engine(dev, model)
    token = 1
    for pos in range(min(seq, 12)):
        lo = orc.forward(token, pos)
        eng.forward(token, pos)
        assert_bits_equal(eng.logits(), lo, f"ragged pos {pos} logits")
        assert_bits_equal(eng.buffer("key_cache", 2 * seq * dim), orc.s["key_cache"], f"ragged pos {pos} key_cache")
        assert_bits_equal(eng.buffer("hb", hidden), orc.s["hb"], f"ragged pos {pos} hb")
        token = O.argmax(lo)
    eng.free(); model.free()


def test_model_matmul_views_take_the_chain_copy(dev):
    """Device::matmul on a layer-aligned view of a resident model's tensor streams the chain-order copy;
    any other view (row offset, fewer rows) takes the row-major kernel: same bits either way"""
    import rama_amd
    from rama_amd._lib import check
    from .helpers import to_rama_cfg
    cfg = O.Config(64, 176, 3, 4, 4, 96, 16, False)
    rope = S.rope_tables(cfg.seq_len, cfg.head_size)
    w = S.synth_weights(cfg, 5, rope=rope)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 5, rope=rope)
    x = rnd(cfg.dim, 9, 1.5); xh = rnd(cfg.hidden_dim, 10, 1.5)
    tx = up(dev, x); txh = up(dev, xh)
    cases = [("wq", 1, cfg.dim, cfg.dim, 0, x, tx), ("wo", 2, cfg.dim, cfg.dim, 0, x, tx), ("w1", 1, cfg.hidden_dim, cfg.dim, 0, x, tx),
             ("w2", 2, cfg.dim, cfg.hidden_dim, 0, xh, txh), ("wcls", 0, cfg.vocab_size, cfg.dim, 0, x, tx),
             ("wk", 1, cfg.dim - 8, cfg.dim, 8, x, tx)]      # the last one starts 8 rows into the layer: not chain-aligned
    for name, layer, rows, K, row0, xv, txv in cases:
        full = w[name].reshape(-1)
        per = (cfg.vocab_size if name == "wcls" else (cfg.hidden_dim if name in ("w1", "w3") else cfg.dim)) * K
        off = layer * per + row0 * K
        want = np.empty(rows, np.float32)
        O.matmul(want, np.ascontiguousarray(full[off:off + rows * K]), xv, K, rows)
        to = up(dev, np.zeros(rows, np.float32))
        a_ptr = getattr(model.weights, name) + 4 * off
        check(dev.lib.rama_matmul(dev.ctx, to.ptr, a_ptr, txv.ptr, K, rows, 1))
        assert_bits_equal(dev.download(to), want, f"matmul view {name} layer {layer} row0 {row0}")
    model.free()


@pytest.mark.parametrize("n,scale", [(1, 1.0), (63, 1.0), (64, 1.0), (65, 3.0), (1000, 0.01), (4096, 1.0), (4097, 50.0), (11008, 1.0), (16384, 1e-3)])
def test_exact_sequential_sums(dev, n, scale):
    """rmsnorm's sum of squares and softmax's sum of exponentials are sequential fp32 sums in the
    reference (Iterator::sum, cpu.rs:112; the oracle's softmax order): the parallel scan must give the
    sequential bits for every length, magnitude and structure (ties, powers of two, zeros, spikes)"""
    rng = np.random.default_rng(n)
    variants = [rnd(n, n, scale), np.full(n, scale, np.float32), (2.0 ** rng.integers(-6, 6, n)).astype(np.float32) * np.float32(scale),
                np.where(rng.random(n) < 0.5, 0.0, scale).astype(np.float32)]
    spike = rnd(n, n + 1, scale); spike[n // 2] = np.float32(1000.0 * scale); variants.append(spike)
    grow = (np.arange(1, n + 1, dtype=np.float32) * np.float32(scale / n)); variants.append(grow)
    for vi, x in enumerate(variants):
        w = rnd(n, 7)
        want = np.empty(n, np.float32)
        O.rmsnorm(want, x, w, n)
        tx = up(dev, x); tw = up(dev, w); to = up(dev, np.zeros(n, np.float32))
        dev.rmsnorm(to, tx.as_view(), tw.as_view(), n)
        assert_bits_equal(dev.download(to), want, f"rmsnorm n={n} variant {vi}")
        s = (x * np.float32(4.0 / max(scale, 1e-6))).astype(np.float32)
        ts = up(dev, s)
        dev.softmax(ts, n)
        O.softmax(s, n)
        assert_bits_equal(dev.download(ts), s, f"softmax n={n} variant {vi}")


def _seq_sum(a):
    return np.add.accumulate(np.ascontiguousarray(a, np.float32), dtype=np.float32)[-1] if len(a) else np.float32(0)


def fast_sum_lists(n, seed):
    """non-negative lists that exercise seqsum_fast.hpp: squares of activations at several scales, all-equal terms (every add a
    tie or none), powers of two, zeros, a spike, a ramp, terms that are exact half-ulps of the running sum, tiny and huge magnitudes"""
    rng = np.random.default_rng(seed)
    out = []
    for scale in (1.0, 0.02, 37.0, 1e-12, 3e15):
        x = (rng.standard_normal(n) * scale).astype(np.float32)
        out.append((x * x).astype(np.float32))
    out.append(np.full(n, 0.75, np.float32))
    out.append(np.full(n, 1.0, np.float32))
    out.append((2.0 ** rng.integers(-8, 8, n)).astype(np.float32))
    out.append(np.where(rng.random(n) < 0.5, 0.0, 1.3).astype(np.float32))
    spike = (rng.random(n).astype(np.float32)); spike[n // 2] = np.float32(5000.0); out.append(spike)
    out.append((np.arange(1, n + 1, dtype=np.float32) * np.float32(1.0 / n)))
    half = rng.random(n).astype(np.float32); half[::7] = np.float32(2.0 ** -13); half[::11] = np.float32(3 * 2.0 ** -14); out.append(half)   # half-ulps of sums in [2^10, 2^11)
    out.append(np.zeros(n, np.float32))
    z = np.zeros(n, np.float32); z[-1] = 2.5; out.append(z)
    out.append((rng.random(n) ** 8).astype(np.float32) * np.float32(100.0))       # a long tail: few large terms
    return out


@pytest.mark.parametrize("n", [1, 7, 64, 65, 288, 512, 768, 1000, 2048, 4096, 4097, 8192, 11008, 16384])
def test_fast_sequential_sum(dev, n):
    """seqsum_fast.hpp (the leader workgroup's exact sum of squares, [r5]): the sequential fp32 sum bit for bit on 1, 2 and 4 waves, and it
    must HOLD (no fallback) on ordinary data"""
    from rama_amd._lib import check
    f = dev.lib.rama_internal_seqsum_fast
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    to = up(dev, np.zeros(4, np.float32))
    for li, a in enumerate(fast_sum_lists(n, n)):
        want = _seq_sum(a)
        ta = up(dev, a)
        for nw in (1, 2, 4):
            if n > 64 * 64 * nw:
                continue
            check(f(dev.ctx, ta.ptr, n, nw, to.ptr))
            got = dev.download(to)
            assert_bits_equal(got[:1], np.array([want], np.float32), f"fast sum n={n} list {li} waves {nw}")
            if li < 3 and n >= 64:
                assert got[1] == 1.0, f"fast sum n={n} list {li} waves {nw}: fell back on ordinary data"


@pytest.mark.parametrize("heads,hs", [(32, 128), (6, 48), (3, 2)])
def test_apply_position_runs_are_one_launch_and_keep_their_order(dev, heads, hs):
    """[r5] a run of Device::apply_position calls on consecutive heads (infer.rs:25-29) is recorded and issued as ONE launch by whatever enters the
    library next: bit for bit what one launch per call gives (cpu.rs:87-96's three roundings) -- in call order with other ops in between, out of order,
    the same head twice, a run followed at once by a download, and with "rope_batch" = 0"""
    from rama_amd._lib import check
    dim = heads * hs
    q0, k0 = rnd(dim, 41, 1.3), rnd(dim, 42, 0.7)
    ang = np.random.default_rng(43).uniform(0, 6.28, hs // 2)
    pr, pi = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    tpr, tpi = up(dev, pr), up(dev, pi)

    def head(t, h):
        return t.mut_slice(h * hs)

    def oracle_heads(q, k, order):
        for h in order:
            qq, kk = q[h * hs:(h + 1) * hs].copy(), k[h * hs:(h + 1) * hs].copy()
            O.apply_position(qq, kk, pr, pi, hs)
            q[h * hs:(h + 1) * hs], k[h * hs:(h + 1) * hs] = qq, kk

    orders = [list(range(heads)), list(reversed(range(heads))), [0, 0, 1] + list(range(1, heads)), [heads - 1]]
    for batch in (1, 0):
        check(dev.lib.rama_set_tuning(dev.ctx, b"rope_batch", batch))
        for order in orders:
            tq, tk = up(dev, q0), up(dev, k0)
            eq, ek = q0.copy(), k0.copy()
            for h in order:
                dev.apply_position(head(tq, h), head(tk, h), tpr.as_view(), tpi.as_view(), hs)
            oracle_heads(eq, ek, order)
            assert_bits_equal(dev.download(tq), eq, f"rope_batch {batch} order {order[:4]} q")      # (the download issues the pending run)
            assert_bits_equal(dev.download(tk), ek, f"rope_batch {batch} order {order[:4]} k")
            # an op between two runs sees the first run's result and the second run sees the op's
            tadd = up(dev, np.full(dim, 0.5, np.float32))
            for h in range(heads):
                dev.apply_position(head(tq, h), head(tk, h), tpr.as_view(), tpi.as_view(), hs)
            dev.array_add(tq, tadd.as_view(), dim)
            for h in range(heads):
                dev.apply_position(head(tq, h), head(tk, h), tpr.as_view(), tpi.as_view(), hs)
            oracle_heads(eq, ek, range(heads)); eq = eq + np.float32(0.5); oracle_heads(eq, ek, range(heads))
            assert_bits_equal(dev.download(tq), eq, f"rope_batch {batch} runs around an op, q")
            assert_bits_equal(dev.download(tk), ek, f"rope_batch {batch} runs around an op, k")
    check(dev.lib.rama_set_tuning(dev.ctx, b"rope_batch", 1))


def test_recorded_ops_keep_program_order(dev):
    """[r5] Device::sinu, Device::copy_from_slice and parity-mode Device::matmul calls are recorded and issued with the call that follows (sinu + array_mult
    on the same vector, two copies, up to three matmuls with the same activations: one launch each); every hazard must fall back to program order: a
    copy that reads what the recorded copy writes, a product into another vector, a matmul whose output is the run's input, a download in between"""
    import rama_amd
    from rama_amd._lib import check
    n = 1000
    a, b, c3 = rnd(n, 51), rnd(n, 52), rnd(n, 53)
    for batch in (1, 0):
        for key in (b"ew_batch", b"matmul_batch", b"rope_batch"):
            check(dev.lib.rama_set_tuning(dev.ctx, key, batch))
        # sinu then the product on the same vector; then a product into ANOTHER vector
        ta, tb, tc = up(dev, a), up(dev, b), up(dev, c3)
        ea = a.copy(); O.sinu(ea, n); ea = (ea * b).astype(np.float32)
        dev.sinu(ta, n); dev.array_mult(ta, tb.as_view(), n)
        assert_bits_equal(dev.download(ta), ea, f"batch {batch}: sinu + array_mult")
        dev.sinu(tb, n); dev.array_mult(tc, tb.as_view(), n)                   # the product reads what the recorded sinu writes
        eb = b.copy(); O.sinu(eb, n)
        assert_bits_equal(dev.download(tc), (c3 * eb).astype(np.float32), f"batch {batch}: sinu, then a product that reads it")
        assert_bits_equal(dev.download(tb), eb, f"batch {batch}: sinu alone")
        # two independent copies; then a chain of copies (the second reads the first one's target)
        t1, t2 = up(dev, np.zeros(n, np.float32)), up(dev, np.zeros(n, np.float32))
        dev.copy_from_slice(t1, ta.as_view(), n); dev.copy_from_slice(t2, tc.as_view(), n)
        assert_bits_equal(dev.download(t1), ea, f"batch {batch}: copy 1 of 2"); assert_bits_equal(dev.download(t2), (c3 * eb).astype(np.float32), f"batch {batch}: copy 2 of 2")
        dev.copy_from_slice(t1, tb.as_view(), n); dev.copy_from_slice(t2, t1.as_view(), n)
        assert_bits_equal(dev.download(t2), eb, f"batch {batch}: a copy that reads the recorded copy's target")
        # matmul runs on uploaded matrices: three with the same x; then one whose output is the run's x
        K, rows = 64, 48
        ws = [rnd(rows * K, 60 + i, 0.3) for i in range(4)]
        x = rnd(K, 70)
        tws = [up(dev, w_) for w_ in ws]
        tx = up(dev, x)
        outs = [up(dev, np.zeros(rows, np.float32)) for _ in range(3)]
        for i in range(3):
            dev.matmul(outs[i], tws[i].as_view(), tx.as_view(), K, rows, 1)
        for i in range(3):
            want = np.empty(rows, np.float32); O.matmul(want, ws[i], x, K, rows)
            assert_bits_equal(dev.download(outs[i]), want, f"batch {batch}: matmul {i} of a run")
        wsq = rnd(K * K, 80, 0.2); twsq = up(dev, wsq)
        y = up(dev, np.zeros(rows, np.float32))
        dev.matmul(y, tws[3].as_view(), tx.as_view(), K, rows, 1)               # recorded ...
        tx2 = up(dev, x)
        dev.matmul(tx2, twsq.as_view(), tx2.as_view(), K, K, 1)                # ... and one that overwrites its own input: not part of any run
        want = np.empty(rows, np.float32); O.matmul(want, ws[3], x, K, rows)
        assert_bits_equal(dev.download(y), want, f"batch {batch}: a recorded matmul in front of an in-place one")
    for key in (b"ew_batch", b"matmul_batch", b"rope_batch"):
        check(dev.lib.rama_set_tuning(dev.ctx, key, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("K,rows,nmat", [(4096, 4096, 3), (4096, 11008, 2), (288, 288, 3), (768, 2048, 2), (512, 1000, 1), (4096, 32000, 1)])
def test_recorded_rmsnorm_rides_with_the_matmul_run(dev, K, rows, nmat):
    """[r5] a parity-mode Device::rmsnorm is recorded; the run of matmuls on its output carries it as the launch's leader workgroup, which also stores
    the normalised vector (infer.rs:19-23, :40-42, :52-53).  Bit for bit the separate launches' results ("norm_fold" = 0), the oracle's on the small
    shapes; hazards fall back to program order: an in-place norm, a download in between, a matmul on ANOTHER vector, an output that overlaps the
    norm's input."""
    from rama_amd._lib import check
    x, g = rnd(K, 91, 1.5), (rnd(K, 92, 0.1) + np.float32(1.0)).astype(np.float32)
    ws = [rnd(rows * K, 93 + i, 0.05) for i in range(nmat)]
    tws = [up(dev, w_) for w_ in ws]
    tg = up(dev, g)
    res = {}
    for fold in (1, 0):
        check(dev.lib.rama_set_tuning(dev.ctx, b"norm_fold", fold))
        tx, txb = up(dev, x), up(dev, np.zeros(K, np.float32))
        outs = [up(dev, np.zeros(rows, np.float32)) for _ in range(nmat)]
        dev.rmsnorm(txb, tx.as_view(), tg.as_view(), K)
        for i in range(nmat):
            dev.matmul(outs[i], tws[i].as_view(), txb.as_view(), K, rows, 1)
        res[fold] = [dev.download(o_) for o_ in outs] + [dev.download(txb)]
        # the norm alone, then a download (nothing takes it along)
        txc = up(dev, np.zeros(K, np.float32))
        dev.rmsnorm(txc, tx.as_view(), tg.as_view(), K)
        assert_bits_equal(dev.download(txc), res[fold][-1], f"fold {fold}: a norm by itself")
        # a norm, then a matmul on another vector: the norm is issued by itself, in front
        txd = up(dev, np.zeros(K, np.float32))
        dev.rmsnorm(txd, tx.as_view(), tg.as_view(), K)
        dev.matmul(outs[0], tws[0].as_view(), txc.as_view(), K, rows, 1)
        assert_bits_equal(dev.download(outs[0]), res[fold][0], f"fold {fold}: a matmul on another vector behind a recorded norm")
        assert_bits_equal(dev.download(txd), res[fold][-1], f"fold {fold}: the norm in front of it")
        # in place (infer.rs:52): never recorded
        txe = up(dev, x)
        dev.rmsnorm(txe, txe.as_view(), tg.as_view(), K)
        dev.matmul(outs[0], tws[0].as_view(), txe.as_view(), K, rows, 1)
        assert_bits_equal(dev.download(txe), res[fold][-1], f"fold {fold}: in-place norm")
        assert_bits_equal(dev.download(outs[0]), res[fold][0], f"fold {fold}: matmul behind an in-place norm")
        if rows == K:     # an output that IS the norm's input: the run may not read x while it is written
            txf, txg = up(dev, x), up(dev, np.zeros(K, np.float32))
            dev.rmsnorm(txg, txf.as_view(), tg.as_view(), K)
            dev.matmul(txf, tws[0].as_view(), txg.as_view(), K, rows, 1)
            assert_bits_equal(dev.download(txf), res[fold][0], f"fold {fold}: a matmul into the norm's input")
    check(dev.lib.rama_set_tuning(dev.ctx, b"norm_fold", 1))
    for i in range(nmat + 1):
        assert_bits_equal(res[1][i], res[0][i], f"recorded norm, result {i}")
    if rows * K <= 4096 * 4096:
        xb = np.empty(K, np.float32); O.rmsnorm(xb, x, g, K)
        assert_bits_equal(res[1][-1], xb, "the normalised vector against the oracle")
        for i in range(nmat):
            want = np.empty(rows, np.float32); O.matmul(want, ws[i], xb, K, rows)
            assert_bits_equal(res[1][i], want, f"matmul {i} against the oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("K,rows", [(4096, 4096), (11008, 4096), (288, 288), (768, 288), (512, 1000)])
def test_array_add_of_a_recorded_matmul_is_its_residual_epilogue(dev, K, rows):
    """[r5] Device::matmul then Device::array_add of its output (infer.rs:35-37, :46-47) is one launch with the residual epilogue: the product stored, the
    target incremented -- the separate launches' bits ("resid_fold" = 0), the oracle's; an add into the run's own input, or of another vector, stays apart"""
    from rama_amd._lib import check
    w, x, t0 = rnd(rows * K, 101, 0.05), rnd(K, 102), rnd(rows, 103)
    tw = up(dev, w)
    want = np.empty(rows, np.float32); O.matmul(want, w, x, K, rows)
    for fold in (1, 0):
        check(dev.lib.rama_set_tuning(dev.ctx, b"resid_fold", fold))
        tx, to, tt = up(dev, x), up(dev, np.zeros(rows, np.float32)), up(dev, t0)
        dev.matmul(to, tw.as_view(), tx.as_view(), K, rows, 1)
        dev.array_add(tt, to.as_view(), rows)
        assert_bits_equal(dev.download(to), want, f"fold {fold}: the product")
        assert_bits_equal(dev.download(tt), (t0 + want).astype(np.float32), f"fold {fold}: the target")
        # the add of ANOTHER vector behind a recorded matmul
        to2, tt2 = up(dev, np.zeros(rows, np.float32)), up(dev, t0)
        dev.matmul(to2, tw.as_view(), tx.as_view(), K, rows, 1)
        dev.array_add(tt2, to.as_view(), rows)
        assert_bits_equal(dev.download(tt2), (t0 + want).astype(np.float32), f"fold {fold}: add of another vector")
        assert_bits_equal(dev.download(to2), want, f"fold {fold}: the matmul in front of it")
        if rows == K:     # x += W x: the target is the run's input
            tx3, to3 = up(dev, x), up(dev, np.zeros(rows, np.float32))
            dev.matmul(to3, tw.as_view(), tx3.as_view(), K, rows, 1)
            dev.array_add(tx3, to3.as_view(), rows)
            assert_bits_equal(dev.download(tx3), (x + want).astype(np.float32), f"fold {fold}: add into the matmul's input")
    check(dev.lib.rama_set_tuning(dev.ctx, b"resid_fold", 1))


@pytest.mark.gpu
@pytest.mark.parametrize("dim,heads", [(4096, 32), (288, 6), (512, 8)])
def test_qkv_run_rotations_and_cache_copies_are_one_launch(dev, dim, heads):
    """[r5] infer.rs:19-33 op by op -- rmsnorm, three matmuls, apply_position per head, two copies into cache rows -- is recorded and issued as ONE launch
    with the Wq|Wk|Wv epilogue; the separate launches' bits ("qkv_fold" = 0) and the oracle's.  Sequences that stop short (fewer heads rotated, one
    copy only, a copy of q) or touch the run's vectors in between are issued in program order."""
    from rama_amd._lib import check
    hs = dim // heads
    x, g = rnd(dim, 111, 1.2), (rnd(dim, 112, 0.1) + np.float32(1.0)).astype(np.float32)
    ws = [rnd(dim * dim, 113 + i, 0.04) for i in range(3)]
    tws = [up(dev, w_) for w_ in ws]
    tg = up(dev, g)
    fr, fi = np.cos(np.arange(hs // 2, dtype=np.float32) * np.float32(0.37)).astype(np.float32), np.sin(np.arange(hs // 2, dtype=np.float32) * np.float32(0.37)).astype(np.float32)
    tfr, tfi = up(dev, fr), up(dev, fi)
    # the oracle
    xb = np.empty(dim, np.float32); O.rmsnorm(xb, x, g, dim)
    want = []
    for i in range(3):
        o_ = np.empty(dim, np.float32); O.matmul(o_, ws[i], xb, dim, dim); want.append(o_)
    q_, k_ = want[0].copy(), want[1].copy()
    for h in range(heads):
        qh, kh = q_[h * hs:(h + 1) * hs], k_[h * hs:(h + 1) * hs]
        O.apply_position(qh, kh, fr, fi, hs)

    def run(n_rot, copies, poke=None):
        tx, txb = up(dev, x), up(dev, np.zeros(dim, np.float32))
        tq, tk, tv = (up(dev, np.zeros(dim, np.float32)) for _ in range(3))
        cache = up(dev, np.zeros(4 * dim, np.float32))
        dev.rmsnorm(txb, tx.as_view(), tg.as_view(), dim)
        for t_, w_ in zip((tq, tk, tv), tws):
            dev.matmul(t_, w_.as_view(), txb.as_view(), dim, dim, 1)
        if poke == "download":
            dev.download(tq)
        for h in range(n_rot):
            dev.apply_position(tq.mut_slice(h * hs, (h + 1) * hs), tk.mut_slice(h * hs, (h + 1) * hs), tfr.as_view(), tfi.as_view(), hs)
        if poke == "add":
            dev.array_add(tq, tv.as_view(), dim)
        srcs = {"k": tk, "v": tv, "q": tq}
        for j, name in enumerate(copies):
            dev.copy_from_slice(cache.mut_slice((j + 1) * dim, (j + 2) * dim), srcs[name].as_view(), dim)
        return [dev.download(t_) for t_ in (tq, tk, tv, cache, txb)]

    for fold in (1, 0):
        check(dev.lib.rama_set_tuning(dev.ctx, b"qkv_fold", fold))
        got = run(heads, "kv")
        assert_bits_equal(got[0], q_, f"fold {fold}: q"); assert_bits_equal(got[1], k_, f"fold {fold}: k"); assert_bits_equal(got[2], want[2], f"fold {fold}: v")
        assert_bits_equal(got[3][dim:2 * dim], k_, f"fold {fold}: key row"); assert_bits_equal(got[3][2 * dim:3 * dim], want[2], f"fold {fold}: value row")
        assert not got[3][:dim].any() and not got[3][3 * dim:].any(), f"fold {fold}: the cache around the two rows"
        assert_bits_equal(got[4], xb, f"fold {fold}: the normalised vector")
        # fewer heads rotated; one copy; a copy of q in front; something in between
        got = run(heads - 1, "kv")
        qp = want[0].copy(); qp[:(heads - 1) * hs] = q_[:(heads - 1) * hs]
        kp = want[1].copy(); kp[:(heads - 1) * hs] = k_[:(heads - 1) * hs]
        assert_bits_equal(got[0], qp, f"fold {fold}: q, one head not rotated"); assert_bits_equal(got[3][dim:2 * dim], kp, f"fold {fold}: key row, one head not rotated")
        got = run(heads, "k")
        assert_bits_equal(got[3][dim:2 * dim], k_, f"fold {fold}: one copy only"); assert_bits_equal(got[0], q_, f"fold {fold}: q, one copy only")
        got = run(heads, "qkv")
        assert_bits_equal(got[3][dim:2 * dim], q_, f"fold {fold}: q copied first"); assert_bits_equal(got[3][2 * dim:3 * dim], k_, f"fold {fold}: then k")
        assert_bits_equal(got[3][3 * dim:], want[2], f"fold {fold}: then v")
        got = run(heads, "kv", poke="download")
        assert_bits_equal(got[3][dim:2 * dim], k_, f"fold {fold}: a download behind the matmuls")
        got = run(heads, "kv", poke="add")
        assert_bits_equal(got[0], (q_ + want[2]).astype(np.float32), f"fold {fold}: an add between rotations and copies")
        assert_bits_equal(got[3][2 * dim:3 * dim], want[2], f"fold {fold}: value row behind it")
    check(dev.lib.rama_set_tuning(dev.ctx, b"qkv_fold", 1))


@pytest.mark.gpu
@pytest.mark.parametrize("K,rows", [(128, 4741), (256, 4096 + 16 * 100), (512, 16 * 257), (64, 16 * 300 + 9)])
def test_half_row_groups_bit_exact(dev, K, rows):
    """[r5] one wave per row group and a number of row groups that does not divide by the compute units: the remainder runs as half groups (8 rows on 32
    lanes; "chain_split").  The oracle's bits with and without, a last group of fewer than 16 (or 8) rows included."""
    from rama_amd._lib import check
    w, x = rnd(rows * K, 121, 0.1), rnd(K, 122)
    tw, tx = up(dev, w), up(dev, x)
    want = np.empty(rows, np.float32); O.matmul(want, w, x, K, rows)
    for split in (1, 0):
        check(dev.lib.rama_set_tuning(dev.ctx, b"chain_split", split))
        to = up(dev, np.full(rows + 32, 7.0, np.float32))
        dev.matmul(rama_view(to, 0, rows), tw.as_view(), tx.as_view(), K, rows, 1)
        got = dev.download(to)
        assert_bits_equal(got[:rows], want, f"split {split}")
        assert (got[rows:] == 7.0).all(), f"split {split}: floats behind the output were written"
    check(dev.lib.rama_set_tuning(dev.ctx, b"chain_split", 1))


def rama_view(t, a, b):
    return t.mut_slice(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_recorded_ops_random_sequences(dev, seed):
    """[r5] the library records parity-mode ops and issues them merged (runs of rotations and matmuls, a norm as the matmuls' leader, an add as the matmul's
    residual epilogue, Wq|Wk|Wv + rotations + cache copies as one launch).  Random sequences of the trait's ops over a small pool of vectors -- with the
    reference's own sequence (infer.rs:19-47) mixed in whole and cut short -- must leave every vector bit for bit as with all recording off."""
    from rama_amd._lib import check
    rng = np.random.default_rng(seed)
    dim, heads = 256, 4
    hs = dim // heads
    npool = 7
    init = [rnd(dim, 200 + i) for i in range(npool)]
    mats = [up(dev, rnd(dim * dim, 300 + i, 0.06)) for i in range(5)]
    gains = [up(dev, (rnd(dim, 310 + i, 0.1) + np.float32(1.0)).astype(np.float32)) for i in range(2)]
    tfr = up(dev, np.cos(np.arange(hs // 2, dtype=np.float32) * np.float32(0.3)).astype(np.float32))
    tfi = up(dev, np.sin(np.arange(hs // 2, dtype=np.float32) * np.float32(0.3)).astype(np.float32))
    # the program: a list of closures over (pool, cache)
    prog = []
    def two(distinct=True):
        a = int(rng.integers(npool)); b = int(rng.integers(npool))
        while distinct and b == a:
            b = int(rng.integers(npool))
        return a, b
    def layer_like(cut):
        x, xb, q, k, v, xb2 = (int(i) for i in rng.permutation(npool)[:6])
        row = int(rng.integers(3))
        ops = [("rmsnorm", xb, x, 0), ("matmul", q, 0, xb), ("matmul", k, 1, xb), ("matmul", v, 2, xb)]
        ops += [("rope", q, k, h) for h in range(heads)]
        ops += [("tocache", row, k), ("tocache", row + 3, v), ("matmul", xb2, 3, q), ("add", x, xb2), ("rmsnorm", xb, x, 1), ("matmul", q, 4, xb), ("matmul", k, 0, xb),
                ("sinu", q), ("mult", q, k), ("matmul", xb2, 1, q), ("add", x, xb2)]
        return ops[:cut]
    while len(prog) < 260:
        r = rng.random()
        if r < 0.25:
            prog += layer_like(int(rng.integers(1, 24)))
        elif r < 0.40:
            a, b = two(); prog.append(("matmul", a, int(rng.integers(5)), b))
        elif r < 0.50:
            a, b = two(distinct=rng.random() < 0.8); prog.append(("rmsnorm", a, b, int(rng.integers(2))))
        elif r < 0.60:
            a, b = two(); prog.append(("add", a, b))
        elif r < 0.68:
            a, b = two(); prog.append(("mult", a, b))
        elif r < 0.74:
            prog.append(("sinu", int(rng.integers(npool))))
        elif r < 0.82:
            a, b = two(); prog.append(("copy", a, b))
        elif r < 0.90:
            a, b = two(); prog.append(("rope", a, b, int(rng.integers(heads))))
        elif r < 0.96:
            prog.append(("tocache", int(rng.integers(6)), int(rng.integers(npool))))
        else:
            prog.append(("fromcache", int(rng.integers(npool)), int(rng.integers(6))))
    keys = (b"rope_batch", b"matmul_batch", b"ew_batch", b"norm_fold", b"resid_fold", b"qkv_fold")
    results = {}
    for on in (1, 0):
        for key in keys:
            check(dev.lib.rama_set_tuning(dev.ctx, key, on))
        pool = [up(dev, a) for a in init]
        cache = up(dev, np.zeros(6 * dim, np.float32))
        for op in prog:
            if op[0] == "matmul":
                dev.matmul(pool[op[1]], mats[op[2]].as_view(), pool[op[3]].as_view(), dim, dim, 1)
            elif op[0] == "rmsnorm":
                dev.rmsnorm(pool[op[1]], pool[op[2]].as_view(), gains[op[3]].as_view(), dim)
            elif op[0] == "add":
                dev.array_add(pool[op[1]], pool[op[2]].as_view(), dim)
            elif op[0] == "mult":
                dev.array_mult(pool[op[1]], pool[op[2]].as_view(), dim)
            elif op[0] == "sinu":
                dev.sinu(pool[op[1]], dim)
            elif op[0] == "copy":
                dev.copy_from_slice(pool[op[1]], pool[op[2]].as_view(), dim)
            elif op[0] == "rope":
                h = op[3]
                dev.apply_position(pool[op[1]].mut_slice(h * hs, (h + 1) * hs), pool[op[2]].mut_slice(h * hs, (h + 1) * hs), tfr.as_view(), tfi.as_view(), hs)
            elif op[0] == "tocache":
                dev.copy_from_slice(cache.mut_slice(op[1] * dim, (op[1] + 1) * dim), pool[op[2]].as_view(), dim)
            elif op[0] == "fromcache":
                dev.copy_from_slice(pool[op[1]], cache.slice(op[2] * dim, (op[2] + 1) * dim), dim)
        results[on] = [dev.download(t) for t in pool] + [dev.download(cache)]
    for key in keys:
        check(dev.lib.rama_set_tuning(dev.ctx, key, 1))
    for i, (a, b) in enumerate(zip(results[1], results[0])):
        assert np.array_equal(bits(a), bits(b)), f"seed {seed}: vector {i} differs between recorded and unrecorded issue"
    assert sum(bool(np.isfinite(a).all() and np.abs(a).max() > 0) for a in results[1]) >= 5, f"seed {seed}: the program degenerated (inf / nan / zeros): nothing is compared"


@pytest.mark.gpu
def test_op_path_at_long_contexts_matches_the_fused_entry(dev):
    """[r5] forward() composed op for op from the trait (recorded and merged launches) against the resident model's fused entry at positions around every
    switch of the parity attention (128: the spread form, 256, 1 000, 1 900), both caches filled with the same rows: logits and the residual stream bit for bit
    (tools/ops_long_check.py is the same check by hand over more positions)"""
    import rama_amd
    from rama_amd._lib import check
    d, h, L, H, V, seq = 4096, 11008, 2, 32, 640, 2048
    cfg = rama_amd.Config(d, h, L, H, H, V, seq, False)
    model = rama_amd.Model.synth(dev, cfg, seed=3)
    ws = rama_amd.TransformerWeights.synth(cfg, 3, dev)
    wv = rama_amd.TransformerWeightsView.from_gpu_ws(ws)
    rs = rama_amd.RunState.from_config(cfg, dev)
    rsv = rama_amd.RunStateView.from_rs(rs)
    ref = rama_amd.Engine(dev, model)
    rng = np.random.default_rng(7)
    kc = (rng.standard_normal(L * seq * d) * 0.5).astype(np.float32)
    vc = (rng.standard_normal(L * seq * d) * 0.5).astype(np.float32)
    try:
        for pos in (127, 128, 256, 1000, 1900):      # (the module's context is in parity mode: the fixture)
            ref.set_buffer("key_cache", kc); ref.set_buffer("value_cache", vc)
            dev.upload_into(rsv.key_cache, kc); dev.upload_into(rsv.value_cache, vc)
            rama_amd.forward(cfg, wv, rsv, 11, pos, dev)
            ref.forward(11, pos)
            assert_bits_equal(dev.download(rsv.logits), ref.logits(), f"op path vs fused entry, logits at position {pos}")
            assert_bits_equal(dev.download(rsv.x), ref.buffer("x", d), f"op path vs fused entry, x at position {pos}")
    finally:
        ref.free(); rs.free(); ws.free(); model.free()


def _chain_lookup(dev, ptr, rows, K):
    f = dev.lib.rama_internal_chain_lookup
    f.restype = C.c_void_p
    f.argtypes = [C.c_void_p, C.c_int, C.c_int]
    return f(ptr, rows, K)


@pytest.mark.parametrize("dim,hidden,heads,layers,vocab,seq,shared", [(4096, 11008, 32, 2, 640, 64, False), (288, 768, 6, 3, 512, 48, True)])
def test_uploaded_weights_run_the_chain_kernels(dev, dim, hidden, heads, layers, vocab, seq, shared):
    """[r5] weights uploaded tensor by tensor (hbm.rs:55-90, what the Rust shim does) are ADOPTED by the fused entry in parity mode -- the same
    chain-order copies and launches as a resident model -- and Device::matmul on a matrix of no model (the 1:1 path's w1 / w3 views too) makes a
    chain-order copy of the tensor on first use: logits and run state bit for bit the oracle's on both paths; the copies go when a tensor is
    freed or overwritten, and a new upload at the same address is not served from a stale copy"""
    import rama_amd
    cfg = O.Config(dim, hidden, layers, heads, heads, vocab, seq, shared)
    rope = S.rope_tables(seq, dim // heads)
    token = 1
    for seed in (21, 22):      # the second round re-uploads other weights, most likely at the same addresses
        w = S.synth_weights(cfg, seed, rope=rope)
        orc = O.Oracle(cfg, w)
        rcfg, ws, wv, rs, rsv = gpu_views(dev, cfg, w)
        rs2 = rama_amd.RunState.from_config(rcfg, dev); rsv2 = rama_amd.RunStateView.from_rs(rs2)
        assert not _chain_lookup(dev, wv.wq.ptr, dim, dim), "a chain-order copy of a tensor that was just uploaded"
        for pos in range(5):
            lo = orc.forward(token, pos)
            rama_amd.forward_fused(rcfg, wv, rsv, token, pos, dev)
            rama_amd.forward(rcfg, wv, rsv2, token, pos, dev)
            for buf in ("logits", "x", "xb", "hb", "q", "key_cache"):
                assert_bits_equal(dev.download(getattr(rsv, buf)), orc.s[buf], f"adopted, fused pos {pos} {buf}")
                assert_bits_equal(dev.download(getattr(rsv2, buf)), orc.s[buf], f"uploaded, 1:1 ops pos {pos} {buf}")
            token = O.argmax(lo)
        # the fused entry adopted the tensors (every matrix, W1|W3 interleaved); the 1:1 ops made copies of w1 and w3 by themselves
        assert _chain_lookup(dev, wv.wq.ptr, dim, dim) and _chain_lookup(dev, wv.w2.ptr, dim, hidden) and _chain_lookup(dev, wv.w1.ptr, 2 * hidden, dim)
        assert _chain_lookup(dev, wv.w1.ptr + 4 * hidden * dim, hidden, dim) and _chain_lookup(dev, wv.w3.ptr, hidden, dim)
        wq_ptr, w3_ptr = wv.wq.ptr, wv.w3.ptr
        # overwriting a tensor dissolves what was derived from it ...
        dev.upload_into(ws.w3, w["w3"])
        assert not _chain_lookup(dev, w3_ptr, hidden, dim) and not _chain_lookup(dev, wq_ptr, dim, dim)
        rama_amd.forward_fused(rcfg, wv, rsv, 1, 5, dev)      # ... and the next call makes it again
        assert _chain_lookup(dev, wq_ptr, dim, dim)
        rs.free(); rs2.free(); ws.free()
        assert not _chain_lookup(dev, wq_ptr, dim, dim), "a chain-order copy outlived its tensor"


@pytest.mark.parametrize("n_heads,hs", [(2, 128), (3, 64)])
def test_model_long_context_bit_exact(dev, n_heads, hs):
    """parity mode over a pre-filled cache at positions around the 4-wave / 16-wave switch (256) and deep into the
    context (the exact sequential softmax sum over ~2000 terms, ~30 value tiles): every logit bit as the oracle's"""
    import rama_amd
    from .helpers import to_rama_cfg
    dim, seq = n_heads * hs, 2048
    cfg = O.Config(dim, 2 * dim, 1, n_heads, n_heads, 64, seq, False)
    rope = S.rope_tables(seq, hs)
    w = S.synth_weights(cfg, 11, rope=rope)
    rng = np.random.default_rng(11)
    kc = rng.standard_normal(seq * dim).astype(np.float32)
    vc = rng.standard_normal(seq * dim).astype(np.float32)
    orc = O.Oracle(cfg, w)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 11, rope=rope)
    eng = rama_amd.Engine(dev, model)
    # [r4] the exact attention is spread over the chip from position 128 on (scores | softmax + value chains in one launch); the
    # forms it replaced stay selectable: "attn_fv" = 0 (softmax and value chains as two launches), "spread_pos" (one workgroup per
    # head below it: 4 waves, 8 from position 256 on)
    variants = [({}, (100, 127, 128, 129, 191, 192, 193, 255, 256, 257, 383, 384, 385, 1000, 1023, 1024, 1025, 2047)),
                ({"attn_fv": 0}, (128, 200, 1024, 2047)),
                ({"spread_pos": 1024}, (255, 256, 257, 1000, 1023, 1024, 1025)),
                ({"spread_pos": 1 << 20}, (1500, 2047))]
    try:
        for tune, positions in variants:
            for k, v in tune.items():
                eng.set_tuning(k, v)
            for graph in (False, True):
                eng.set_graph_mode(graph)
                for pos in positions:
                    orc.s["key_cache"][:] = kc; orc.s["value_cache"][:] = vc
                    eng.set_buffer("key_cache", kc); eng.set_buffer("value_cache", vc)
                    lo = orc.forward(5, pos).copy()
                    eng.forward(5, pos)
                    assert_bits_equal(eng.logits(), lo, f"long context pos {pos} graph {graph} {tune}")
                    assert_bits_equal(eng.buffer("xb2", dim), orc.s["xb2"], f"long context pos {pos} xb2 {tune}")
                    att = eng.buffer("att", n_heads * seq).reshape(n_heads, seq)[:, :pos + 1]
                    assert_bits_equal(att, orc.s["att"].reshape(n_heads, seq)[:, :pos + 1], f"long context pos {pos} att {tune}")
            eng.set_graph_mode(False)
            eng.set_tuning("attn_fv", 1); eng.set_tuning("spread_pos", 128)
    finally:
        eng.set_graph_mode(False)
        eng.set_tuning("attn_fv", 1); eng.set_tuning("spread_pos", 128)
    eng.free(); model.free()


@pytest.mark.parametrize("dim,hidden,heads,layers,seq,bar_pos", [(256, 512, 2, 2, 512, None), (4096, 11008, 32, 2, 2048, None), (288, 768, 6, 2, 256, 16)])
def test_bar_mode_is_parity_until_its_switch_then_the_fast_attention(dev, dim, hidden, heads, layers, seq, bar_pos):
    """[r6] "ref_order" = 3, bar mode: every launch parity mode's -- the oracle's bits -- up to position "bar_pos" (default 128, the spread attention's
    switch), the fast path's attention from there on (one workgroup per head, split-T from 256 at llama2-7B's width) while matvecs and norms stay exact.
    Over a pre-filled cache: below the switch every buffer bit for bit, behind it logits within 5e-5 of the oracle at this depth (2 layers) and the
    greedy token the same; eager and from a hipGraph; the 1:1 Device ops take the same switch.  (Full depth, whole context: tools/tol_sweep.py,
    profiles/r06_*; tests/test_hip_parity_7b.py asserts <= 1e-4 per position there.)"""
    import rama_amd
    from rama_amd._lib import check
    from .helpers import to_rama_cfg
    vocab = 320
    cfg = O.Config(dim, hidden, layers, heads, heads, vocab, seq, False)
    rope = S.rope_tables(seq, dim // heads)
    w = S.synth_weights(cfg, 19, rope=rope)
    rng = np.random.default_rng(19)
    kc = rng.standard_normal(layers * seq * dim, dtype=np.float32)
    vc = rng.standard_normal(layers * seq * dim, dtype=np.float32)
    orc = O.Oracle(cfg, w)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 19, rope=rope)
    eng = rama_amd.Engine(dev, model)
    switch = bar_pos if bar_pos is not None else 128
    positions = sorted({0, 5, switch - 1, switch, switch + 1, min(seq - 1, 255), min(seq - 1, 256), min(seq - 1, 300), seq - 1})
    check(dev.lib.rama_set_tuning(dev.ctx, b"ref_order", 3))
    if bar_pos is not None:
        check(dev.lib.rama_set_tuning(dev.ctx, b"bar_pos", bar_pos))
    try:
        for graph in (False, True):
            eng.set_graph_mode(graph)
            for pos in positions:
                orc.s["key_cache"][:] = kc; orc.s["value_cache"][:] = vc
                eng.set_buffer("key_cache", kc); eng.set_buffer("value_cache", vc)
                lo = orc.forward(9, pos).copy()
                eng.forward(9, pos)
                lg = eng.logits()
                what = f"bar mode pos {pos} graph {graph}"
                if pos < switch:
                    assert_bits_equal(lg, lo, f"{what}: logits (below the switch: parity mode's bits)")
                    assert_bits_equal(eng.buffer("xb2", dim), orc.s["xb2"], f"{what}: xb2")
                else:
                    assert float(np.abs(lg - lo).max()) <= 5e-5, (what, float(np.abs(lg - lo).max()))      # (random cache rows: logits of magnitude ~2, 2 layers)
                    assert int(np.flatnonzero(lg == lg.max())[-1]) == O.argmax(lo), what
                for buf in ("key_cache", "value_cache"):      # the appended rows come from the exact matvecs in either regime... of layer 0 (layer 1's input has passed an attention)
                    got = eng.buffer(buf, seq * dim).reshape(seq, dim)[pos]
                    assert_bits_equal(got, orc.s[buf].reshape(layers, seq, dim)[0, pos], f"{what}: {buf} row of layer 0")
        eng.set_graph_mode(False)
        # the 1:1 op: Device::multi_head_attention behind the switch = the fast kernel (close, not bit-identical by contract), below it the exact one
        rcfg = to_rama_cfg(cfg)
        rs = rama_amd.RunState.from_config(rcfg, dev); rsv = rama_amd.RunStateView.from_rs(rs)
        q = rnd(dim, 5, 1.5)
        dev.upload_into(rsv.key_cache, kc); dev.upload_into(rsv.value_cache, vc); dev.upload_into(rsv.q, q)
        for pos in (switch - 1, switch + 3):
            orc.s["key_cache"][:] = kc; orc.s["value_cache"][:] = vc; orc.s["q"][:] = q
            orc.multi_head_attention(1, pos)
            dev.multi_head_attention(rsv, rcfg, 1, pos)
            got = dev.download(rsv.xb)
            if pos < switch:
                assert_bits_equal(got, orc.s["xb"], f"op path, bar mode, pos {pos}")
            else:
                assert float(np.abs(got - orc.s["xb"]).max()) <= 1e-5
        rs.free()
    finally:
        eng.set_graph_mode(False)
        check(dev.lib.rama_set_tuning(dev.ctx, b"bar_pos", 128))
        check(dev.lib.rama_set_tuning(dev.ctx, b"ref_order", 1))
        eng.free(); model.free()


@pytest.mark.parametrize("dim,hidden,heads,layers,seq,steps", [(288, 768, 6, 3, 256, 200), (2048, 2048, 16, 2, 1100, 300)])
def test_bar_mode_chained_decode_crosses_its_switches(dev, dim, hidden, heads, layers, seq, steps):
    """[r6] the device-chained generate() loop in bar mode, replayed from hipGraphs (four steps per graph at dim <= 1024), straight through position 128 --
    where the graph variant changes from parity mode's exact attention to the fast path's -- and through 256 (the wider shape: a head's K + V cache
    exceeds 1 MiB there, so the fast attention turns split-T): the oracle's greedy tokens at every step, the positions below 128 from parity mode's bits"""
    import rama_amd
    from rama_amd._lib import check
    from .helpers import to_rama_cfg
    cfg = O.Config(dim, hidden, layers, heads, heads, 512, seq, True)
    rope = S.rope_tables(seq, dim // heads)
    w = S.synth_weights(cfg, 23, rope=rope)
    prompt = [17, 4, 99]
    want = O.Oracle(cfg, w).generate_greedy(prompt, steps)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 23, rope=rope)
    check(dev.lib.rama_set_tuning(dev.ctx, b"ref_order", 3))
    try:
        for graph in (True, False):
            eng = rama_amd.Engine(dev, model)
            eng.set_graph_mode(graph)
            got = eng.generate_greedy(prompt, steps)
            eng.set_graph_mode(False)
            first_bad = next((i for i, (a_, b_) in enumerate(zip(got, want)) if a_ != b_), None)
            assert got == want, (graph, first_bad, got[first_bad - 2:first_bad + 3], want[first_bad - 2:first_bad + 3])
            eng.free()
    finally:
        check(dev.lib.rama_set_tuning(dev.ctx, b"ref_order", 1))
        model.free()


def test_llama2_7b_width_deep_context_bit_exact(dev):
    """[r6] the HEADLINE shape's width (dim 4096, hidden 11008, 32 heads of 128, seq_len 2048; 2 of the 32 layers, so that the oracle needs
    1.6 GB instead of 27) deep into the context: both caches filled with the same random rows on both sides, parity mode with the default
    tunings, eager and replayed from a hipGraph, at the positions around every switch of the exact attention (128: spread over the chip;
    256; the 1 024-row tiles) and at its far end (1 900, 2 047: a 2 048-term exact softmax sum, 32 heads x 8 slices of value chains):
    logits, the residual stream, xb, xb2 and every probability bit for bit the ORACLE's (cpu.rs:23-52 multi_head_attention,
    cpu.rs:187-192 softmax_num).  Round 5 had this comparison at 2 x 128 / 3 x 64 heads only (test_model_long_context_bit_exact)."""
    import rama_amd
    from .helpers import to_rama_cfg
    dim, hidden, heads, layers, seq, vocab = 4096, 11008, 32, 2, 2048, 640
    cfg = O.Config(dim, hidden, layers, heads, heads, vocab, seq, False)
    rope = S.rope_tables(seq, dim // heads)
    w = S.synth_weights(cfg, 13, rope=rope)
    rng = np.random.default_rng(13)
    kc = rng.standard_normal(layers * seq * dim, dtype=np.float32)
    vc = rng.standard_normal(layers * seq * dim, dtype=np.float32)
    orc = O.Oracle(cfg, w)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 13, rope=rope)
    eng = rama_amd.Engine(dev, model)
    positions = (127, 128, 255, 256, 257, 1000, 1023, 1024, 1900, 2047)
    try:
        for graph in (False, True):
            eng.set_graph_mode(graph)
            for pos in positions:
                orc.s["key_cache"][:] = kc; orc.s["value_cache"][:] = vc
                eng.set_buffer("key_cache", kc); eng.set_buffer("value_cache", vc)
                lo = orc.forward(7, pos).copy()
                eng.forward(7, pos)
                what = f"7B width, pos {pos}, graph {graph}"
                assert_bits_equal(eng.logits(), lo, f"{what}: logits")
                for buf, n in (("x", dim), ("xb", dim), ("xb2", dim), ("hb", hidden), ("q", dim)):
                    assert_bits_equal(eng.buffer(buf, n), orc.s[buf], f"{what}: {buf}")
                # (the probabilities of the LAST layer: att is one [heads, seq] scratch, cpu.rs:29)
                att = eng.buffer("att", heads * seq).reshape(heads, seq)[:, :pos + 1]
                assert_bits_equal(att, orc.s["att"].reshape(heads, seq)[:, :pos + 1], f"{what}: att")
                for buf in ("key_cache", "value_cache"):      # the appended rows of both layers
                    got = eng.buffer(buf, layers * seq * dim).reshape(layers, seq, dim)[:, pos]
                    assert_bits_equal(got, orc.s[buf].reshape(layers, seq, dim)[:, pos], f"{what}: {buf} row")
    finally:
        eng.set_graph_mode(False)
        eng.free(); model.free()


@pytest.mark.parametrize("shape,n_tokens,pos0", [((64, 176, 2, 4, 96, 48), 2, 0), ((64, 176, 2, 4, 96, 48), 5, 0), ((64, 176, 2, 4, 96, 48), 6, 3),
                                                 ((128, 352, 2, 2, 256, 80), 9, 0), ((128, 352, 2, 2, 256, 80), 17, 0), ((128, 352, 2, 2, 256, 80), 18, 5),
                                                 ((288, 768, 2, 6, 512, 96), 40, 0), ((272, 720, 2, 17, 333, 64), 21, 2),
                                                 ((768, 2048, 1, 12, 1024, 64), 34, 0), ((48, 80, 2, 3, 50, 40), 12, 1)])
def test_parity_prefill_bit_exact(dev, shape, n_tokens, pos0):
    """parity mode's rama_prefill: the forced positions go through the chain-order token-batch kernels 16 at a time
    (csrc/chain.hpp gemm_chain_kernel; every weight read once per 16 positions) and the last one through forward().
    Cache rows of every position, the logits and the run state of the last are the ORACLE's, bit for bit -- and
    therefore those of one forward() per position ("prefill_chain" = 0), which is compared too.  Shapes: widths
    that are no whole 256-float chunk, rows that are no multiple of 16, one to three passes, partly filled passes of 1-3,
    5-8 and 9-16 tokens (the three instantiations), a start position > 0."""
    import rama_amd
    from rama_amd._lib import check
    from .helpers import to_rama_cfg
    dim, hidden, L, H, V, seq = shape
    cfg = O.Config(dim, hidden, L, H, H, V, seq, False)
    rope = S.rope_tables(seq, dim // H)
    w = S.synth_weights(cfg, 7, rope=rope)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 7, rope=rope)
    rng = np.random.default_rng(n_tokens)
    toks = [1] + [int(t) for t in rng.integers(0, V, pos0 + n_tokens - 1)]
    orc = O.Oracle(cfg, w)
    for pos, t in enumerate(toks):
        lo = orc.forward(t, pos)
    arr = (C.c_int32 * n_tokens)(*toks[pos0:])
    outs = {}
    for batched in (1, 0):
        eng = rama_amd.Engine(dev, model)
        for pos in range(pos0):
            eng.forward(toks[pos], pos)
        check(dev.lib.rama_set_tuning(dev.ctx, b"prefill_chain", batched))
        try:
            check(dev.lib.rama_prefill(dev.ctx, C.byref(model.ccfg), C.byref(model.weights), C.byref(eng.state), arr, n_tokens, pos0), "rama_prefill")
        finally:
            check(dev.lib.rama_set_tuning(dev.ctx, b"prefill_chain", 1))
        outs[batched] = {"logits": eng.logits(), "key_cache": eng.buffer("key_cache", L * seq * dim), "value_cache": eng.buffer("value_cache", L * seq * dim),
                         "x": eng.buffer("x", dim), "xb": eng.buffer("xb", dim), "hb": eng.buffer("hb", hidden), "q": eng.buffer("q", dim)}
        assert_bits_equal(outs[batched]["logits"], lo, f"logits, batched={batched}")
        for buf in ("key_cache", "value_cache", "x", "xb", "hb", "q"):
            assert_bits_equal(outs[batched][buf], orc.s[buf], f"{buf}, batched={batched}")
        eng.free()
    model.free()


@pytest.mark.parametrize("shape,n_seq", [((64, 176, 2, 4, 96, 24), 2), ((64, 176, 2, 4, 96, 24), 5), ((128, 352, 2, 2, 250, 24), 16),
                                         ((128, 352, 2, 2, 250, 24), 17), ((288, 768, 2, 6, 512, 24), 40)])
def test_parity_decode_batch_bit_exact(dev, shape, n_seq):
    """parity mode's rama_decode_batch: independent sequences at different positions share every weight pass, 16 at a
    time, through the chain-order token-batch kernels; each sequence's logits and cache rows are the ORACLE's for its own
    forward(), bit for bit, over several steps"""
    import rama_amd
    from .helpers import to_rama_cfg
    dim, hidden, L, H, V, seq = shape
    cfg = O.Config(dim, hidden, L, H, H, V, seq, True)
    rope = S.rope_tables(seq, dim // H)
    w = S.synth_weights(cfg, 9, rope=rope)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 9, rope=rope)
    batch = [rama_amd.Engine(dev, model) for _ in range(n_seq)]
    orcs = [O.Oracle(cfg, w) for _ in range(n_seq)]
    rng = np.random.default_rng(n_seq)
    cur = [int(t) for t in rng.integers(0, V, n_seq)]
    pos = [0] * n_seq
    for i in range(n_seq):                        # stagger: sequence i is advanced alone i % 4 times first
        for _ in range(i % 4):
            lo = orcs[i].forward(cur[i], pos[i]); batch[i].forward(cur[i], pos[i])
            cur[i] = O.argmax(lo); pos[i] += 1
    for _ in range(3):
        rama_amd.decode_batch(batch, cur, pos)
        for i in range(n_seq):
            lo = orcs[i].forward(cur[i], pos[i])
            assert_bits_equal(batch[i].logits(), lo, f"sequence {i} position {pos[i]} logits")
            cur[i] = O.argmax(lo); pos[i] += 1
    for i in (0, n_seq // 2, n_seq - 1):
        for buf in ("key_cache", "value_cache"):
            assert_bits_equal(batch[i].buffer(buf, orcs[i].s[buf].size), orcs[i].s[buf], f"sequence {i} {buf}")
    for e in batch: e.free()
    model.free()


def test_release_copies_and_rebuild(dev):
    """rama_model_release_copies gives the chain-order / tile-order copies back (92 -> 38 GB at llama2-7B for a server that
    only decodes in fast mode); the next parity-mode call makes the chain-order copy again and produces the same bits;
    rama_stream_query reports an idle stream as 0"""
    import rama_amd
    from rama_amd._lib import check
    from .helpers import to_rama_cfg
    cfg, w, g = load_case("synth_d288_h6")
    toks = g["tokens"].tolist()[:6]
    orc = O.Oracle(cfg, w)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), int(g["seed"]), rope=(g["freq_cis_real"], g["freq_cis_imag"]))
    eng = rama_amd.Engine(dev, model)
    eng.set_graph_mode(True)
    try:
        for rnd_ in range(3):
            orc = O.Oracle(cfg, w)
            for pos, t in enumerate(toks):
                lo = orc.forward(t, pos)
                eng.forward(t, pos)
                assert_bits_equal(eng.logits(), lo, f"round {rnd_} pos {pos}")
            model.release_copies(3 if rnd_ == 0 else 1)
            dev.sync()
            assert dev.lib.rama_stream_query(dev.ctx) == 0
    finally:
        eng.set_graph_mode(False)
    with pytest.raises(rama_amd.RamaError):
        check(dev.lib.rama_model_release_copies(dev.ctx, model.handle, 8))
    eng.free(); model.free()
