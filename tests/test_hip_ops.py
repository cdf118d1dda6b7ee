"""-m gpu: every Device<T> op (device.rs:3-24) through the C ABI against the CPU oracle on
the same seeded inputs.  Elementwise / copy ops must be bit-exact; reductions are held to
a few fp32 ulps of the magnitude involved (tolerances written at each check)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from oracle import synth as S
from tests.helpers import synth_at

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import rama_amd
    d = rama_amd.Hip(0)
    yield d
    d.close()


def up(dev, a):
    import rama_amd
    return rama_amd.MutView(dev.allocate(a))


def rnd(n, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(n) * scale).astype(np.float32)


# ------------------------------------------------------------------ elementwise (bit-exact)

@pytest.mark.parametrize("n", [1, 4, 288, 1000, 11008, 70001])
def test_array_add_mult_copy_exact(dev, n):
    a, b = rnd(n, 1), rnd(n, 2)
    ta = up(dev, a); tb = up(dev, b)
    dev.array_add(ta, tb.as_view(), n)
    ea = a.copy(); O.array_add(ea, b, n)
    assert np.array_equal(dev.download(ta), ea)
    dev.array_mult(ta, tb.as_view(), n)
    O.array_mult(ea, b, n)
    assert np.array_equal(dev.download(ta), ea)
    tc = up(dev, np.zeros(n, np.float32))
    dev.copy_from_slice(tc, ta.as_view(), n)
    assert np.array_equal(dev.download(tc), ea)


def test_elementwise_n_zero_is_noop(dev):
    a = rnd(8, 3)
    ta = up(dev, a)
    dev.array_add(ta, ta.as_view(), 0)
    dev.sinu(ta, 0)
    assert np.array_equal(dev.download(ta), a)


@pytest.mark.parametrize("n", [5, 768, 11008])
def test_sinu(dev, n):
    a = rnd(n, 4, 3.0)
    a[:3] = [0.0, -30.0, 30.0][:min(3, n)]
    ta = up(dev, a)
    dev.sinu(ta, n)
    e = a.copy(); O.sinu(e, n)
    # expf differs by <= 2 ulp between libm and the device; silu amplifies by |a| <= 30
    np.testing.assert_allclose(dev.download(ta), e, rtol=3e-6, atol=1e-7)


@pytest.mark.parametrize("hs", [2, 16, 48, 64, 128])
def test_apply_position(dev, hs):
    q, k = rnd(hs, 5), rnd(hs, 6)
    ang = np.random.default_rng(7).uniform(0, 6.28, hs // 2)
    pr, pi = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    tq, tk, tpr, tpi = up(dev, q), up(dev, k), up(dev, pr), up(dev, pi)
    dev.apply_position(tq, tk, tpr.as_view(), tpi.as_view(), hs)
    eq, ek = q.copy(), k.copy()
    O.apply_position(eq, ek, pr, pi, hs)
    # a*c - b*s may contract to one FMA on the device: <= 1 ulp of the larger product
    np.testing.assert_allclose(dev.download(tq), eq, rtol=0, atol=4e-7)
    np.testing.assert_allclose(dev.download(tk), ek, rtol=0, atol=4e-7)


@pytest.mark.parametrize("n", [4, 288, 768, 4096])
def test_rmsnorm(dev, n):
    x, w = rnd(n, 8), (1.0 + rnd(n, 9, 0.1)).astype(np.float32)
    to = up(dev, np.zeros(n, np.float32))
    dev.rmsnorm(to, up(dev, x).as_view(), up(dev, w).as_view(), n)
    e = np.zeros(n, np.float32); O.rmsnorm(e, x, w, n)
    np.testing.assert_allclose(dev.download(to), e, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("n", [1, 7, 256, 32000])
def test_softmax(dev, n):
    x = rnd(n, 10, 4.0)
    tx = up(dev, x)
    dev.softmax(tx, n)
    e = x.copy(); O.softmax(e, n)
    got = dev.download(tx)
    # the oracle's (= reference's) SEQUENTIAL fp32 sum of n positive terms is itself off by
    # ~sqrt(n)*eps relative; the device tree sum is closer to exact
    np.testing.assert_allclose(got, e, rtol=max(1e-5, 4e-7 * n ** 0.5), atol=1e-12)
    ex = np.exp(x.astype(np.float64) - float(x.max())); ex /= ex.sum()
    np.testing.assert_allclose(got, ex, rtol=5e-6, atol=1e-12)
    assert abs(float(got.astype(np.float64).sum()) - 1.0) < 1e-5


# ------------------------------------------------------------------ matmul

MATVEC_SHAPES = [   # (rows, width): reference shapes + ragged tails
    (288, 288), (768, 288), (288, 768), (32000, 288),      # stories15M (width 288 = 256 + 32)
    (768, 768), (2048, 768), (768, 2048),                  # stories110M
    (4096, 4096), (11008, 4096), (4096, 11008),            # llama2-7B layer
    (5, 4), (3, 260), (1, 1024), (7, 1028), (9, 12),       # row tails / tiny widths
]


@pytest.mark.parametrize("rows,width", MATVEC_SHAPES)
def test_matmul_vs_oracle(dev, rows, width):
    W = rnd(rows * width, rows + width, 0.05)
    x = rnd(width, 11)
    to = up(dev, np.full(rows, 7.0, np.float32))
    dev.matmul(to, up(dev, W).as_view(), up(dev, x).as_view(), width, rows, 1)
    e = np.zeros(rows, np.float32); O.matmul(e, W, x, width, rows, 1)
    # both are fp32 sums of `width` products in different orders; bound: ~ sqrt(width) * eps * |W||x|
    tol = 4e-7 * np.sqrt(width) * float(np.abs(W).max() * np.abs(x).max()) * 8 + 1e-7
    got = dev.download(to)
    assert np.abs(got - e).max() <= tol, (np.abs(got - e).max(), tol)
    # and against exact arithmetic
    ex = (W.reshape(rows, width).astype(np.float64) @ x.astype(np.float64))
    assert np.abs(got - ex).max() <= tol


def test_matmul_one_hot_is_exact_column(dev):
    """W . e_j is column j bit for bit (every other product is an exact 0)."""
    rows, width = 1000, 1028
    W = rnd(rows * width, 12)
    for j in (0, 255, 256, 1027):
        x = np.zeros(width, np.float32); x[j] = 1.0
        to = up(dev, np.zeros(rows, np.float32))
        dev.matmul(to, up(dev, W).as_view(), up(dev, x).as_view(), width, rows, 1)
        assert np.array_equal(dev.download(to), W.reshape(rows, width)[:, j])


def test_matmul_known_answer_and_ocols(dev):
    """gpu.rs:249-288 test_blas data (k zero-padded to 4): the o_cols = 2 product."""
    l = np.array([3, 3, 3, 0, 3, 5, 3, 0, 3, 3, 3, 0, 3, 3, 3, 0], dtype=np.float32)
    r = np.array([2, 2, 2, 2, 2, 2, 0, 0], dtype=np.float32)
    to = up(dev, np.ones(8, np.float32))
    dev.matmul(to, up(dev, l).as_view(), up(dev, r).as_view(), 4, 4, 2)
    assert dev.download(to).tolist() == [18, 18, 22, 22, 18, 18, 18, 18]


def test_matmul_width_not_multiple_of_4_is_an_error(dev):
    import rama_amd
    to = up(dev, np.zeros(4, np.float32))
    with pytest.raises(rama_amd.RamaError):   # reference CPU body panics (cpu.rs:142-143)
        dev.matmul(to, up(dev, rnd(12, 1)).as_view(), up(dev, rnd(3, 2)).as_view(), 3, 4, 1)


def test_matmul_unaligned_view(dev):
    """a View that starts 4 bytes into its storage (legal in the trait) takes the scalar path"""
    import rama_amd
    rows, width = 6, 16
    W = rnd(rows * width + 1, 13); x = rnd(width + 1, 14)
    sw, sx = dev.allocate(W), dev.allocate(x)
    to = up(dev, np.zeros(rows, np.float32))
    dev.matmul(to, rama_amd.View(sw).slice(1), rama_amd.View(sx).slice(1), width, rows, 1)
    e = np.zeros(rows, np.float32); O.matmul(e, W[1:].copy(), x[1:].copy(), width, rows, 1)
    np.testing.assert_allclose(dev.download(to), e, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ attention

@pytest.mark.parametrize("n_heads,hs,seq_len,positions", [
    (4, 16, 32, [0, 1, 5, 31]),
    (6, 48, 256, [0, 3, 17, 255]),
    (12, 64, 64, [0, 2, 63]),
    (2, 128, 300, [0, 1, 64, 299]),
    (1, 128, 16, [0, 15]),
    (3, 4, 8, [0, 7]),
])
def test_multi_head_attention(dev, n_heads, hs, seq_len, positions):
    import rama_amd
    dim, L = n_heads * hs, 2
    cfg = O.Config(dim, 4 * dim, L, n_heads, n_heads, 8, seq_len, True)
    rcfg = rama_amd.Config(dim, 4 * dim, L, n_heads, n_heads, 8, seq_len, True)
    orc = O.Oracle(cfg, dict(token_embedding_table=np.zeros((8, dim), np.float32), rms_att_weight=np.zeros((L, dim), np.float32),
                             rms_ffn_weight=np.zeros((L, dim), np.float32), wq=np.zeros(1, np.float32), wk=np.zeros(1, np.float32),
                             wv=np.zeros(1, np.float32), wo=np.zeros(1, np.float32), w1=np.zeros(1, np.float32), w2=np.zeros(1, np.float32),
                             w3=np.zeros(1, np.float32), rms_final_weight=np.zeros(dim, np.float32),
                             freq_cis_real=np.zeros(1, np.float32), freq_cis_imag=np.zeros(1, np.float32)))
    kc = rnd(L * seq_len * dim, 20); vc = rnd(L * seq_len * dim, 21); q = rnd(dim, 22, 2.0)
    rs = rama_amd.RunState.from_config(rcfg, dev)
    rsv = rama_amd.RunStateView.from_rs(rs)
    dev.upload_into(rsv.key_cache, kc); dev.upload_into(rsv.value_cache, vc); dev.upload_into(rsv.q, q)
    for layer in (0, 1):
        for pos in positions:
            orc.s["key_cache"][:] = kc; orc.s["value_cache"][:] = vc; orc.s["q"][:] = q
            orc.multi_head_attention(layer, pos)
            dev.multi_head_attention(rsv, rcfg, layer, pos)
            got = dev.download(rsv.xb)
            np.testing.assert_allclose(got, orc.s["xb"], rtol=2e-5, atol=3e-6)
            att = dev.download(rsv.att).reshape(n_heads, seq_len)[:, :pos + 1]
            np.testing.assert_allclose(att, orc.s["att"].reshape(n_heads, seq_len)[:, :pos + 1], rtol=2e-5, atol=1e-7)
            np.testing.assert_allclose(att.sum(axis=1), 1.0, atol=1e-5)
    rs.free()


def test_attention_uniform_keys_average_values(dev):
    """property: identical keys => uniform softmax => output = mean of the cached values"""
    import rama_amd
    n_heads, hs, seq_len = 2, 64, 128
    dim = n_heads * hs
    rcfg = rama_amd.Config(dim, dim, 1, n_heads, n_heads, 8, seq_len, True)
    rs = rama_amd.RunState.from_config(rcfg, dev); rsv = rama_amd.RunStateView.from_rs(rs)
    k = np.tile(rnd(dim, 30), seq_len); v = rnd(seq_len * dim, 31)
    dev.upload_into(rsv.key_cache, k); dev.upload_into(rsv.value_cache, v); dev.upload_into(rsv.q, rnd(dim, 32))
    pos = 99
    dev.multi_head_attention(rsv, rcfg, 0, pos)
    want = v.reshape(seq_len, dim)[:pos + 1].astype(np.float64).mean(axis=0)
    np.testing.assert_allclose(dev.download(rsv.xb), want, atol=2e-6)
    rs.free()


# ------------------------------------------------------------------ sampling

def test_sample_argmax_ties_last_index(dev):
    import rama_amd
    n = 32000
    x = rnd(n, 40)
    x[123] = x[31999] = x[17000] = 9.0
    cfg = rama_amd.Config(4, 4, 1, 1, 1, n, 4, True)
    rs = rama_amd.RunState.from_config(cfg, dev); rsv = rama_amd.RunStateView.from_rs(rs)
    dev.upload_into(rsv.logits, x)
    assert dev.sample(cfg, rsv, 0.0, 0.9) == 31999 == O.argmax(x)
    x[31999] = 0.0
    dev.upload_into(rsv.logits, x)
    assert dev.sample(cfg, rsv, 0.0, 0.9) == 17000 == O.argmax(x)
    rs.free()


@pytest.mark.parametrize("temperature,topp,u", [(1.0, 0.9, 0.2721174359321594), (0.7, 0.9, 0.5), (1.5, 0.5, 0.03743588924407959), (1.0, 1.0, 0.999)])
def test_sample_topp_matches_oracle(dev, temperature, topp, u):
    import rama_amd
    n = 4096
    x = rnd(n, 41, 3.0)
    cfg = rama_amd.Config(4, 4, 1, 1, 1, n, 4, True)
    rs = rama_amd.RunState.from_config(cfg, dev); rsv = rama_amd.RunStateView.from_rs(rs)
    dev.upload_into(rsv.logits, x)
    assert dev.sample(cfg, rsv, temperature, topp, u) == O.sample(x.copy(), temperature, topp, u)
    rs.free()


# ------------------------------------------------------------------ synthetic fill (bit-exact)

@pytest.mark.parametrize("n,seed,tag,offset,bias", [(1, 0, 1, 0, 0.0), (1000, 3, 5, 0, 0.0), (70001, 9, 12, 1 << 33, 0.0), (4096, 2, 7, 12345, 1.0)])
def test_fill_synth_bit_exact(dev, n, seed, tag, offset, bias):
    sc = np.float32(0.02 / S.IH4_STD)
    s = dev.alloc(n)
    import rama_amd
    rama_amd._lib.check(dev.lib.rama_fill_synth(dev.ctx, s.ptr, n, seed, tag, offset, sc, np.float32(bias)))
    assert np.array_equal(dev.download(s), O.fill_synth(n, seed, tag, sc, bias, offset))


# ------------------------------------------------------------------ top-p sampling on the device

def _topp_dev(dev, x, temperature, topp, u):
    """rama_sample_topp_dev on a host vector -> token id"""
    import ctypes as C
    from rama_amd._lib import check
    d_x = dev.allocate(x)
    d_r = dev.allocate(np.zeros(1, dtype=np.float32))
    check(dev.lib.rama_sample_topp_dev(dev.ctx, d_x.ptr, x.size, temperature, topp, u, d_r.ptr))
    out = dev.download(d_r).view(np.int32)[0]
    d_x.free(); d_r.free()
    return int(out)


@pytest.mark.parametrize("n,scale,seed", [(32000, 3.0, 1), (32000, 0.05, 2), (32000, 12.0, 3), (4097, 2.0, 4), (13, 1.0, 5), (2, 1.0, 6), (70001, 4.0, 7)])
@pytest.mark.parametrize("temperature,topp,u", [(1.0, 0.9, 0.2721174359321594), (0.7, 0.9, 0.5), (1.5, 0.5, 0.03743588924407959),
                                                (1.0, 1.0, 0.999), (0.3, 0.95, 0.0), (1.0, 0.2, 0.75)])
def test_sample_topp_dev_matches_oracle(dev, n, scale, seed, temperature, topp, u):
    """device sampler == the oracle's Device::sample restatement on peaked (scale 12), ordinary and
    nearly flat (scale 0.05: every token is a candidate, 8 LDS chunks of running sums) logits"""
    x = rnd(n, seed, scale)
    got, want = _topp_dev(dev, x, temperature, topp, u), O.sample(x.copy(), temperature, topp, u)
    if got != want:
        assert got >= 0 and want >= 0, (got, want)     # -1 = no candidate above the cutoff, on both sides or neither
        # Only legitimate where the draw lands in the far tail: there the fp32 running sum sits on
        # plateaus (p_i << ulp(cum)), so the index depends on the last bit of the softmax's sum,
        # whose order the reference itself does not fix (rayon).  The two picks must
        # then be neighbours in cumulative mass.
        z = x.astype(np.float64) / (temperature if temperature < 1.0 else 1.0)
        pr = np.exp(z - z.max()); pr /= pr.sum()
        order = np.argsort(-pr, kind="stable")
        cum = np.cumsum(pr[order])
        rank = {int(t): i for i, t in enumerate(order)}
        assert abs(cum[rank[got]] - cum[rank[want]]) < 2e-6, (got, want)
        assert u * min(topp, 1.0) > 0.9, "a mismatch away from the tail is a bug"


@pytest.mark.parametrize("n", [32000, 5000, 2500])
@pytest.mark.parametrize("temperature,topp,u", [(1.0, 0.9, 0.2721174359321594), (0.7, 0.95, 0.5)])
def test_sample_topp_dev_masked_vocabulary(dev, n, temperature, topp, u):
    """[r5] whole 1 024-entry stretches of -inf logits: the per-stretch softmax partial has no mass (it was exp(-inf - -inf) = nan, and every
    probability with it); the oracle's token on every path"""
    from rama_amd._lib import check
    x = rnd(n, 31 + n % 7, 1.5)
    x[1024:2048] = -np.inf
    if n > 4096:
        x[3072:4096] = -np.inf
    x[n - 1100:] = -np.inf
    want = O.sample(x.copy(), temperature, topp, u)
    assert want >= 0
    for key, val in ((b"topp_block", 1024), (b"topp_block", 512), (b"topp_block", 2048)):      # (2 048: round 3's sort, every workgroup its own statistics)
        check(dev.lib.rama_set_tuning(dev.ctx, key, val))
        try:
            assert _topp_dev(dev, x, temperature, topp, u) == want, (key, val)
        finally:
            check(dev.lib.rama_set_tuning(dev.ctx, b"topp_block", 1024))


@pytest.mark.parametrize("n", [2, 65, 2047, 2048, 2049, 4097, 20000, 32000, 32768])
def test_sample_topp_dev_pair_ranking_equals_lds_ranking(dev, n):
    """the sorted blocks are merged into one order either by topp_rank_kernel (every workgroup searches all other blocks in its
    LDS, "topp_pairs" = 0) or by one workgroup per (block, block) pair + integer atomics + a scatter launch (default): the same
    token as each other and as the oracle for every draw -- flat, ordinary and peaked lists, lists of a few distinct values (equal
    probabilities across blocks: the earlier block's entry first), a block with nothing kept, nothing kept at all"""
    from rama_amd._lib import check
    rng = np.random.default_rng(n)
    spike = np.full(n, -30.0, np.float32); spike[rng.integers(0, n, 5)] = 4.0
    half = rnd(n, n + 5, 2.0); half[: n // 2] = -40.0                                       # the first blocks keep nothing
    lists = [rnd(n, n + 1, 0.05), rnd(n, n + 2, 1.0), rnd(n, n + 3, 3.0), rnd(n, n + 4, 9.0), np.zeros(n, np.float32),
             (rng.integers(0, 4, n).astype(np.float32) * np.float32(0.6931472)), spike, half]
    draws = [(1.0, 0.9, 0.2721174359321594), (1.0, 0.9, 0.0), (1.0, 0.9, 0.999999), (0.6, 0.95, 0.5), (1.0, 0.05, 0.7), (1.0, 1.0, 0.93)]
    try:
        for li, x in enumerate(lists):
            for temperature, topp, u in draws:
                got = {}
                for mode, block in ((1, 1024), (1, 512), (1, 2048), (0, 2048)):       # small blocks + statistics once (default: 1024) | round 3's blocks, ranked by pairs | round 3
                    check(dev.lib.rama_set_tuning(dev.ctx, b"topp_pairs", mode))
                    check(dev.lib.rama_set_tuning(dev.ctx, b"topp_block", block))
                    got[(mode, block)] = _topp_dev(dev, x, temperature, topp, u)
                assert got[(1, 2048)] == got[(0, 2048)], (n, li, temperature, topp, u, got)
                assert got[(1, 1024)] == got[(1, 512)], (n, li, temperature, topp, u, got)       # the same statistics, another block size
                if got[(1, 512)] != got[(1, 2048)]:
                    # the small-block path folds the softmax sum from per-1024 partial sums: another (equally unpinned) summation
                    # order, so the sum may differ in its last bit and with it a pick that sits on a boundary; both must then be neighbours
                    z = x.astype(np.float64) / (temperature if temperature < 1.0 else 1.0)
                    pr = np.exp(z - z.max()); pr /= pr.sum()
                    order = np.argsort(-pr, kind="stable")
                    cum = np.cumsum(pr[order])
                    rank = {int(t): i for i, t in enumerate(order)}
                    a, b_ = got[(1, 512)], got[(1, 2048)]
                    assert a >= 0 and b_ >= 0 and abs(cum[rank[a]] - cum[rank[b_]]) < 2e-6, (n, li, temperature, topp, u, got)
    finally:
        check(dev.lib.rama_set_tuning(dev.ctx, b"topp_pairs", 1))
        check(dev.lib.rama_set_tuning(dev.ctx, b"topp_block", 1024))


def test_sample_topp_dev_ties_keep_index_order(dev):
    """equal probabilities: the reference's stable sort keeps ascending index order among them"""
    x = np.full(1000, -3.0, dtype=np.float32)
    x[[17, 400, 401, 999]] = 2.0                      # four equal maxima, p ~ 0.249 each
    for u in (0.0, 0.26, 0.51, 0.76, 0.99):
        assert _topp_dev(dev, x, 1.0, 0.9, u) == O.sample(x.copy(), 1.0, 0.9, u)


@pytest.mark.parametrize("n", [2047, 2048, 2049, 4096, 31999, 32768, 32769, 50257, 131072])
@pytest.mark.parametrize("scale", [0.05, 3.0, 12.0])
def test_sample_topp_dev_both_orderings_agree(dev, n, scale):
    """ranking through LDS (n <= 32768) and through global memory (csrc/topp_sort.hpp topp_rank_global_kernel: any n,
    "topp_sort" = 0) give the same token as the oracle at the block-size boundaries and beyond one workgroup's LDS"""
    from rama_amd._lib import check
    x = rnd(n, 100 + n % 97, scale)
    x[[0, n // 2, n - 1]] = x.max()                      # equal maxima in the first, a middle and the last block
    for temperature, topp, u in [(1.0, 0.9, 0.2721174359321594), (0.8, 0.95, 0.6), (1.0, 1.0, 0.97)]:
        want = O.sample(x.copy(), temperature, topp, u)
        got = {}
        try:
            for mode in (1, 0):
                check(dev.lib.rama_set_tuning(dev.ctx, b"topp_sort", mode))
                got[mode] = _topp_dev(dev, x, temperature, topp, u)
        finally:
            check(dev.lib.rama_set_tuning(dev.ctx, b"topp_sort", 1))
        assert got[1] == got[0], (n, scale, temperature, topp, u, got, want)
        # away from the far tail (see the test above); a flat distribution over 131072 entries is skipped: its 1e5 running
        # sums of 7.6e-6 each round by 0.4 % of an element per add, so one ulp in the softmax sum (whose order the
        # reference does not fix) moves the drawn index by thousands
        if u * topp <= 0.9 and not (n > 100000 and scale < 1.0):
            assert got[1] == want, (n, scale, temperature, topp, u, got, want)


def test_sample_topp_dev_many_ties_across_blocks(dev):
    """every logit equal: 32000 equal probabilities, all kept -- the order is the index order, in and
    across the 2048-entry sorting blocks (infer.rs:64 stable sort)"""
    x = np.full(32000, 0.25, dtype=np.float32)
    for u in (0.0, 0.1, 0.5, 0.9, 0.999):
        assert _topp_dev(dev, x, 1.0, 0.9, u) == O.sample(x.copy(), 1.0, 0.9, u)
    x[::3] = 0.5                                           # two probability levels interleaved over all blocks
    for u in (0.0, 0.3, 0.6, 0.99):
        assert _topp_dev(dev, x, 1.0, 0.9, u) == O.sample(x.copy(), 1.0, 0.9, u)


def _topp_scratch(dev, nmax):
    """(m, sorted probabilities, sorted indices, running sums) the sampler left on the device"""
    import ctypes as C
    from rama_amd._lib import check
    f = dev.lib.rama_internal_topp_scratch
    f.restype = None
    f.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 4
    keys, vals, prefix, m = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    f(dev.ctx, C.byref(keys), C.byref(vals), C.byref(prefix), C.byref(m))
    def down(ptr, n):
        out = np.empty(n, dtype=np.float32)
        check(dev.lib.rama_download_f32(dev.ctx, ptr, n, out.ctypes.data), "rama_download_f32")
        return out
    mm = int(down(m, 1).view(np.int32)[0])
    return mm, down(keys, nmax)[:mm], down(vals, nmax).view(np.int32)[:mm], down(prefix, nmax)


def _seq_cumsum_until(p, topp):
    """infer.rs:70-73 on the host: sequential fp32 sums; (sums up to and including the crossing, last)"""
    cum = np.cumsum(p, dtype=np.float32)                  # add.accumulate: one rounding per element, in order
    over = np.nonzero(cum > np.float32(topp))[0]
    last = int(over[0]) if over.size else len(p) - 1
    return cum[:last + 1], last


TOPP_SUM_CASES = [("flat", 32000, 0.05, 0.9), ("flat", 32000, 0.05, 1.0), ("ordinary", 32000, 3.0, 0.9), ("peaked", 32000, 12.0, 0.95),
                  ("equal", 32000, 0.0, 0.9), ("equal", 30011, 0.0, 0.99), ("two_levels", 32000, 0.0, 0.9), ("pow2", 32768, 0.0, 0.9),
                  ("flat", 2049, 0.1, 0.9), ("ordinary", 4097, 2.0, 0.999), ("flat", 100, 0.1, 0.5), ("steps", 32000, 0.0, 0.97),
                  ("masked", 32000, 1.0, 0.9), ("masked", 5000, 0.1, 0.95)]


def _dist_bad(dev):
    """diagnostic word of topp_pick_dist_kernel: bit 0 a hand-off wait timed out, bit 1 a predicted binade did not hold"""
    import ctypes as C
    bad = C.c_uint(0)
    f = dev.lib.rama_internal_topp_dist_bad
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
    assert f(dev.ctx, C.byref(bad)) == 0
    return bad.value


def test_topp_dist_error_word_fails_the_next_synchronising_call_and_is_cleared(dev):
    """[r5] the distributed pick's error word (a wait that gave up / a prediction that did not hold) is read at the synchronising exits like the
    in-launch hand-offs' word: the call fails, the word is cleared -- a hiccup neither leaves with rc 0 nor poisons the context"""
    import ctypes as C
    import rama_amd
    from rama_amd._lib import check
    x = (np.random.default_rng(5).standard_normal(32000) * 0.05).astype(np.float32)
    tl = rama_amd.MutView(dev.allocate(x))
    res = rama_amd.MutView(dev.allocate(np.zeros(4, np.float32)))
    check(dev.lib.rama_sample_topp_dev(dev.ctx, tl.ptr, x.size, 1.0, 0.9, 0.3, res.ptr))      # makes the scratch, runs the distributed pick
    dev.sync()
    assert _dist_bad(dev) == 0
    want = int(dev.download(res).view(np.int32)[0])
    poke = dev.lib.rama_internal_topp_dist_poke
    poke.restype = C.c_int
    poke.argtypes = [C.c_void_p, C.c_uint]
    for word in (1, 2):
        assert poke(dev.ctx, word) == 0
        assert dev.lib.rama_sync(dev.ctx) != 0, "a raised error word left rama_sync with rc 0"
        assert b"top-p sampler" in dev.lib.rama_last_error()
        assert dev.lib.rama_sync(dev.ctx) == 0 and _dist_bad(dev) == 0, "the word was not cleared"
        check(dev.lib.rama_sample_topp_dev(dev.ctx, tl.ptr, x.size, 1.0, 0.9, 0.3, res.ptr))
        assert int(dev.download(res).view(np.int32)[0]) == want


@pytest.mark.parametrize("dist", [1, 0])
@pytest.mark.parametrize("kind,n,scale,topp", TOPP_SUM_CASES)
def test_sample_topp_dev_running_sums_are_the_sequential_ones(dev, kind, n, scale, topp, dist):
    """the sampler's sorted order and every running sum up to the crossing, bit for bit against a
    sequential fp32 accumulation of the device's own sorted probabilities (csrc/topp_sort.hpp forms
    them with a parallel scan over integer increments): flat lists where cum walks through ~15
    binades, all-equal probabilities (every add in a binade rounds the same way; ties to even when
    the probability's low bits are 10..0), power-of-two probabilities, staircase lists"""
    if kind == "equal":
        x = np.full(n, 0.125, dtype=np.float32)
    elif kind == "two_levels":
        x = np.full(n, 0.0, dtype=np.float32); x[::3] = np.float32(np.log(2.0))
    elif kind == "pow2":
        x = np.zeros(n, dtype=np.float32)                  # p = 2^-15 exactly: every add is exact until the ties begin
    elif kind == "steps":
        x = (np.arange(n) // 1000).astype(np.float32) * np.float32(0.6931472)       # 32 plateaus, each twice the last
    else:
        x = rnd(n, 7 + n % 13, scale)
        if kind == "masked":                                  # [r5] whole 1 024-entry stretches of -inf logits (a masked vocabulary): no mass, no nan
            x[1024:3072] = -np.inf; x[n - 900:] = -np.inf
    from rama_amd._lib import check
    check(dev.lib.rama_set_tuning(dev.ctx, b"topp_keep_sums", 1))
    check(dev.lib.rama_set_tuning(dev.ctx, b"topp_dist", dist))       # the sums by 32 workgroups in one launch (default) | one workgroup's scan rounds
    try:
        _topp_dev(dev, x, 1.0, topp, 0.5)
    finally:
        check(dev.lib.rama_set_tuning(dev.ctx, b"topp_keep_sums", 0))
        check(dev.lib.rama_set_tuning(dev.ctx, b"topp_dist", 1))
    assert _dist_bad(dev) == 0
    m, ps, idx, prefix = _topp_scratch(dev, n)
    assert m > 0
    # order: descending probability, equal ones by ascending index (stable sort of the index-ordered list)
    assert np.all((ps[:-1] > ps[1:]) | ((ps[:-1] == ps[1:]) & (idx[:-1] < idx[1:])))
    assert len(set(idx.tolist())) == m
    want, last = _seq_cumsum_until(ps, topp)
    got = prefix[:last + 1]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (kind, int(np.nonzero(got != want)[0][0]), last, m)


@pytest.mark.parametrize("dist", [1, 0])
@pytest.mark.parametrize("seed", range(40))
def test_sample_topp_dev_running_sums_random_structures(dev, seed, dist):
    """the exact parallel running sum on randomly structured lists: mixtures of plateaus (equal
    probabilities: ties in every binade or in none), exact powers of two (adds that are exact until
    they tie), geometric tails and noise, random sizes up to the LDS path's 32768 and random topp"""
    from rama_amd._lib import check
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([3, 64, 65, 1000, 2048, 4097, 20000, 32000, 32768]))
    kind = seed % 4
    if kind == 0:      # plateaus of equal logits, heights on a log-2 grid
        levels = rng.integers(0, 12, size=int(rng.integers(1, 9)))
        x = (rng.choice(levels, size=n) * np.log(2.0)).astype(np.float32)
    elif kind == 1:    # gaussian noise of a random scale, a few exact duplicates
        x = (rng.standard_normal(n) * rng.choice([0.02, 0.3, 1.0, 2.5, 6.0])).astype(np.float32)
        x[rng.integers(0, n, size=max(1, n // 50))] = x[0]
    elif kind == 2:    # geometric tail
        x = (-np.arange(n) * rng.choice([1e-4, 1e-3, 0.01, 0.3])).astype(np.float32)
        rng.shuffle(x)
    else:              # two plateaus plus noise below the float resolution of the sums
        x = np.where(rng.random(n) < 0.1, 3.0, 0.0).astype(np.float32) + (rng.standard_normal(n) * 1e-6).astype(np.float32)
    topp = float(rng.choice([0.5, 0.9, 0.95, 0.999, 1.0]))
    check(dev.lib.rama_set_tuning(dev.ctx, b"topp_keep_sums", 1))
    check(dev.lib.rama_set_tuning(dev.ctx, b"topp_dist", dist))
    try:
        got_tok = _topp_dev(dev, x, 1.0, topp, 0.37)
    finally:
        check(dev.lib.rama_set_tuning(dev.ctx, b"topp_keep_sums", 0))
        check(dev.lib.rama_set_tuning(dev.ctx, b"topp_dist", 1))
    assert _dist_bad(dev) == 0
    m, ps, idx, prefix = _topp_scratch(dev, n)
    if m == 0:
        assert got_tok == -1
        return
    assert np.all((ps[:-1] > ps[1:]) | ((ps[:-1] == ps[1:]) & (idx[:-1] < idx[1:])))
    want, last = _seq_cumsum_until(ps, topp)
    got = prefix[:last + 1]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (seed, kind, n, topp, int(np.nonzero(got != want)[0][0]), last, m)
    # and the draw inside the prefix (infer.rs:75-84) from those sums
    r = np.float32(0.37) * want[last]
    below = int(np.count_nonzero(~(r < want[:last])))
    assert got_tok == int(idx[min(below, last)])


@pytest.mark.parametrize("n", [2049, 3000, 4097, 9000, 20000, 32000, 32768])
def test_sample_topp_dev_dist_pick_equals_scan_pick(dev, n):
    """csrc/topp_pick.hpp (the running sums by up to 32 workgroups in one launch: binades predicted from the exact fixed-point mass in
    front of every entry, integer maps, a lane ripple over the chunks' items) against topp_pick_scan_kernel (one workgroup, a scan
    round per binade): the same token for every draw -- u = 0, u next to 1, topp 1.0 (the whole list, up to the binade of 1.0), a
    tiny topp (the crossing in the first chunk), lists of equal probabilities (every add of a binade rounds the same way: the
    largest drift between the real-number mass and the fp32 sum), two plateaus, a steep geometric tail -- and no prediction that
    failed, no wait that timed out"""
    from rama_amd._lib import check
    rng = np.random.default_rng(7 * n)
    geo = (-np.arange(n) * 0.01).astype(np.float32); rng.shuffle(geo)
    lists = [rnd(n, n + 11, 0.05), rnd(n, n + 12, 1.0), rnd(n, n + 13, 3.0), rnd(n, n + 14, 8.0), np.full(n, 0.125, np.float32),
             np.where(rng.random(n) < 0.3, 2.0, 0.0).astype(np.float32), geo,
             (rng.integers(0, 3, n).astype(np.float32) * np.float32(0.6931472))]
    draws = [(1.0, 0.9, 0.2721174359321594), (1.0, 0.9, 0.0), (1.0, 0.9, 0.99999994), (1.0, 1.0, 0.5), (1.0, 1.0, 0.99999994), (1.0, 0.01, 0.6),
             (0.5, 0.95, 0.41), (1.0, 0.5, 0.999)]
    try:
        for li, x in enumerate(lists):
            for temperature, topp, u in draws:
                got = {}
                for dist in (1, 0):
                    check(dev.lib.rama_set_tuning(dev.ctx, b"topp_dist", dist))
                    got[dist] = _topp_dev(dev, x, temperature, topp, u)
                assert got[1] == got[0], (n, li, temperature, topp, u, got)
    finally:
        check(dev.lib.rama_set_tuning(dev.ctx, b"topp_dist", 1))
    assert _dist_bad(dev) == 0


def test_sample_topp_dev_temperature_zero_is_argmax(dev):
    x = rnd(5000, 9, 2.0)
    x[[10, 4000]] = 50.0
    assert _topp_dev(dev, x, 0.0, 0.9, 0.3) == 4000 == O.argmax(x)


def test_sample_topp_dev_without_candidates(dev):
    """uniform logits and topp = 0: every p = 1/n is below the cutoff 1/(n-1); the reference's index
    arithmetic underflows there (infer.rs:56-84), the oracle and the device sampler both say -1"""
    x = np.zeros(500, dtype=np.float32)
    assert _topp_dev(dev, x, 1.0, 0.0, 0.5) == -1 == O.sample(x.copy(), 1.0, 0.0, 0.5)
