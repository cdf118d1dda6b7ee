"""CPU: the sampler's constant uniform draw derived from the published algorithms (rand_core's PCG32 seed expansion, ChaCha20
block 0, rand's f32 conversion) and the hand-worked edge cases of sample_top_q (engine/src/transformer/infer.rs:55-85)."""
import re
from pathlib import Path

import numpy as np

from oracle import oracle as O
from rama_amd import sampler_const as SC

REPO = Path(__file__).resolve().parents[1]


def test_chacha_core_against_rfc7539_zero_key_block():
    # RFC 7539 section 2.3.2-style known answer: all-zero key, nonce and counter -> keystream 76 b8 e0 ad a0 f1 3d 90 ...
    blk = SC.chacha20_block(bytes(32))
    assert blk[0] == 0xADE0B876 and blk[1] == 0x903DF1A0 and blk[2] == 0xE56A5D40 and blk[3] == 0x28BD8653


def test_pcg32_expansion_is_the_documented_stream():
    # rand_core's doc example: seed_from_u64(0) of a 32-byte seed starts with PCG32's first outputs for state 0 advanced once
    s = SC.pcg32_seed_bytes(0)
    assert len(s) == 32 and s != bytes(32)
    # the stream is a permutation-congruential generator: consecutive seeds give unrelated keys
    assert SC.pcg32_seed_bytes(1)[:4] != s[:4]


def test_the_two_draws_of_the_reference():
    assert SC.first_word(100) == 0x45A97D74
    assert SC.first_word(10) == 0x0995660C
    assert SC.first_f32(100) == SC.TOPP_U_CPU == 0.2721174359321594       # cpu.rs:161-162
    assert SC.first_f32(10) == SC.TOPP_U_CUDA == 0.03743588924407959      # gpu.rs:151-152
    assert np.float32(SC.TOPP_U_CPU) == SC.TOPP_U_CPU                     # 24 significant bits: exact in fp32


def test_every_host_mirror_uses_the_derived_literal():
    hpp = (REPO / "rama_amd" / "csrc" / "host" / "engine.hpp").read_text()
    m = re.search(r"RAMA_TOPP_U.*?:\s*([0-9.]+)f;", hpp)
    assert m and float(m.group(1)) == SC.TOPP_U_CPU
    import bench
    assert bench.TOPP_U == SC.TOPP_U_CPU
    import inspect
    import rama_amd.transformer as T
    assert inspect.signature(T.Hip.sample).parameters["u"].default == SC.TOPP_U_CPU
    assert inspect.signature(T.generate_device).parameters["u"].default == SC.TOPP_U_CPU


# ---- sample_top_q, worked by hand from infer.rs:55-85 (cutoff = (1 - topp) / (n - 1); keep p > cutoff; stable sort descending;
# cum += p until cum > topp -> last_index; r = u * cum; walk cdf over [0, last_index): first i with r < cdf; else last_index)

def _topq(p, topp, u):
    return O.sample_top_q(np.asarray(p, np.float32), len(p), np.float32(topp), np.float32(u))


def test_sample_top_q_all_equal_probabilities():
    # n = 4, p = 0.25 each, topp = 0.9: cutoff = 0.1 / 3 = 0.0333 -> all kept, order = index order (stable sort).
    # cum: 0.25, 0.5, 0.75, 1.0 > 0.9 at i = 3 -> last_index = 3, cum = 1.0.  r = u: cdf walk over i < 3: 0.25, 0.5, 0.75
    assert _topq([0.25] * 4, 0.9, 0.0) == 0
    assert _topq([0.25] * 4, 0.9, 0.2721174359321594) == 1      # 0.25 <= r < 0.5
    assert _topq([0.25] * 4, 0.9, 0.6) == 2
    assert _topq([0.25] * 4, 0.9, 0.8) == 3                     # never r < cdf inside the walk: falls through to last_index


def test_sample_top_q_cum_never_exceeds_topp():
    # topp = 1.0: cutoff = 0 -> every p > 0 kept; cum reaches exactly 1.0, never > 1.0 -> last_index stays len - 1, cum = total.
    # p = [0.5, 0.25, 0.25]: r = u * 1.0; walk i < 2: cdf 0.5, 0.75
    assert _topq([0.5, 0.25, 0.25], 1.0, 0.49) == 0
    assert _topq([0.5, 0.25, 0.25], 1.0, 0.5) == 1               # r == cdf edge: `r < cdf` is false at i = 0
    assert _topq([0.5, 0.25, 0.25], 1.0, 0.74) == 1
    assert _topq([0.5, 0.25, 0.25], 1.0, 0.75) == 2              # edge again: falls through to last_index = 2
    # a zero probability is not > cutoff 0: dropped, the rest as before
    assert _topq([0.5, 0.0, 0.25, 0.25], 1.0, 0.6) == 2


def test_sample_top_q_single_candidate():
    # n = 3, topp = 0.5: cutoff = 0.25; p = [0.1, 0.8, 0.1] -> only index 1 kept; cum = 0.8 > 0.5 at i = 0 -> last_index 0:
    # the walk is empty, whatever u
    for u in (0.0, 0.2721174359321594, 0.999):
        assert _topq([0.1, 0.8, 0.1], 0.5, u) == 1


def test_sample_top_q_stable_order_among_ties_and_descending_sort():
    # p = [0.2, 0.3, 0.2, 0.3], topp = 0.9: cutoff 0.0333; sorted (stable, descending): idx 1 (0.3), 3 (0.3), 0 (0.2), 2 (0.2)
    # cum: 0.3, 0.6, 0.8, 1.0 > 0.9 -> last_index 3, cum = 1.0 (fp32: 0.3f + 0.3f + 0.2f + 0.2f = 1.0000001 > 0.9 too)
    p = [0.2, 0.3, 0.2, 0.3]
    assert _topq(p, 0.9, 0.1) == 1
    assert _topq(p, 0.9, 0.45) == 3
    assert _topq(p, 0.9, 0.7) == 0
    assert _topq(p, 0.9, 0.95) == 2


def test_sample_top_q_r_scales_with_the_truncated_mass():
    # p = [0.6, 0.3, 0.1], topp = 0.5: cutoff 0.25 -> kept 0.6, 0.3; cum = 0.6 > 0.5 at i = 0 -> last_index 0 -> always index 0
    assert _topq([0.6, 0.3, 0.1], 0.5, 0.99) == 0
    # topp = 0.8: cutoff 0.1 -> kept 0.6, 0.3 (0.1 is not > 0.1f? 0.1f > (1 - 0.8f) / 2 = 0.099999994: kept); cum 0.6, 0.9 > 0.8 -> last 1, cum 0.9
    # r = u * 0.9; walk i < 1: cdf 0.6 -> u < 2/3 gives 0, else 1
    assert _topq([0.6, 0.3, 0.1], 0.8, 0.66) == 0
    assert _topq([0.6, 0.3, 0.1], 0.8, 0.67) == 1
