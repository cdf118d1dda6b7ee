"""CPU: the bound behind csrc/topp_pick.hpp's prediction, checked against the thing it bounds.

The distributed top-p pick tells in advance WHICH binade the sequential fp32 sum s_{i-1} (infer.rs:70-73) is in when element i
is added, from the exact fixed-point mass A_i in front of i (unit 2^-47, truncated: topp_sort.hpp topp_fixed):

    eps_i = (i + 128) 2^-23
    SAFE for binade E  <=>  2^E <= A_i (1 - eps_i) - i 2^-47   and   (A_i (1 + eps_i) + i 2^-47 + p_i)(1 + 2^-20) < 2^(E+1)

and claims: SAFE => s_{i-1} is in binade E and s_i = fl(s_{i-1} + p_i) still is (no fallback exists in the kernel).  Here the
same classification is formed with numpy in float32 (the kernel's arithmetic) and held against numpy's sequential float32
running sum on the lists the sampler meets and on adversarial ones: equal probabilities (every add of a binade rounds the same
way -- the largest drift between the real-number mass and the fp32 sum), powers of two, staircases, a spike, descending
geometric lists.  It also counts the elements that are NOT safe in front of the crossing of topp = 0.9 (the sequential part of the walk): a few
hundred at most."""
import zlib

import numpy as np
import pytest


def _classify(p):
    """(safe, E) per element, in the kernel's float32 arithmetic; p sorted descending, float32"""
    n = p.size
    unit = 2.0 ** -47
    fixed = np.floor(p.astype(np.float64) / unit).astype(np.int64)            # topp_fixed: truncation to 2^-47 (p <= 1: below 2^48)
    mass = np.concatenate(([0], np.cumsum(fixed)[:-1]))                          # exact integer sums, any order
    f32 = np.float32
    av = (mass.astype(np.float64)).astype(f32) * f32(unit)                       # (float)mass * 2^-47: one rounding
    i = np.arange(n, dtype=np.float32)
    eps = (i + f32(128)) * f32(2.0 ** -23)
    absm = i * f32(unit)
    lo = av * (f32(1) - eps) - absm
    hi = (av * (f32(1) + eps) + absm + p) * f32(1 + 2.0 ** -20)
    el = (lo.view(np.uint32) >> 23).astype(np.int64)
    eh = (hi.view(np.uint32) >> 23).astype(np.int64)
    safe = (lo > 0) & (el == eh) & (el >= 24) & (el <= 253)
    return safe, el


LISTS = {
    "flat": lambda rng, n: np.exp(rng.standard_normal(n) * 0.05),
    "ordinary": lambda rng, n: np.exp(rng.standard_normal(n) * 3.0),
    "peaked": lambda rng, n: np.exp(rng.standard_normal(n) * 9.0),
    "equal": lambda rng, n: np.ones(n),
    "two_levels": lambda rng, n: np.where(np.arange(n) % 3 == 0, 2.0, 1.0),
    "pow2": lambda rng, n: 2.0 ** -rng.integers(0, 12, n),
    "steps": lambda rng, n: 2.0 ** (np.arange(n) // 1000),
    "geometric": lambda rng, n: np.exp(-np.arange(n) * 0.01),
    "spike": lambda rng, n: np.concatenate(([1e6], np.ones(n - 1))),
}


@pytest.mark.parametrize("kind", sorted(LISTS))
@pytest.mark.parametrize("n", [2049, 8193, 32000, 32768])
def test_safe_elements_are_in_their_binade(kind, n):
    rng = np.random.default_rng(zlib.crc32(f"{kind}-{n}".encode()))
    w = LISTS[kind](rng, n).astype(np.float64)
    p = np.sort((w / w.sum()).astype(np.float32))[::-1].copy()                   # the sampler's order: descending
    s = np.cumsum(p, dtype=np.float32)                                           # one rounding per add, in order
    before = np.concatenate(([np.float32(0)], s[:-1]))
    safe, E = _classify(p)
    eb = (before.view(np.uint32) >> 23).astype(np.int64)
    ea = (s.view(np.uint32) >> 23).astype(np.int64)
    wrong = safe & ((eb != E) | (ea != E))
    assert not wrong.any(), (kind, n, int(np.flatnonzero(wrong)[0]))
    assert not safe[0]                                                           # the first element has no sum in front of it
    # what is left to the sequential walk in front of the crossing of topp = 0.9: the elements around the powers of two the sum
    # passes.  (Behind it the sum approaches 1.0 -- a power of two: with topp = 1 a peaked list is walked element by element.)
    front = before.astype(np.float64) < 0.95
    assert int((~safe & front).sum()) <= 400, (kind, n, int((~safe & front).sum()))


def test_margin_covers_the_worst_drift():
    """equal probabilities: the rounding error of every add in a binade has the same sign, the fp32 sum drifts from the real
    sum linearly -- the case the (i - 1) 2^-24 term of the bound is for; the drift stays inside eps_i and outside eps_i / 64"""
    n = 32768
    p = np.full(n, np.float32(1.0 / 32000.0))
    s = np.cumsum(p, dtype=np.float32).astype(np.float64)
    exact = np.cumsum(p.astype(np.float64))
    rel = np.abs(s - exact) / exact
    i = np.arange(n)
    eps = (i + 128) * 2.0 ** -23
    assert np.all(rel <= eps / 2)
    assert rel.max() > eps[-1] / 64
