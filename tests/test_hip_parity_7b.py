"""FULL-depth llama2-7B parity (BASELINE.json config 4): 32 layers, 27 GB of synthetic fp32 weights
generated bit-identically on both sides, the generate() loop of transformer/mod.rs:169-206 on
'once upon a time' for 200 positions (the README bench length, README.md:80-83).

Per position five logit vectors are compared:
    HIP      the product's fast path (rama_forward through the C ABI)
    HIP-tol  the same entry in the TOLERANCE-mode experiment (rama_set_tuning "ref_order" = 2): the chain-order
             matvecs of parity mode -- the reference's own rounding sequence -- with the rmsnorm sums tree-shaped
             and folded into them and the fast path's attention.  Round 3's review expected ~1e-5 from it; it
             measures 1.4e-4: the sum of squares in another order already moves v = 1/sqrt(ss/n + eps) by a few
             ulps, 65 times per token, and that is enough (tools/tol_sweep.py: exact norms + fast attention 8.2e-5,
             tree norms + exact attention 1.4e-4)
    HIP-ref  the same entry in reference-order mode (rama_set_tuning "ref_order" = 1,
             csrc/ref_order.hpp): every sum in the reference's own order, glibc's expf restated
    oracle   oracle/rama_oracle.c, the line-by-line restatement of engine/src/device/cpu.rs
             (4-lane strided sums of 4096..11008 terms, sequential softmax / rmsnorm sums)
    f64      the same network with every sum accumulated in double (oracle_forward_f64): the arbiter
             that says how far each fp32 path sits from the exact result
All four are fed the SAME token sequence (the oracle's greedy choice), so caches stay comparable.

The per-position numbers are written to gpurun_out/r06_parity_llama2_7b_200pos.json (copied to
profiles/ by the builder) whatever the outcome; the assertions come last.

What is asserted, and why not simply "fast path within 1e-4 of the oracle": the reference arithmetic
itself sits up to ~1.5e-4 from the exact logits at this depth (4-lane sequential sums of 4096..11008
terms, 32 layers), the fast path ~2e-5; their difference is therefore the reference's own rounding
error and crosses 1e-4 from about position 50 on.  So:
  * HIP-ref vs oracle <= 1e-4 at every position -- the north_star bar, met by reproducing the
    reference's rounding (expected: identical bits);
  * HIP-tol vs oracle is RECORDED per position (it crosses 1e-4 like the fast path; asserted <= 2e-4, tokens identical):
    the evidence that only the bit-exact mode meets the bar at this depth;
  * HIP (fast) vs f64 <= 1e-4 and never worse than oracle vs f64; greedy tokens identical to the
    oracle's at every position;
  * HIP (fast) vs oracle is recorded per position, positions over 1e-4 are listed in the JSON, and it
    must stay below oracle-vs-f64 + HIP-vs-f64 (it is explained by those two, nothing else).

Needs ~35 GB of host memory and ~3 minutes (CPU oracle 0.17 s/token, fp64 arbiter ~0.25 s/token).
RAMA_PARITY_POSITIONS overrides the length (e.g. 8 for a quick run)."""
from __future__ import annotations

import json
import os
import time
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O
from oracle import synth as S

from .helpers import LOGIT_ATOL, to_rama_cfg

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parent.parent
PROMPT = [10646, 2501, 263, 931]          # Rama-BPE of 'once upon a time' 


@pytest.fixture(scope="module")
def dev():
    import rama_amd
    d = rama_amd.Hip(0)
    yield d
    d.close()


def _library_stamp():
    import sys
    sys.path.insert(0, str(REPO))
    from bench import library_stamp
    return library_stamp()


def _host_mem_gb() -> float:
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def test_llama2_7b_full_depth_200_positions(dev):
    import rama_amd
    n_pos = int(os.environ.get("RAMA_PARITY_POSITIONS", "200"))
    d, h, L, H, V, seq = 4096, 11008, 32, 32, 32000, 2048
    if _host_mem_gb() < 40.0:
        pytest.skip(f"full-depth 7B oracle needs ~33 GB of host memory, {_host_mem_gb():.0f} GB available")
    cfg = O.Config(d, h, L, H, H, V, seq, False)
    t0 = time.time()
    rope = S.rope_tables(seq, d // H)
    w = S.synth_weights(cfg, 0, rope=rope)
    t_gen = time.time() - t0
    threads = min(16, len(os.sched_getaffinity(0)))
    orc = O.Oracle(cfg, w, threads=threads)
    orc64 = O.Oracle(cfg, w, threads=threads)
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), 0, rope=rope)
    eng = rama_amd.Engine(dev, model)
    eng_ref = rama_amd.Engine(dev, model)
    eng_tol = rama_amd.Engine(dev, model)
    eng_bar = rama_amd.Engine(dev, model)      # [r6] "ref_order" = 3: parity mode up to position 127, the fast path's attention from 128 on

    rows, toks_cpu, toks_hip, toks_tol, toks_bar = [], [], [], [], []
    token = 1
    t_cpu = t_f64 = 0.0
    for pos in range(n_pos):
        t1 = time.time()
        lo = orc.forward(token, pos).copy()
        t2 = time.time()
        l64 = orc64.forward_f64(token, pos).copy()
        t3 = time.time()
        t_cpu += t2 - t1
        t_f64 += t3 - t2
        eng.forward(token, pos)
        lg = eng.logits()
        eng_ref.set_tuning("ref_order", 1)
        try:
            eng_ref.forward(token, pos)
            lr = eng_ref.logits()
        finally:
            eng_ref.set_tuning("ref_order", 0)
        eng_tol.set_tuning("ref_order", 2)
        try:
            eng_tol.forward(token, pos)
            lt = eng_tol.logits()
        finally:
            eng_tol.set_tuning("ref_order", 0)
        eng_bar.set_tuning("ref_order", 3)
        try:
            eng_bar.forward(token, pos)
            lb = eng_bar.logits()
        finally:
            eng_bar.set_tuning("ref_order", 0)
        toks_bar.append(int(np.flatnonzero(lb == lb.max())[-1]))
        rows.append({"pos": pos, "token": int(token),
                     "hip_bar_vs_oracle": float(np.abs(lb - lo).max()),
                     "hip_bar_bits_equal": bool(np.array_equal(lb.view(np.uint32), lo.view(np.uint32))),
                     "hip_tolerance_vs_oracle": float(np.abs(lt - lo).max()),
                     "hip_tolerance_vs_f64": float(np.abs(lt - l64).max()),
                     "hip_ref_order_vs_oracle": float(np.abs(lr - lo).max()),
                     "hip_ref_order_bits_equal": bool(np.array_equal(lr.view(np.uint32), lo.view(np.uint32))),
                     "hip_vs_oracle": float(np.abs(lg - lo).max()),
                     "hip_vs_f64": float(np.abs(lg - l64).max()),
                     "oracle_vs_f64": float(np.abs(lo - l64).max())})
        toks_cpu.append(int(O.argmax(lo)))
        toks_hip.append(int(np.flatnonzero(lg == lg.max())[-1]))
        toks_tol.append(int(np.flatnonzero(lt == lt.max())[-1]))
        token = PROMPT[pos] if pos < len(PROMPT) else toks_cpu[-1]

    # [r6] the deep tail: both sides jump to a context of 1 900 positions -- the cache rows n_pos..1 899 of every layer are filled with the same random
    # rows on both sides (rows 0..n_pos-1 stay what the generation wrote) -- and four more positions are compared at FULL depth: parity mode
    # bit for bit (the spread attention's slice geometry, the ~1 900-term exact softmax sum, at 32 layers x 32 heads x 128), the fast path recorded
    tail = []
    tail_at = int(os.environ.get("RAMA_PARITY_TAIL_AT", "1900"))
    if tail_at > n_pos and tail_at + 4 <= seq:
        rng = np.random.default_rng(5)
        for name in ("key_cache", "value_cache"):
            full = orc.s[name].reshape(L, seq, d)
            full[:, n_pos:tail_at, :] = rng.standard_normal((L, tail_at - n_pos, d), dtype=np.float32)
            for e in (eng, eng_ref, eng_bar):
                e.set_buffer(name, full)
        for pos in range(tail_at, tail_at + 4):
            t1 = time.time()
            lo = orc.forward(token, pos).copy()
            t_or = time.time() - t1
            eng.forward(token, pos)
            lg = eng.logits()
            eng_ref.set_tuning("ref_order", 1)
            try:
                eng_ref.forward(token, pos)
                lr = eng_ref.logits()
            finally:
                eng_ref.set_tuning("ref_order", 0)
            eng_bar.set_tuning("ref_order", 3)
            try:
                eng_bar.forward(token, pos)
                lb = eng_bar.logits()
            finally:
                eng_bar.set_tuning("ref_order", 0)
            tail.append({"pos": pos, "token": int(token), "oracle_s": round(t_or, 3),
                         "hip_bar_vs_oracle": float(np.abs(lb - lo).max()),
                         "greedy_token_equal_bar": int(np.flatnonzero(lb == lb.max())[-1]) == int(O.argmax(lo)),
                         "hip_ref_order_vs_oracle": float(np.abs(lr - lo).max()),
                         "hip_ref_order_bits_equal": bool(np.array_equal(lr.view(np.uint32), lo.view(np.uint32))),
                         "hip_vs_oracle": float(np.abs(lg - lo).max()),
                         "greedy_token_equal_ref_order": int(np.flatnonzero(lr == lr.max())[-1]) == int(O.argmax(lo))})
            token = int(O.argmax(lo))

    worst = max(r["hip_vs_oracle"] for r in rows)
    out = {
        "shape": "llama2-7B fp32, 32 layers, synthetic weights seed 0 (bit-identical on both sides)",
        "prompt": "BOS + 'once upon a time' (Rama-BPE), greedy continuation chosen by the oracle",
        "positions": n_pos, "bar": LOGIT_ATOL,
        "worst_hip_ref_order_vs_oracle": max(r["hip_ref_order_vs_oracle"] for r in rows),
        "positions_ref_order_bit_identical": sum(r["hip_ref_order_bits_equal"] for r in rows),
        "worst_hip_bar_vs_oracle": max([r["hip_bar_vs_oracle"] for r in rows] + [t["hip_bar_vs_oracle"] for t in tail]),
        "positions_bar_bit_identical": sum(r["hip_bar_bits_equal"] for r in rows),
        "greedy_tokens_equal_bar": toks_cpu == toks_bar,
        "worst_hip_tolerance_vs_oracle": max(r["hip_tolerance_vs_oracle"] for r in rows),
        "positions_tolerance_over_bar": [r["pos"] for r in rows if r["hip_tolerance_vs_oracle"] > LOGIT_ATOL],
        "greedy_tokens_equal_tolerance": toks_cpu == toks_tol,
        "worst_hip_vs_oracle": worst,
        "worst_hip_vs_f64": max(r["hip_vs_f64"] for r in rows),
        "worst_oracle_vs_f64": max(r["oracle_vs_f64"] for r in rows),
        "positions_over_bar": [r["pos"] for r in rows if r["hip_vs_oracle"] > LOGIT_ATOL],
        "greedy_tokens_equal": toks_cpu == toks_hip,
        "first_token_mismatch": next((i for i, (a, b) in enumerate(zip(toks_cpu, toks_hip)) if a != b), None),
        "weights_gen_s": round(t_gen, 1), "oracle_s_per_token": round(t_cpu / n_pos, 3),
        "f64_s_per_token": round(t_f64 / n_pos, 3), "oracle_threads": threads,
        "library": _library_stamp(),
        "deep_tail": tail,
        "per_position": rows,
    }
    path = Path(os.environ.get("RAMA_PARITY_JSON", REPO / "gpurun_out" / "r06_parity_llama2_7b_200pos.json"))
    try:
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(json.dumps(out, indent=1))
    except OSError:
        pass
    print(json.dumps({k: v for k, v in out.items() if k != "per_position"}))
    eng.free()

    eng_ref.free(); eng_tol.free(); eng_bar.free(); model.free()
    # the north_star bar, literally: logits within 1e-4 of the CPU reference path at every position
    assert out["worst_hip_ref_order_vs_oracle"] <= LOGIT_ATOL, out["worst_hip_ref_order_vs_oracle"]
    assert all(r["hip_ref_order_bits_equal"] for r in rows), [r["pos"] for r in rows if not r["hip_ref_order_bits_equal"]][:8]
    # ... and deep into the context (positions 1 900..1 903 over a filled cache, all 32 layers): the same bits
    assert all(t["hip_ref_order_bits_equal"] and t["greedy_token_equal_ref_order"] for t in tail), tail
    # [r6] bar mode: parity mode's bits below its switch (position 128), <= 1e-4 from the oracle behind it -- here over 72 + 4 positions; over the whole
    # 2 048-position context: profiles/r06_tolerance_sweep_7b_2048pos_bar.jsonl (tools/tol_sweep.py) -- and the oracle's tokens
    assert all(r["hip_bar_bits_equal"] for r in rows if r["pos"] < 128), [r["pos"] for r in rows if r["pos"] < 128 and not r["hip_bar_bits_equal"]][:8]
    assert out["worst_hip_bar_vs_oracle"] <= LOGIT_ATOL, out["worst_hip_bar_vs_oracle"]
    assert toks_cpu == toks_bar and all(t["greedy_token_equal_bar"] for t in tail)
    # the tolerance-mode experiment: same tokens, the same ~1.4e-4 from the oracle as the fast path (recorded above)
    assert out["worst_hip_tolerance_vs_oracle"] <= 2 * LOGIT_ATOL, out["worst_hip_tolerance_vs_oracle"]
    assert toks_cpu == toks_tol
    # the fast path: within 1e-4 of the exact logits, never further from them than the reference is
    assert out["worst_hip_vs_f64"] <= LOGIT_ATOL, out["worst_hip_vs_f64"]
    assert all(r["hip_vs_f64"] <= r["oracle_vs_f64"] for r in rows)
    assert toks_cpu == toks_hip, (out["first_token_mismatch"], toks_cpu[:16], toks_hip[:16])
    # and what separates it from the oracle is those two distances, nothing else
    assert all(r["hip_vs_oracle"] <= r["oracle_vs_f64"] + r["hip_vs_f64"] + 1e-7 for r in rows)
