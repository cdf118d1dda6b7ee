#!/usr/bin/env python3
"""How far do the reference's OWN admissible executions sit from each other?  (CPU only; test infrastructure: uses oracle/)

engine/src/device/cpu.rs leaves two summation orders to its crates (oracle/rama_oracle.h):
  * cpu.rs:148  `v.reduce_add()`            -- wide's final 4-lane sum: pairwise | strided | sequential (1 ulp apart per output)
  * cpu.rs:190  `x.par_iter().sum::<f32>()` -- rayon's halving tree: 2^levels leaves, levels = floor(log2(threads)) + 1 without
                                               steals (1 thread: 2 leaves; 16 threads: 32), more with steals
The oracle's default (pairwise lanes, ONE front-to-back softmax sum) is what parity mode reproduces bit for bit.  This script
runs the default and every variant over the same positions of the full-depth llama2-7B shape (the default's greedy tokens fed
to all of them, as tests/test_hip_parity_7b.py does) and records max |logit - default's logit| per position: the distance
between two executions the reference itself may produce.  "Within 1e-4 of the reference" cannot mean less than that.

    python tools/ref_self_spread.py [n_positions=200] [shape=llama2-7B]     -> profiles/r06_reference_self_spread.json
"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from oracle import oracle as O  # noqa: E402
from oracle import synth as S  # noqa: E402

SHAPES = {"llama2-7B": (4096, 11008, 32, 32, 32000, 2048, False),
          "stories110M": (768, 2048, 12, 12, 32000, 1024, True),
          "stories15M": (288, 768, 6, 6, 32000, 256, True)}
n_pos = int(sys.argv[1]) if len(sys.argv) > 1 else 200
shape = sys.argv[2] if len(sys.argv) > 2 else "llama2-7B"
d, h, L, H, V, seq, shared = SHAPES[shape]
n_pos = min(n_pos, seq)
cfg = O.Config(d, h, L, H, H, V, max(64, n_pos), shared)       # caches sized for the sample's positions
rope = S.rope_tables(cfg.seq_len, d // H)
t0 = time.time()
w = S.synth_weights(cfg, 0, rope=rope)
try:
    threads = min(16, len(os.sched_getaffinity(0)))
except AttributeError:
    threads = 8
VARIANTS = [("lanes_strided", dict(lane_reduce="strided")),
            ("lanes_sequential", dict(lane_reduce="sequential")),
            ("softmax_2_leaves", dict(softmax_split=1)),           # rayon, 1 thread
            ("softmax_4_leaves", dict(softmax_split=2)),
            ("softmax_16_leaves", dict(softmax_split=4)),
            ("softmax_32_leaves", dict(softmax_split=5)),          # rayon, 16 threads, no steals
            ("lanes_strided+softmax_32_leaves", dict(lane_reduce="strided", softmax_split=5))]
base = O.Oracle(cfg, w, threads=threads)
orcs = {name: O.Oracle(cfg, w, threads=threads) for name, _ in VARIANTS}
curves = {name: [] for name, _ in VARIANTS}
tok_equal = {name: True for name, _ in VARIANTS}
prompt = [10646, 2501, 263, 931]
out_path = Path(os.environ.get("RAMA_SPREAD_JSON", REPO / "profiles" / "r06_reference_self_spread.json"))


def dump(done):
    rows = []
    for name, kw in VARIANTS:
        per = np.asarray(curves[name])
        over = np.flatnonzero(per > 1e-4)
        rows.append({"variant": name, "orders": kw, "worst_vs_default": float(per.max()), "median": float(np.median(per)),
                     "positions_over_1e-4": int(over.size), "first_over": int(over[0]) if over.size else None,
                     "greedy_tokens_equal_default": tok_equal[name]})
    out_path.write_text(json.dumps({
        "what": "max |logit - logit of the default oracle| per position between admissible executions of engine/src/device/cpu.rs "
                "(lane order of wide::f32x4::reduce_add, cpu.rs:148; leaves of rayon's par_iter().sum(), cpu.rs:190); the default "
                "(pairwise lanes, one front-to-back softmax sum) is what parity mode reproduces bit for bit",
        "shape": f"{shape} fp32, all {L} layers, synthetic weights seed 0", "positions": done, "oracle_threads": threads,
        "prompt": "BOS + 'once upon a time' (Rama-BPE), the default oracle's greedy continuation fed to every variant",
        "command": "python tools/ref_self_spread.py " + " ".join(sys.argv[1:]),
        "summaries": rows, "per_position": curves}, indent=1))


print(json.dumps({"weights_s": round(time.time() - t0, 1), "threads": threads, "positions": n_pos}), flush=True)
token, t0 = 1, time.time()
for pos in range(n_pos):
    lo = base.forward(token, pos).copy()
    nxt = int(O.argmax(lo))
    for name, kw in VARIANTS:
        with O.orders(**kw):
            lv = orcs[name].forward(token, pos)
        curves[name].append(float(np.abs(lv - lo).max()))
        tok_equal[name] = tok_equal[name] and int(O.argmax(lv)) == nxt
    token = prompt[pos] if pos < len(prompt) else nxt
    if (pos + 1) % 10 == 0 or pos + 1 == n_pos:
        print(json.dumps({"progress": pos + 1, "elapsed_s": round(time.time() - t0, 1),
                          "worst_so_far": {n: max(c) for n, c in curves.items()}}), flush=True)
        dump(pos + 1)
