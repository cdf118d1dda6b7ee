#!/usr/bin/env python3
"""Which op carries how much of the distance to the CPU path?  Full-depth shape (default llama2-7B, 27 GB of
synthetic weights on both sides), the generate() loop on 'once upon a time' with the ORACLE's greedy tokens;
every mode / "tol_mask" runs the same positions on the GPU, the oracle beside it:

    fast            the fast path (fused multiply-adds, tree sums)
    parity          every op in the reference's rounding order (bit-identical)
    tol             tolerance mode ("ref_order" = 2): chain-order matvecs + tree-summed norms folded in + fast attention
    bar             bar mode ("ref_order" = 3): parity mode up to position 127, the fast attention from 128 on (exact matvecs and norms)
    tol+<mask>      tolerance mode with ops swapped: 1 / 2 / 4 / 8 / 16 = the FAST qkv / wo / w13 / w2 / cls launch,
                    32 = parity mode's attention, 64 = parity mode's exact-sum norm launches

Prints one JSON line per configuration: worst |dlogit| vs the oracle, positions over 1e-4, greedy-token equality.
[r6] The oracle and every configuration advance TOGETHER position by position (one engine per configuration), a progress line
every 64 positions, and the per-position curves are written to RAMA_TOL_JSON (default gpurun_out/r06_tol_curve.json) as they
grow -- a run over the whole 2 048-position context is ~8 minutes of oracle and leaves its curve even when cut short.
Usage: python tools/tol_sweep.py [n_positions] [shape] [mask,mask,...] [modes]     (test infrastructure: uses oracle/)
       modes: comma list of fast,parity,tol,bar (default fast,parity,tol) run beside the tol+<mask> configurations"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import synth as S  # noqa: E402

from bench import library_stamp  # noqa: E402
LIBRARY = library_stamp()
SHAPES = {"llama2-7B": (4096, 11008, 32, 32, 32000, 2048, False),
          "stories110M": (768, 2048, 12, 12, 32000, 1024, True),
          "stories15M": (288, 768, 6, 6, 32000, 256, True)}
n_pos = int(sys.argv[1]) if len(sys.argv) > 1 else 200
shape = sys.argv[2] if len(sys.argv) > 2 else "llama2-7B"
masks = [int(v) for v in sys.argv[3].split(",") if v != ""] if len(sys.argv) > 3 else [1, 2, 4, 8, 16, 32, 64, 5, 10, 15, 31]
base = sys.argv[4].split(",") if len(sys.argv) > 4 else ["fast", "parity", "tol"]
d, h, L, H, V, seq, shared = SHAPES[shape]
n_pos = min(n_pos, seq)
cfg = O.Config(d, h, L, H, H, V, seq, shared)
rope = S.rope_tables(seq, d // H)
w = S.synth_weights(cfg, 0, rope=rope)
try:
    threads = min(16, len(os.sched_getaffinity(0)))
except AttributeError:
    threads = 16
orc = O.Oracle(cfg, w, threads=threads)
prompt = [10646, 2501, 263, 931]

dev = rama_amd.Hip(0)
rcfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
model = rama_amd.Model.synth(dev, rcfg, 0, rope=rope)
configs = [(m, {"fast": 0, "parity": 1, "tol": 2, "bar": 3}[m], 0) for m in base] + [(f"tol+{m}", 2, m) for m in masks]
engines = {name: rama_amd.Engine(dev, model) for name, _, _ in configs}
curves = {name: [] for name, _, _ in configs}
same = {name: True for name, _, _ in configs}
out_path = Path(os.environ.get("RAMA_TOL_JSON", Path(__file__).resolve().parent.parent / "gpurun_out" / "r06_tol_curve.json"))
out_path.parent.mkdir(parents=True, exist_ok=True)


def summary(name):
    per = np.asarray(curves[name])
    over = np.flatnonzero(per > 1e-4)
    return {"config": name, "positions": int(per.size), "worst_vs_oracle": float(per.max()), "positions_over_1e-4": int(over.size),
            "first_over": int(over[0]) if over.size else None, "greedy_tokens_equal": same[name],
            "median": float(np.median(per)), "at_last": float(per[-1]),
            "worst_by_512": [float(per[i:i + 512].max()) for i in range(0, per.size, 512)]}


def dump():
    out_path.write_text(json.dumps({"library": LIBRARY, "shape": shape, "positions_done": len(next(iter(curves.values()))), "oracle_threads": threads,
                                    "what": "max |logit - oracle logit| per position, the oracle's greedy tokens fed to every configuration",
                                    "summaries": [summary(n) for n in curves], "per_position": curves}))


token, t0 = 1, time.time()
try:
    for pos in range(n_pos):
        lo = orc.forward(token, pos)
        tok = int(O.argmax(lo))
        for name, ro, mask in configs:
            eng = engines[name]
            eng.set_tuning("ref_order", ro)
            eng.set_tuning("tol_mask", mask)
            eng.forward(token, pos)
            lg = eng.logits()
            curves[name].append(float(np.abs(lg - lo).max()))
            same[name] = same[name] and int(np.flatnonzero(lg == lg.max())[-1]) == tok
        token = prompt[pos] if pos < len(prompt) else tok
        if (pos + 1) % 64 == 0 or pos + 1 == n_pos:
            print(json.dumps({"progress": pos + 1, "elapsed_s": round(time.time() - t0, 1),
                              "worst_so_far": {n: max(c) for n, c in curves.items()}}), flush=True)
            dump()
finally:
    for eng in engines.values():
        eng.set_tuning("ref_order", 0)
        eng.set_tuning("tol_mask", 0)
        eng.free()
for name in curves:
    print(json.dumps(summary(name)), flush=True)
model.free()
dev.close()
