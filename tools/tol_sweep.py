#!/usr/bin/env python3
"""Which op carries how much of the distance to the CPU path?  Full-depth shape (default llama2-7B, 27 GB of
synthetic weights on both sides), the generate() loop on 'once upon a time' with the ORACLE's greedy tokens;
the oracle's logits are computed once, then every mode / "tol_mask" runs the same positions on the GPU:

    fast            the fast path (fused multiply-adds, tree sums)
    parity          every op in the reference's rounding order (bit-identical)
    tol             tolerance mode ("ref_order" = 2): chain-order matvecs + tree-summed norms folded in + fast attention
    tol+<mask>      tolerance mode with ops swapped: 1 / 2 / 4 / 8 / 16 = the FAST qkv / wo / w13 / w2 / cls launch,
                    32 = parity mode's attention, 64 = parity mode's exact-sum norm launches

Prints one JSON line per configuration: worst |dlogit| vs the oracle, positions over 1e-4, greedy-token equality.
Usage: python tools/tol_sweep.py [n_positions] [shape] [mask,mask,...]      (test infrastructure: uses oracle/)"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import synth as S  # noqa: E402

SHAPES = {"llama2-7B": (4096, 11008, 32, 32, 32000, 2048, False),
          "stories110M": (768, 2048, 12, 12, 32000, 1024, True),
          "stories15M": (288, 768, 6, 6, 32000, 256, True)}
n_pos = int(sys.argv[1]) if len(sys.argv) > 1 else 200
shape = sys.argv[2] if len(sys.argv) > 2 else "llama2-7B"
masks = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8, 16, 32, 64, 5, 10, 15, 31]
d, h, L, H, V, seq, shared = SHAPES[shape]
n_pos = min(n_pos, seq)
cfg = O.Config(d, h, L, H, H, V, seq, shared)
rope = S.rope_tables(seq, d // H)
w = S.synth_weights(cfg, 0, rope=rope)
orc = O.Oracle(cfg, w, threads=16)
prompt = [10646, 2501, 263, 931]
token, fed, ref, toks = 1, [], [], []
t0 = time.time()
for pos in range(n_pos):
    lo = orc.forward(token, pos).copy()
    fed.append(int(token)); ref.append(lo); toks.append(int(O.argmax(lo)))
    token = prompt[pos] if pos < len(prompt) else toks[-1]
print(json.dumps({"shape": shape, "positions": n_pos, "oracle_s_per_token": round((time.time() - t0) / n_pos, 3)}), flush=True)
del orc, w

dev = rama_amd.Hip(0)
rcfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
model = rama_amd.Model.synth(dev, rcfg, 0, rope=rope)
configs = [("fast", 0, 0), ("parity", 1, 0), ("tol", 2, 0)] + [(f"tol+{m}", 2, m) for m in masks]
for name, ro, mask in configs:
    eng = rama_amd.Engine(dev, model)
    eng.set_tuning("ref_order", ro)
    eng.set_tuning("tol_mask", mask)
    worst, over, first, same = 0.0, 0, None, True
    per = []
    try:
        for pos in range(n_pos):
            eng.forward(fed[pos], pos)
            lg = eng.logits()
            dlt = float(np.abs(lg - ref[pos]).max())
            per.append(dlt)
            worst = max(worst, dlt)
            if dlt > 1e-4:
                over += 1
                first = pos if first is None else first
            same = same and int(np.flatnonzero(lg == lg.max())[-1]) == toks[pos]
    finally:
        eng.set_tuning("ref_order", 0)
        eng.set_tuning("tol_mask", 0)
        eng.free()
    print(json.dumps({"config": name, "worst_vs_oracle": worst, "positions_over_1e-4": over, "first_over": first,
                      "greedy_tokens_equal": same, "median": float(np.median(per)), "at_last": per[-1]}), flush=True)
model.free()
dev.close()
