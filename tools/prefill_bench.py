#!/usr/bin/env python3
"""Prompt processing at the llama2-7B shape: rama_prefill (weights streamed once per 8 positions)
vs the same positions decoded one by one.  Prints one JSON line."""
import ctypes as C
import json
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import rama_amd
from rama_amd._lib import check
from bench import SHAPES

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rounds = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1]
d, h, L, H, V, seq, shared = SHAPES["llama2-7B"]
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
model = rama_amd.Model.synth(dev, cfg, seed=0)
a, b = rama_amd.Engine(dev, model), rama_amd.Engine(dev, model)
toks = [1] + [int(v) for v in np.random.default_rng(0).integers(2, V, n - 1)]
arr = (C.c_int32 * n)(*toks)
def prefill():
    check(dev.lib.rama_prefill(dev.ctx, C.byref(model.ccfg), C.byref(model.weights), C.byref(a.state), arr, n, 0))
sweep = {}
for r in rounds:
    check(dev.lib.rama_set_tuning(dev.ctx, b"prefill_rounds", r))
    prefill(); dev.sync()
    t0 = time.perf_counter(); prefill(); dev.sync(); sweep[r] = time.perf_counter() - t0
best = min(sweep, key=sweep.get)
check(dev.lib.rama_set_tuning(dev.ctx, b"prefill_rounds", best))
prefill(); dev.sync()
t_pf = sweep[best]
for i, t in enumerate(toks): b.forward(t, i)
dev.sync()
t0 = time.perf_counter()
for i, t in enumerate(toks): b.forward(t, i)
dev.sync(); t_seq = time.perf_counter() - t0
diff = float(np.abs(a.logits() - b.logits()).max())
print(json.dumps({"positions": n, "prefill_rounds": best, "sweep_ms": {str(k): round(v * 1e3, 2) for k, v in sweep.items()}, "prefill_ms": round(t_pf * 1e3, 2), "sequential_ms": round(t_seq * 1e3, 2),
                  "prefill_tok_s": round(n / t_pf, 1), "sequential_tok_s": round(n / t_seq, 1), "speedup": round(t_seq / t_pf, 2),
                  "max_abs_logit_diff_last_position": diff}))
