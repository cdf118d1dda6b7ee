#!/usr/bin/env python3
"""Prompt processing at a BASELINE shape (default llama2-7B): rama_prefill (fp32 MFMA GEMMs, the
weights streamed once per 128 positions) vs the same positions decoded one forward() at a time.
Usage: python tools/prefill_bench.py [n_positions,...] [shape] [positions per pass: 128 | 64] [parity]      Prints one JSON line per length."""
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

lengths = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64]
shape = sys.argv[2] if len(sys.argv) > 2 else "llama2-7B"
per_pass = int(sys.argv[3]) if len(sys.argv) > 3 else 128
parity = len(sys.argv) > 4 and sys.argv[4] == "parity"      # parity mode: the chain-order token-batch kernels, 32 positions per pass
d, h, L, H, V, seq, shared = SHAPES[shape]
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
check(dev.lib.rama_set_tuning(dev.ctx, b"prefill_tok", per_pass))
if parity:
    check(dev.lib.rama_set_tuning(dev.ctx, b"ref_order", 1))
model = rama_amd.Model.synth(dev, cfg, seed=0)
a, b = rama_amd.Engine(dev, model), rama_amd.Engine(dev, model)
for n in lengths:
    toks = [1] + [int(v) for v in np.random.default_rng(0).integers(2, V, n - 1)]
    arr = (C.c_int32 * n)(*toks)

    def prefill():
        check(dev.lib.rama_prefill(dev.ctx, C.byref(model.ccfg), C.byref(model.weights), C.byref(a.state), arr, n, 0))
    prefill(); dev.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); prefill(); dev.sync(); best = min(best, time.perf_counter() - t0)
    for i, t in enumerate(toks): b.forward(t, i)
    dev.sync()
    t0 = time.perf_counter()
    for i, t in enumerate(toks): b.forward(t, i)
    dev.sync(); t_seq = time.perf_counter() - t0
    diff = float(np.abs(a.logits() - b.logits()).max())
    print(json.dumps({"shape": shape, "positions": n, "per_pass": 32 if parity else per_pass, "mode": "parity" if parity else "fast", "prefill_ms": round(best * 1e3, 2), "sequential_ms": round(t_seq * 1e3, 2),
                      "prefill_tok_s": round(n / best, 1), "sequential_tok_s": round(n / t_seq, 1), "speedup": round(t_seq / best, 2),
                      "max_abs_logit_diff_last_position": diff}), flush=True)
