#!/usr/bin/env python3
"""Parity-mode decode rate and per-kernel times for several chain-order matvec geometries (chain_d = 100 W + D,
0 = the built-in choice by row groups per CU).  Usage: python tools/chain_sweep.py [values...]"""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from bench import PROMPT, SHAPES
vals = [int(v) for v in sys.argv[1:]] or [0, 116, 132, 216, 232, 416, 432]
d, h, L, H, V, seq, shared = SHAPES["llama2-7B"]
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
eng = rama_amd.Engine(dev, rama_amd.Model.synth(dev, cfg, seed=0))
eng.set_tuning("ref_order", 1)
for v in vals:
    eng.set_tuning("chain_d", v)
    eng.set_graph_mode(True)
    eng.decode_begin(1, 0, PROMPT); eng.decode_steps(8); dev.sync()
    t0 = time.perf_counter(); eng.decode_steps(64); dev.sync()
    tok = 64 / (time.perf_counter() - t0)
    eng.set_graph_mode(False)
    ks = {}
    for k in ("qkv", "wo", "w13", "w2", "cls"):
        eng.decode_begin(1, 72, [])
        ms, n = eng.kprof(k, 4)
        ks[k] = round(ms * 1e3, 1)
    print(json.dumps({"chain_d": v, "tok_s": round(tok, 1), "us": ks}), flush=True)
