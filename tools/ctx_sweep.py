#!/usr/bin/env python3
"""Decode throughput and attention launch time vs context length (llama2-7B shape, synthetic weights).
The cache rows before the start position are zeros (uniform attention): timing only."""
import json
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from bench import SHAPES

starts = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 200, 380, 400, 1000, 1900]
d, h, L, H, V, seq, shared = SHAPES["llama2-7B"]
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
eng = rama_amd.Engine(dev, rama_amd.Model.synth(dev, cfg, seed=0))
rows = []
for s in starts:
    eng.set_graph_mode(True)
    eng.decode_begin(1, s, [])
    eng.decode_steps(4); dev.sync()
    t0 = time.perf_counter(); eng.decode_steps(32); dev.sync()
    toks = 32 / (time.perf_counter() - t0)
    eng.set_graph_mode(False)
    eng.decode_begin(1, s, [])
    eng.decode_steps(2)
    attn_ms, n = eng.kprof("attn", 8)
    rows.append({"pos": s, "tok_s": round(toks, 1), "attn_us": round(attn_ms * 1e3, 2), "kv_MB_per_layer": round(2 * (s + 1) * d * 4 / 1e6, 2)})
print(json.dumps(rows))
