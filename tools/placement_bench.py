#!/usr/bin/env python3
"""Does placement tuning of W3 pay on this box?  tokens/s before and after rama_model_tune_placement
(llama2-7B shape, hipGraph, pos 8..135), in one process."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from bench import PROMPT, SHAPES
d, h, L, H, V, seq, shared = SHAPES["llama2-7B"]
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
model = rama_amd.Model.synth(dev, cfg, seed=0)
def rate():
    eng = rama_amd.Engine(dev, model)
    eng.set_graph_mode(True)
    best = 0
    for _ in range(2):
        eng.decode_begin(1, 0, PROMPT); eng.decode_steps(8); dev.sync()
        t0 = time.perf_counter(); eng.decode_steps(128); dev.sync()
        best = max(best, 128 / (time.perf_counter() - t0))
    eng.set_graph_mode(False); eng.free()
    return round(best, 1)
before = rate()
eng = rama_amd.Engine(dev, model)
eng.set_graph_mode(True)
def step_ms():
    eng.decode_begin(1, 0, PROMPT); eng.decode_steps(6); dev.sync()
    t = time.perf_counter(); eng.decode_steps(24); dev.sync()
    return (time.perf_counter() - t) * 1e3 / 24
rep = model.tune_placement(int(sys.argv[1]) if len(sys.argv) > 1 else 8, timer=step_ms)
eng.set_graph_mode(False); eng.free()
after = rate()
print(json.dumps({"tok_s_before": before, "tok_s_after": after, "tuning": rep}))
