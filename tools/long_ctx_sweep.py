#!/usr/bin/env python3
"""Fast-mode decode rate at a long context for several split-T geometries (slices per head x waves per slice)."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from bench import SHAPES
start = int(sys.argv[1]) if len(sys.argv) > 1 else 1900
d, h, L, H, V, seq, shared = SHAPES["llama2-7B"]
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
eng = rama_amd.Engine(dev, rama_amd.Model.synth(dev, cfg, seed=0))
eng.set_graph_mode(True)
res = {}
for rnd in range(3):
    for ns, wv in ((0, 8), (4, 16), (4, 8), (8, 16), (6, 8), (16, 8)):
        eng.set_tuning("attn_nsplit", ns); eng.set_tuning("attn_waves", wv)
        eng.decode_begin(1, start, []); eng.decode_steps(4); dev.sync()
        t0 = time.perf_counter(); eng.decode_steps(48); dev.sync()
        res.setdefault(f"{ns}x{wv}", []).append(round(48 / (time.perf_counter() - t0), 1))
print(json.dumps({"start": start, "tok_s": res}))
