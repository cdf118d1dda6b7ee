#!/usr/bin/env python3
"""tokens/s vs start position for several split-T thresholds (attention: one workgroup per head
below the threshold, split over the sequence + combine launch above)."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from bench import SHAPES
name = sys.argv[1] if len(sys.argv) > 1 else "stories110M"
d, h, L, H, V, seq, shared = SHAPES[name]
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
eng = rama_amd.Engine(dev, rama_amd.Model.synth(dev, cfg, seed=0))
eng.set_graph_mode(True)
out = {}
starts = [s for s in (100, 200, 300, 400, 600, 900, 1500, 1900) if s + 70 < seq]
for thr in (0, 128, 256, 384, 512, 768, 1 << 20):
    eng.set_tuning("split_pos", thr)
    row = {}
    for s in starts:
        best = 0
        for _ in range(2):
            eng.decode_begin(1, s, []); eng.decode_steps(4); dev.sync()
            t0 = time.perf_counter(); eng.decode_steps(48); dev.sync()
            best = max(best, 48 / (time.perf_counter() - t0))
        row[s] = round(best, 1)
    out[thr] = row
print(json.dumps({"config": name, "tok_s[threshold][start]": out}))
