#!/usr/bin/env python3
"""llama2-7B decode at short contexts: the attention launch with 16 waves per head (default) vs the
small-attention variant (4 or 8 waves, below a position limit).  Prints one JSON line per setting."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from bench import PROMPT, SHAPES
name = sys.argv[1] if len(sys.argv) > 1 else "llama2-7B"
d, h, L, H, V, seq, shared = SHAPES[name]
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
eng = rama_amd.Engine(dev, rama_amd.Model.synth(dev, cfg, seed=0))
eng.set_graph_mode(True)
settings = [(0, 4, 0), (-1, 4, 64), (-1, 4, 128), (-1, 8, 64), (-1, 8, 128), (-1, 8, 192), (-1, 8, 256), (0, 4, 0)]
for sa, wv, lim in settings:
    eng.set_tuning("small_attn", sa); eng.set_tuning("small_attn_waves", wv); eng.set_tuning("small_attn_pos", lim)
    best = 0.0
    for _ in range(3):
        eng.decode_begin(1, 0, PROMPT); eng.decode_steps(8); dev.sync()
        t0 = time.perf_counter(); eng.decode_steps(128); dev.sync()
        best = max(best, 128 / (time.perf_counter() - t0))
    print(json.dumps({"small_attn": sa, "waves": wv, "pos_limit": lim, "tok_s_pos8_135": round(best, 2)}), flush=True)
