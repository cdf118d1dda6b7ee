#!/usr/bin/env python3
"""Long-context decode at the llama2-7B shape: tokens/s and the attention launches' time vs the number
of split-T slices per head, the cache-load policy and the rows in flight per lane (rama_set_tuning
"attn_nsplit", "attn_nt", "attn_waves", "attn_u").
The cache rows before the start position are zeros (uniform attention): timing only."""
import json
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from bench import SHAPES

starts = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1000, 1900]
d, h, L, H, V, seq, shared = SHAPES["llama2-7B"]
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
eng = rama_amd.Engine(dev, rama_amd.Model.synth(dev, cfg, seed=0))
combos = [(1, 8, 8, 16), (1, 8, 8, 8), (1, 16, 8, 16), (1, 16, 8, 8), (1, 8, 4, 16), (1, 16, 4, 16), (1, 8, 16, 8), (0, 8, 8, 16)]
for nt, ns, wv, au in combos:
    if True:
        eng.set_tuning("attn_nt", nt); eng.set_tuning("attn_nsplit", ns); eng.set_tuning("attn_waves", wv); eng.set_tuning("attn_u", au)
        row = {"attn_nt": nt, "attn_nsplit": ns, "attn_waves": wv, "attn_u": au}
        for s in starts:
            eng.set_graph_mode(True)
            best = 0.0
            for _ in range(2):
                eng.decode_begin(1, s, []); eng.decode_steps(4); dev.sync()
                t0 = time.perf_counter(); eng.decode_steps(48); dev.sync()
                best = max(best, 48 / (time.perf_counter() - t0))
            eng.set_graph_mode(False)
            eng.decode_begin(1, s, []); eng.decode_steps(2)
            attn_ms, n = eng.kprof("attn", 6)
            row[f"pos{s}"] = {"tok_s": round(best, 1), "attn_us": round(attn_ms * 1e3, 2)}
        print(json.dumps(row), flush=True)
