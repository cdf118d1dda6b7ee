#!/usr/bin/env python3
"""generate() vs generate_stream() (tokens handed over as they are produced, from the host-visible ring; the loop stays
chained on the device) vs the per-token host loop (forward + sample, one round trip per token), greedy, hipGraph.
Usage: python tools/stream_bench.py [shape] [steps]      Prints one JSON line."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from bench import PROMPT, SHAPES
name = sys.argv[1] if len(sys.argv) > 1 else "stories15M"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
d, h, L, H, V, seq, shared = SHAPES[name]
steps = min(steps, seq)
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
eng = rama_amd.Engine(dev, rama_amd.Model.synth(dev, cfg, seed=0))
eng.set_graph_mode(True)
res = {}
for rnd in range(3):
    t0 = time.perf_counter(); a = eng.generate(PROMPT, steps); ta = time.perf_counter() - t0
    stamps = []
    t0 = time.perf_counter(); b = eng.generate_stream(PROMPT, steps, lambda i, t: stamps.append(time.perf_counter())); tb = time.perf_counter() - t0
    assert a == b
    gaps = [stamps[i + 1] - stamps[i] for i in range(len(PROMPT) + 2, steps - 1)]
    res = {"generate_tok_s": round(steps / ta, 1), "generate_stream_tok_s": round(steps / tb, 1),
           "first_sampled_token_ms": round((stamps[len(PROMPT)] - t0) * 1e3, 3),
           "median_gap_us": round(sorted(gaps)[len(gaps) // 2] * 1e6, 1), "max_gap_us": round(max(gaps) * 1e6, 1)}
print(json.dumps({"config": name, "steps": steps, **res}))
