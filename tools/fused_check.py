#!/usr/bin/env python3
"""The one-launch stage (csrc/layer_fused.hpp, tuning key "fused") against the separate launches: logits, x, caches of a few
steps on the small shapes, then tokens/s of both (tools/ab.py does the interleaved A/B).  Usage: python tools/fused_check.py"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd  # noqa: E402
from bench import SHAPES  # noqa: E402

dev = rama_amd.Hip(0)
for name in ("stories15M", "stories110M"):
    d, h, L, H, V, seq, shared = SHAPES[name]
    cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
    model = rama_amd.Model.synth(dev, cfg, seed=0)
    outs = {}
    for fused in (0, 1):
        eng = rama_amd.Engine(dev, model)
        eng.set_tuning("fused", fused)
        got = []
        for pos, tok in enumerate([1, 5, 9, 200, 31, 7, 7, 12]):
            eng.forward(tok, pos)
            got.append(eng.logits().copy())
        outs[fused] = (np.stack(got), eng.buffer("x", d).copy(), eng.buffer("key_cache", 8 * d).copy(), eng.buffer("value_cache", 8 * d, (L - 1) * seq * d).copy())
        eng.free()
    for i, what in enumerate(("logits", "x", "key_cache[0]", "value_cache[last]")):
        a, b = outs[0][i], outs[1][i]
        print(f"{name} {what}: max |diff| {np.abs(a - b).max():.3e} (scale {np.abs(a).max():.3e}) argmax same {bool((a.reshape(len(a), -1).argmax(-1) == b.reshape(len(b), -1).argmax(-1)).all()) if i == 0 else '-'}", flush=True)
    eng = rama_amd.Engine(dev, model)
    eng.set_graph_mode(True)
    for rnd in range(2):
        for fused in (0, 1):
            eng.set_tuning("fused", fused)
            eng.decode_begin(1, 0, [])
            eng.decode_steps(8)
            dev.sync()
            t0 = time.perf_counter()
            eng.decode_steps(min(200, seq - 10))
            dev.sync()
            print(f"{name} fused={fused}: {min(200, seq - 10) / (time.perf_counter() - t0):.0f} tok/s", flush=True)
    eng.free()
    model.free()
