#!/usr/bin/env python3
"""Soak of the one-launch stage (csrc/layer_fused.hpp): the same generation again and again, graphs on and off; tokens and the last
logits must be bit-identical to the first round's every time (a stale hand-off word would show as a difference).
Usage: python tools/fused_soak.py [rounds]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd  # noqa: E402
from bench import PROMPT, SHAPES  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = rama_amd.Hip(0)
for name in ("stories15M", "stories110M"):
    d, h, L, H, V, seq, shared = SHAPES[name]
    cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
    eng = rama_amd.Engine(dev, rama_amd.Model.synth(dev, cfg, seed=0))
    steps = min(seq, 250)
    first, bad, t0 = None, 0, time.perf_counter()
    for r in range(rounds):
        eng.set_graph_mode(r % 3 != 0)
        toks = eng.generate(PROMPT if r % 2 else [], steps)
        got = (r % 2, tuple(toks), eng.logits().tobytes())
        if first is None or first.get(r % 2) is None:
            first = first or {}
            first[r % 2] = got
        elif got != first[r % 2]:
            bad += 1
    print(f"{name}: {rounds} generations of {steps} tokens, {bad} differ from the first of their kind, {time.perf_counter() - t0:.1f} s", flush=True)
    eng.free(); eng.model.free()
