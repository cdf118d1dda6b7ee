#!/usr/bin/env python3
"""Device top-p sampler (rama_sample_topp_dev) alone: microseconds per call on flat, ordinary and
peaked logit vectors, for the shipped path and for each earlier one (rama_set_tuning switches):

    r4            statistics once + 1024-entry block sorts + pair ranking with masses + scatter + the running sums by 32 workgroups
    r4 scan pick  ... with round 2's one-workgroup scan rounds for the running sums ("topp_dist" = 0)
    r4 pairs      round 3's 2048-entry block sorts, ranked by (block, block) pairs ("topp_block" = 2048)
    r3            round 3: 2048-entry block sorts, one workgroup searches all blocks in its LDS ("topp_pairs" = 0)
    global        ranking through global memory + the staged lane ripple (any vocabulary size; "topp_sort" = 0)

HIP events on the context's stream around `reps` back-to-back calls.

    python tools/sampler_bench.py [reps]
"""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from rama_amd._lib import check

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = rama_amd.Hip(0)
rng = np.random.default_rng(0)
n = 32000
DEFAULTS = {b"topp_sort": 1, b"topp_pairs": 1, b"topp_block": 1024, b"topp_dist": 1}
PATHS = [("r4", {}), ("r4 scan pick", {b"topp_dist": 0}), ("r4 pairs", {b"topp_block": 2048}), ("r3", {b"topp_block": 2048, b"topp_pairs": 0}),
         ("global", {b"topp_sort": 0})]
out = {}
for name, scale in [("flat (std 0.05: all 32000 kept)", 0.05), ("std 1", 1.0), ("std 3", 3.0), ("peaked (std 8)", 8.0)]:
    x = (rng.standard_normal(n) * scale).astype(np.float32)
    d_x = dev.allocate(x)
    d_r = dev.allocate(np.zeros(1, dtype=np.float32))
    row = {}
    tokens = set()
    for path, tune in PATHS:
        for k, v in {**DEFAULTS, **tune}.items():
            check(dev.lib.rama_set_tuning(dev.ctx, k, v))
        for _ in range(5):
            check(dev.lib.rama_sample_topp_dev(dev.ctx, d_x.ptr, n, 1.0, 0.9, 0.2721174359321594, d_r.ptr))
        check(dev.lib.rama_timer_start(dev.ctx))
        for _ in range(reps):
            check(dev.lib.rama_sample_topp_dev(dev.ctx, d_x.ptr, n, 1.0, 0.9, 0.2721174359321594, d_r.ptr))
        ms = C.c_float()
        check(dev.lib.rama_timer_stop(dev.ctx, C.byref(ms)))
        row[path] = round(ms.value * 1000.0 / reps, 2)
        tokens.add(int(dev.download(d_r).view(np.int32)[0]))
    for k, v in DEFAULTS.items():
        check(dev.lib.rama_set_tuning(dev.ctx, k, v))
    z = x.astype(np.float64); pr = np.exp(z - z.max()); pr /= pr.sum()
    srt = np.sort(pr)[::-1]
    row["kept"] = int((pr > 0.1 / (n - 1)).sum())
    row["crossing"] = int(np.searchsorted(np.cumsum(srt), 0.9))
    row["same_token_on_every_path"] = len(tokens) == 1
    out[name] = row
    d_x.free(); d_r.free()
print(json.dumps({"unit": "us per rama_sample_topp_dev call, eager back-to-back launches", "n": n, "reps": reps, "cases": out}, indent=1))
