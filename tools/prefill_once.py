#!/usr/bin/env python3
"""One rama_prefill of n positions at a BASELINE shape and nothing else: the workload for counter
collection (rocprofv3 --pmc serialises every kernel; tools/prefill_bench.py's sequential reference is
ten thousand launches).  Usage: python tools/prefill_once.py [n_positions] [shape] [parity]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import rama_amd
from rama_amd._lib import check
from bench import SHAPES

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shape = sys.argv[2] if len(sys.argv) > 2 else "llama2-7B"
parity = len(sys.argv) > 3 and sys.argv[3] == "parity"
d, h, L, H, V, seq, shared = SHAPES[shape]
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
if parity:
    check(dev.lib.rama_set_tuning(dev.ctx, b"ref_order", 1))
model = rama_amd.Model.synth(dev, cfg, seed=0)
eng = rama_amd.Engine(dev, model)
toks = [1] + [int(v) for v in np.random.default_rng(0).integers(2, V, n - 1)]
arr = (C.c_int32 * n)(*toks)
check(dev.lib.rama_prefill(dev.ctx, C.byref(model.ccfg), C.byref(model.weights), C.byref(eng.state), arr, n, 0))
dev.sync()
print("prefill of", n, "positions done")
