#!/usr/bin/env python3
"""How often the one-pass exact sequential sum (csrc/chain.hpp seq_sum_predict) holds in a parity-mode generation."""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from bench import PROMPT, SHAPES
name = sys.argv[1] if len(sys.argv) > 1 else "llama2-7B"
d, h, L, H, V, seq, shared = SHAPES[name]
L = int(sys.argv[2]) if len(sys.argv) > 2 else L
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
eng = rama_amd.Engine(dev, rama_amd.Model.synth(dev, cfg, seed=0))
eng.set_tuning("ref_order", 1)
eng.decode_begin(1, 0, PROMPT)
dev.lib.rama_internal_pred_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_int]
a, b = C.c_uint(), C.c_uint()
dev.lib.rama_internal_pred_stats(dev.ctx, C.byref(a), C.byref(b), 1)
eng.decode_steps(32)
dev.lib.rama_internal_pred_stats(dev.ctx, C.byref(a), C.byref(b), 0)
print({"held": a.value, "fell_back": b.value})
