#!/usr/bin/env python3
"""Stability soak: the same generation over and over for a wall-clock budget; every repeat must
reproduce the first run's tokens exactly (hipGraph replay, merged attention+Wo spins, split-T,
device sampler).  Prints one JSON line."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from bench import PROMPT, SHAPES
name = sys.argv[1] if len(sys.argv) > 1 else "llama2-7B"
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 512
T = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
parity = len(sys.argv) > 5 and sys.argv[5] in ("parity", "bar")      # parity mode: chain-order kernels, prompt through the token-batch kernels
bar = len(sys.argv) > 5 and sys.argv[5] == "bar"                     # [r6] bar mode: parity up to position 127, the fast attention from 128 on
d, h, L, H, V, seq, shared = SHAPES[name]
steps = min(steps, seq)
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
eng = rama_amd.Engine(dev, rama_amd.Model.synth(dev, cfg, seed=0))
eng.set_graph_mode(True)
if parity:
    eng.set_tuning("ref_order", 3 if bar else 1)
ref = eng.generate(PROMPT, steps, T, 0.9, 0.2721174359321594)
t0 = time.time(); reps = 0; bad = 0
while time.time() - t0 < budget:
    got = eng.generate(PROMPT, steps, T, 0.9, 0.2721174359321594)
    reps += 1
    bad += got != ref
import ctypes as C
dist_bad = C.c_uint(0)                                    # diagnostics of the distributed top-p pick (csrc/topp_pick.hpp): a wait that timed out, a prediction that failed
f = dev.lib.rama_internal_topp_dist_bad
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
f(dev.ctx, C.byref(dist_bad))
print(json.dumps({"config": name, "mode": "bar" if bar else ("parity" if parity else "fast"), "temperature": T, "steps": steps, "repeats": reps, "mismatches": bad,
                  "tokens": reps * steps, "tok_s": round(reps * steps / (time.time() - t0), 1), "topp_dist_bad": dist_bad.value}))
sys.exit(1 if bad or dist_bad.value else 0)
