#!/usr/bin/env python3
"""Diagnostic: per-phase timeline of the persistent decode step (one workgroup's 100 MHz stamps)."""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from rama_amd._lib import check
from bench import PROMPT, SHAPES

d, h, L, H, V, seq, shared = SHAPES["llama2-7B"]
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
eng = rama_amd.Engine(dev, rama_amd.Model.synth(dev, cfg, seed=0))
eng.set_tuning("persist", 1)
eng.decode_begin(1, 0, PROMPT)
eng.decode_steps(8)
dev.sync()
names = {0: "qkv", 1: "attn", 2: "wo", 3: "w13", 4: "w2"}
for wg in (int(a) for a in (sys.argv[1:] or ["100", "3"])):
    buf = (C.c_ulonglong * (8 * 200))()
    n = C.c_int()
    check(dev.lib.rama_persist_stamps(dev.ctx, C.byref(eng.model.ccfg), C.byref(eng.model.weights), C.byref(eng.state), wg, buf, 200, C.byref(n)))
    t0 = buf[0]
    tot = {}
    for ph in range(n.value):
        s = [buf[ph * 8 + k] for k in range(5)]
        kind = names.get(ph % 5, "?") if ph < 5 * L else "cls"
        stage = (s[1] - s[0]) / 100 if s[1] else 0.0
        run = (s[2] - (s[1] or s[0])) / 100 if s[2] else 0.0
        arrive = (s[3] - (s[2] or s[0])) / 100
        wait = (s[4] - s[3]) / 100
        whole = (s[4] - s[0]) / 100
        a = tot.setdefault(kind, [0, 0, 0, 0, 0, 0])
        for i, v in enumerate((stage, run, arrive, wait, whole)):
            a[i] += v
        a[5] += 1
    print(f"wg {wg}: step total {(buf[(n.value - 1) * 8 + 4] - t0) / 100:.1f} us")
    for k, a in tot.items():
        c = a[5]
        print(f"  {k:5s} x{c:3d}  avg us: stage {a[0]/c:6.2f}  steps {a[1]/c:6.2f}  arrive(+wave0 prefetch) {a[2]/c:6.2f}  wait {a[3]/c:6.2f}  phase {a[4]/c:6.2f}")
