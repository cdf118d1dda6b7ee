#!/usr/bin/env python3
"""Aggregate decode throughput with B independent sequences sharing each weight pass
(rama_decode_batch, B = 1..128) at the llama2-7B shape; greedy tokens are fed back through the host
(argmax of each sequence's logits on the device, 4 bytes each).  Prints one JSON line.
Usage: python tools/batch_bench.py [shape] [steps] [seq_len override] [parity]"""
import ctypes as C
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from rama_amd._lib import check
from bench import SHAPES
name = sys.argv[1] if len(sys.argv) > 1 else "llama2-7B"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
d, h, L, H, V, seq, shared = SHAPES[name]
parity = len(sys.argv) > 4 and sys.argv[4] == "parity"      # parity mode: chain-order token-batch kernels, 32 sequences per weight pass
if len(sys.argv) > 3: seq = int(sys.argv[3])      # a shorter context: 128 sequences x 2 048 positions of KV cache would not fit beside the model
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
model = rama_amd.Model.synth(dev, cfg, seed=0)
if parity:
    check(dev.lib.rama_set_tuning(dev.ctx, b"ref_order", 1))
out = {}
for B in (1, 8, 16, 32, 64, 128):
    if B * 2 * L * seq * d * 4 > 150e9: continue
    engs = [rama_amd.Engine(dev, model) for _ in range(B)]
    cur = [1 + i for i in range(B)]
    nxt = C.c_int32()
    def step(pos):
        rama_amd.decode_batch(engs, cur, [pos] * B)
        for i, e in enumerate(engs):
            check(dev.lib.rama_sample_argmax(dev.ctx, e.state.logits, V, C.byref(nxt)))
            cur[i] = nxt.value
    for p in range(4): step(p)
    dev.sync()
    t0 = time.perf_counter()
    for p in range(4, 4 + steps): step(p)
    dev.sync()
    dt = time.perf_counter() - t0
    out[B] = {"ms_per_step": round(dt * 1e3 / steps, 3), "aggregate_tok_s": round(B * steps / dt, 1)}
    # the same chained on the device: cursors and argmax per sequence in device memory, one hipGraph per step
    for graph in (() if parity else (0, 1)):      # the chained entry points are fast-mode only
        engs[0].set_graph_mode(bool(graph))
        states = (rama_amd._lib.rama_run_state * B)(*[e.state for e in engs])
        toks = (C.c_int32 * B)(*[1 + i for i in range(B)])
        poss = (C.c_int32 * B)(*([0] * B))
        check(dev.lib.rama_decode_batch_begin(dev.ctx, C.byref(model.ccfg), C.byref(model.weights), states, toks, poss, B, steps + 4))
        check(dev.lib.rama_decode_batch_steps(dev.ctx, 4))
        dev.sync()
        t0 = time.perf_counter()
        check(dev.lib.rama_decode_batch_steps(dev.ctx, steps))
        dev.sync()
        dt = time.perf_counter() - t0
        out[B]["chained_graph" if graph else "chained_eager"] = {"ms_per_step": round(dt * 1e3 / steps, 3), "aggregate_tok_s": round(B * steps / dt, 1)}
    engs[0].set_graph_mode(False)
    for e in engs: e.free()
print(json.dumps({"config": name, "mode": "parity" if parity else "fast", "seq_len": seq, "steps": steps, "by_batch": out}))
