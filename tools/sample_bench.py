#!/usr/bin/env python3
"""llama2-7B shape at temperature 1 (the reference README's `-r 1` bench setting): tokens/s with the
top-p sampler on the device inside the chained loop (rama_generate) vs the trait-level loop that
calls Device::sample per token (rama_sample_topp: device sampler + a 4-byte download and a sync
every token; the reference's gpu.rs:149-173 downloads all logits and samples on the host), vs greedy."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from rama_amd.transformer import RunStateView, RunState, TransformerWeightsView
from bench import PROMPT, SHAPES
name = sys.argv[1] if len(sys.argv) > 1 else "llama2-7B"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 96
d, h, L, H, V, seq, shared = SHAPES[name]
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
model = rama_amd.Model.synth(dev, cfg, seed=0)
eng = rama_amd.Engine(dev, model)
eng.set_graph_mode(not (len(sys.argv) > 3 and sys.argv[3] == "eager"))   # `eager`: for rocprofv3, which (ROCm 7.2) crashes in
# hipGraphLaunch once a process has instantiated a second 160-node graph (seen on the greedy path too)
u = 0.2721174359321594
out = {}
for label, T in (("greedy_device", 0.0), ("topp_device", 1.0)):
    eng.generate(PROMPT, 8, T, 0.9, u); dev.sync()
    t0 = time.perf_counter(); toks = eng.generate(PROMPT, steps, T, 0.9, u); dt = time.perf_counter() - t0
    out[label] = round(steps / dt, 1)
# per-token loop over the same fused forward + Device::sample (one sync per token)
import ctypes as C
from rama_amd._lib import check
eng.decode_sampler(0.0)
t0 = time.perf_counter()
token = 1
nxt = C.c_int32()
for pos in range(steps):
    eng.forward(token, pos)
    if pos < len(PROMPT):
        token = PROMPT[pos]
    else:
        check(dev.lib.rama_sample_topp(dev.ctx, eng.state.logits, V, 1.0, 0.9, u, C.byref(nxt)))
        token = nxt.value
dt = time.perf_counter() - t0
out["topp_trait_loop"] = round(steps / dt, 1)
print(json.dumps({"config": name, "steps": steps, "tok_s": out}))
