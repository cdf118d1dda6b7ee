#!/usr/bin/env python3
"""One-off: FULL llama2-7B shape (32 layers, 27 GB of synthetic weights) -- HIP path vs the CPU
oracle on the same weights, prompt 'once upon a time', greedy.  Needs ~32 GB of host RAM and a
minute of CPU; prints one JSON line (max |logit difference| per position, also vs the
fp64-accumulated arbiter at position 0).  Usage: python tools/parity_7b.py [n_positions]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import synth as S  # noqa: E402

n_pos = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d, h, L, H, V, seq = 4096, 11008, 32, 32, 32000, 2048
cfg = O.Config(d, h, L, H, H, V, seq, False)
t0 = time.time()
rope = S.rope_tables(seq, d // H)
w = S.synth_weights(cfg, 0, rope=rope)
t_gen = time.time() - t0
orc = O.Oracle(cfg, w, threads=16)
dev = rama_amd.Hip(0)
rcfg = rama_amd.Config(d, h, L, H, H, V, seq, False)
eng = rama_amd.Engine(dev, rama_amd.Model.synth(dev, rcfg, 0, rope=rope))
prompt = [10646, 2501, 263, 931]
token, diffs, toks_cpu, toks_gpu = 1, [], [], []
fed, logits_cpu = [], []
t_cpu = 0.0
for pos in range(n_pos):
    t1 = time.time()
    lo = orc.forward(token, pos).copy()
    t_cpu += time.time() - t1
    fed.append(int(token)); logits_cpu.append(lo)
    eng.forward(token, pos)
    lg = eng.logits()
    diffs.append(float(np.abs(lg - lo).max()))
    toks_cpu.append(int(O.argmax(lo))); toks_gpu.append(int(np.flatnonzero(lg == lg.max())[-1]))
    token = prompt[pos] if pos < len(prompt) else toks_cpu[-1]
# fp64 arbiter at position 0 of a fresh state (how far each fp32 path sits from exact)
orc2 = O.Oracle(cfg, w, threads=16)
l64 = orc2.forward_f64(1, 0).copy()
orc3 = O.Oracle(cfg, w, threads=16)
l32 = orc3.forward(1, 0).copy()
eng2 = rama_amd.Engine(dev, eng.model)
eng2.forward(1, 0)
# the same positions through the batched-prompt prefill (8 per weight pass) into a fresh state, and
# the device top-p sampler on those logits against the oracle's Device::sample
import ctypes as C
from rama_amd._lib import check
eng3 = rama_amd.Engine(dev, eng.model)
npf = min(n_pos, 21)
arr = (C.c_int32 * npf)(*fed[:npf])
check(dev.lib.rama_prefill(dev.ctx, C.byref(eng3.model.ccfg), C.byref(eng3.model.weights), C.byref(eng3.state), arr, npf, 0))
pf_diff = float(np.abs(eng3.logits() - logits_cpu[npf - 1]).max())
u = 0.2721174359321594
nxt = C.c_int32()
check(dev.lib.rama_sample_topp(dev.ctx, eng3.state.logits, V, 1.0, 0.9, u, C.byref(nxt)))
topp_equal = int(nxt.value) == int(O.sample(logits_cpu[npf - 1].copy(), 1.0, 0.9, u))
# two independent sequences in one weight pass: sequence A continues the run above at position
# npf, sequence B starts a new generation at position 0
engB = rama_amd.Engine(dev, eng.model)
rama_amd.decode_batch([eng3, engB], [fed[npf] if npf < len(fed) else toks_cpu[npf - 1], 1], [npf, 0])
batch_diff_b = float(np.abs(engB.logits() - logits_cpu[0]).max())
batch_diff_a = float(np.abs(eng3.logits() - logits_cpu[npf]).max()) if npf < len(logits_cpu) else None
print(json.dumps({"shape": "llama2-7B (32 layers)", "positions": n_pos, "max_abs_logit_diff_per_pos": [round(v, 9) for v in diffs],
                  "worst": max(diffs), "bar": 1e-4, "argmax_equal": toks_cpu == toks_gpu,
                  "pos0_cpu32_vs_f64": float(np.abs(l32 - l64).max()), "pos0_hip_vs_f64": float(np.abs(eng2.logits() - l64).max()),
                  "logit_abs_max": float(np.abs(l64).max()),
                  "prefill_positions": npf, "prefill_last_logit_diff": pf_diff, "topp_token_equal_on_prefill_logits": topp_equal,
                  "decode_batch_diff_seqA_at_pos_npf": batch_diff_a, "decode_batch_diff_seqB_at_pos0": batch_diff_b, "weights_gen_s": round(t_gen, 1), "cpu_s_per_token": round(t_cpu / n_pos, 3)}))
