#!/usr/bin/env python3
"""The 1:1 op path against the resident model's fused entry at LONG contexts, parity mode: a two-layer model of llama2-7B's layer shape, both caches filled
with the same random rows, one token at positions around every switch of the attention (128: spread form; 256; 1 000; 1 900).  Logits and run state bit for
bit.  (The committed tests compare the paths at the first positions; this is the long-context complement, run by hand: python tools/ops_long_check.py)"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from rama_amd._lib import check

dev = rama_amd.Hip(0)
d, h, L, H, V, seq = 4096, 11008, 2, 32, 640, 2048
cfg = rama_amd.Config(d, h, L, H, H, V, seq, False)
model = rama_amd.Model.synth(dev, cfg, seed=3)
ws = rama_amd.TransformerWeights.synth(cfg, 3, dev)
wv = rama_amd.TransformerWeightsView.from_gpu_ws(ws)
rs = rama_amd.RunState.from_config(cfg, dev)
rsv = rama_amd.RunStateView.from_rs(rs)
ref = rama_amd.Engine(dev, model)
check(dev.lib.rama_set_tuning(dev.ctx, b"ref_order", 1))
rng = np.random.default_rng(7)
kc = (rng.standard_normal(L * seq * d) * 0.5).astype(np.float32)
vc = (rng.standard_normal(L * seq * d) * 0.5).astype(np.float32)
bad = 0
for pos in (3, 127, 128, 129, 200, 255, 256, 257, 700, 1024, 1900, 2047):
    ref.set_buffer("key_cache", kc); ref.set_buffer("value_cache", vc)
    dev.upload_into(rsv.key_cache, kc); dev.upload_into(rsv.value_cache, vc)
    rama_amd.forward(cfg, wv, rsv, 11, pos, dev)
    ref.forward(11, pos)
    lg, rl = dev.download(rsv.logits), ref.logits()
    same = np.array_equal(lg.view(np.uint32), rl.view(np.uint32))
    xs = np.array_equal(dev.download(rsv.x).view(np.uint32), ref.buffer("x", d).view(np.uint32))
    print(f"pos {pos:5d}: logits {'identical' if same else 'DIFFER'} (max |d| {float(np.abs(lg - rl).max()):.3g}), x {'identical' if xs else 'DIFFER'}", flush=True)
    bad += (not same) + (not xs)
print("FAILED" if bad else "all identical")
sys.exit(1 if bad else 0)
