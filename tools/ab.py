#!/usr/bin/env python3
"""A/B one tuning key (0 vs 1, or --values a,b,..) on the llama2-7B-shaped synthetic model, interleaved rounds in one
process; prints tokens/s per arm.  Usage: python tools/ab.py merge [--steps 64] [--rounds 4]"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd  # noqa: E402
from bench import PROMPT, SHAPES  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("key")
ap.add_argument("--config", default="llama2-7B")
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--start", type=int, default=0, help="position the timed steps start at")
ap.add_argument("--values", default="0,1", help="the two (or more) values of the key to compare")
args = ap.parse_args()
d, h, L, H, V, seq, shared = SHAPES[args.config]
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
eng = rama_amd.Engine(dev, rama_amd.Model.synth(dev, cfg, seed=0))
eng.set_graph_mode(True)
vals = [int(v) for v in args.values.split(',')]
res = {v: [] for v in vals}
toks = {}
for rnd in range(args.rounds):
    for v in vals:
        eng.set_tuning(args.key, v)
        eng.decode_begin(1, args.start, PROMPT if args.start == 0 else [])
        eng.decode_steps(4)
        dev.sync()
        t0 = time.perf_counter()
        eng.decode_steps(args.steps)
        dev.sync()
        res[v].append(round(args.steps / (time.perf_counter() - t0), 2))
        toks[v] = eng.decode_tokens()
print(json.dumps({"key": args.key, "config": args.config, "start": args.start, "tok_s": res, "same_tokens": all(toks[v] == toks[vals[0]] for v in vals)}))
