#!/usr/bin/env python3
"""A/B the matvec tuning knobs on the llama2-7B-shaped synthetic model, interleaved rounds in
ONE process (cdna_hip_programming.md rule 24).  Prints tokens/s and per-kernel-class times."""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

import rama_amd  # noqa: E402
from bench import PROMPT, SHAPES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="llama2-7B")
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--geom", default="0,1,2,3")
    ap.add_argument("--kprof", type=int, default=1)
    args = ap.parse_args()
    d, h, L, H, V, seq, shared = SHAPES[args.config]
    cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
    dev = rama_amd.Hip(0)
    model = rama_amd.Model.synth(dev, cfg, seed=0)
    eng = rama_amd.Engine(dev, model)
    bytes_ = rama_amd.algorithmic_bytes(cfg)
    variants = [int(v) for v in args.geom.split(",")]
    res = {v: [] for v in variants}
    for rnd in range(args.rounds):
        for v in variants:
            eng.set_tuning("geom", v)
            eng.set_graph_mode(True)
            eng.decode_begin(1, 0, PROMPT)
            eng.decode_steps(4)
            dev.sync()
            t0 = time.perf_counter()
            eng.decode_steps(args.steps)
            dev.sync()
            dt = time.perf_counter() - t0
            res[v].append(args.steps / dt)
    out = {}
    for v in variants:
        r = sorted(res[v])
        out[f"geom{v}"] = {"tok_s_median": round(r[len(r) // 2], 2), "tok_s_max": round(r[-1], 2),
                         "GBps_median": round(r[len(r) // 2] * bytes_["token"] / 1e9, 1)}
        if args.kprof:
            eng.set_tuning("geom", v)
            eng.set_graph_mode(False)
            ks = {}
            for k in ("qkv", "attn", "wo", "w13", "w2", "cls"):
                eng.decode_begin(1, 0, PROMPT)
                eng.decode_steps(4)
                avg_ms, n = eng.kprof(k, 8)
                b = bytes_.get(k)
                ks[k] = [round(avg_ms * 1e3, 2), round(b / (avg_ms * 1e-3) / 1e9) if b else None]
            out[f"geom{v}"]["kernels_us_GBps"] = ks
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
