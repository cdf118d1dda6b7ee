#!/usr/bin/env python3
"""bench.py -- decode throughput of the HIP path on llama2-7B-shaped synthetic fp32 weights.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1: one decode step = one forward() (infer.rs:8-53) + greedy sample, chained on the device,
starting at pos 0 like generate() (BOS, then the Rama-BPE ids of 'once upon a time').
N > 1: the layer stack is pipeline-sharded over the N ranks (rank r owns layers
[r*L/N, (r+1)*L/N)) with N sequences in flight; a step advances every sequence by one
token; the residual x[dim] travels rank r -> r+1 and the sampled token id last -> first over
RCCL point-to-point (rama_amd/pipeline.py).

Prints ONE JSON line (see the driver contract): whole-job tokens/s, plus `roofline` for the
dominant kernel (W1|W3 SwiGLU matvec, 43 % of the bytes) and `cpu_baseline` (the oracle --
the C restatement of the reference CPU path -- timed on the host cores, rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

SHAPES = {   # dim, hidden, layers, heads, vocab, seq_len, shared classifier
    "llama2-7B": (4096, 11008, 32, 32, 32000, 2048, False),
    "stories110M": (768, 2048, 12, 12, 32000, 1024, True),
    "stories15M": (288, 768, 6, 6, 32000, 256, True),
}
PROMPT = [10646, 2501, 263, 931]   # Rama-BPE of 'once upon a time' 
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="llama2-7B", choices=list(SHAPES))
    ap.add_argument("--graph", type=int, default=1, help="replay each decode step from a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kprof", action="store_true")
    ap.add_argument("--no-placement-tuning", action="store_true", help="accepted and ignored (round-1 flag: the tuner is gone)")
    ap.add_argument("--pos0", type=int, default=0,
                    help="start the timed generation at this position over a pre-filled (zero) cache: long-context timing, "
                         "N = 1 only; the default 0 is generate()'s own start (BOS + prompt)")
    ap.add_argument("--mode", default="fast", choices=["fast", "parity"],
                    help="parity: every op in the reference's own rounding order (bit-identical logits), on the model's chain-order weight copy")
    ap.add_argument("--cpu-tokens", type=int, default=8)
    ap.add_argument("--cpu-layers", type=int, default=8, help="layers of the CPU baseline's sample (all of them when the model has fewer)")
    return ap.parse_args()


def pmc_traffic(kernel_substr):
    """HBM read bytes per launch of the dominant kernel from the committed PMC pass (a run of
    its own: rocprofv3 --pmc FETCH_SIZE cannot ride along with a timed run), already carrying
    the guide's gfx950 correction (FETCH_SIZE counts 128-B requests at 64 B: x2)."""
    import glob
    files = sorted(glob.glob(str(REPO / "profiles" / "r*_bench_7b_pmc_fetch_size.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        rows = json.load(f)["rows"]
    for r in rows:
        if kernel_substr in r["kernel"] and r["counter"] == "FETCH_SIZE":
            return r["hbm_read_bytes_corrected"], f"profiles/{Path(files[-1]).name}"
    return None, None


def cpu_baseline(shape_name, n_tokens, n_layers_sample):
    """Time the oracle (reference-algorithm CPU restatement, OpenMP over rows/heads like the
    reference's rayon) on a bounded sample of the same workload: `n_layers_sample` of the L
    layers + the classifier, same synthetic weights, same token positions; the layer part is
    scaled by L / n_layers_sample."""
    import numpy as np
    from oracle import oracle as O
    from oracle import synth as S
    d, h, L, H, V, seq, shared = SHAPES[shape_name]
    ls = min(n_layers_sample, L)
    cfg = O.Config(d, h, ls, H, H, V, min(seq, 64), shared)
    full = O.Config(d, h, L, H, H, V, seq, shared)
    spec = S.synth_spec(full)
    w = {}
    for name, shp in O.weight_shapes(cfg):
        if name.startswith("freq_cis"):
            continue
        tag, scale, bias = spec[name]
        w[name] = O.fill_synth(int(np.prod(shp)), 0, tag, scale, bias).reshape(shp)
    w["freq_cis_real"], w["freq_cis_imag"] = S.rope_tables(cfg.seq_len, cfg.head_size)
    # the GPU box shows 256 CPUs but a 1-GPU job owns a 16-CPU share
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    orc = O.Oracle(cfg, w, threads=cores)
    toks = [1] + PROMPT
    t_layers = t_cls = 0.0
    token = 1
    for pos in range(n_tokens):
        t0 = time.perf_counter()
        orc.forward_range(token, pos, 0, ls, True, False)
        t1 = time.perf_counter()
        orc.forward_range(token, pos, ls, ls, False, True)
        t2 = time.perf_counter()
        if pos > 0:   # first token pages the weights in
            t_layers += t1 - t0
            t_cls += t2 - t1
        token = toks[pos + 1] if pos + 1 < len(toks) else O.argmax(orc.s["logits"])
    n = max(n_tokens - 1, 1)
    per_token = (t_layers / n) * (L / ls) + t_cls / n
    return {"value": round(1.0 / per_token, 4), "unit": "tokens/s", "cores": O.lib().oracle_get_threads(),
            "kind": "port",
            "sample": f"{shape_name} shape, {ls} of {L} layers + classifier, {n} tokens after 1 warm-up, "
                      f"layer time scaled x{L / ls:g}; oracle/rama_oracle.c (C restatement of engine/src/device/cpu.rs), "
                      f"OpenMP threads={O.lib().oracle_get_threads()}"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import rama_amd

    d, h, L, H, V, seq, shared = SHAPES[args.config]
    cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)

    # RAMA_FORCE_PIPELINE=1 rehearses the N > 1 code path (process group, HipStage, the
    # grouped exchanges' bookkeeping) with a single rank on a 1-GPU box
    if args.gpus > 1 or os.environ.get("RAMA_FORCE_PIPELINE"):
        from rama_amd.pipeline import run_pipeline_bench
        line = run_pipeline_bench(args, cfg, rank, world, local_rank)
        if rank == 0:
            print(json.dumps(line), flush=True)
        return

    need = args.warmup + args.steps
    dev = rama_amd.Hip(local_rank)
    model = rama_amd.Model.synth(dev, cfg, seed=0)
    eng = rama_amd.Engine(dev, model)
    dev.sync()
    bytes_ = rama_amd.algorithmic_bytes(cfg)

    def run_steps(n, pos):
        """n decode steps from position pos; a generation that reaches seq_len is followed by a new
        one (BOS + prompt at position 0), so a run may be longer than the model's context"""
        while n > 0:
            if pos == seq:
                eng.decode_begin(1, 0, PROMPT)
                pos = 0
            m = min(n, seq - pos)
            eng.decode_steps(m)
            n -= m
            pos += m
        return pos

    eng.set_graph_mode(bool(args.graph))
    if args.mode == "parity":
        eng.set_tuning("ref_order", 1)
    pos0 = max(0, min(args.pos0, seq - 1))
    eng.decode_begin(1, pos0, PROMPT if pos0 == 0 else [])
    pos = run_steps(args.warmup, pos0)
    dev.sync()      # N = 1: no other rank to meet; everything runs on the context's stream, which this drains
    t0 = time.perf_counter()
    eng.timer_start()
    pos = run_steps(args.steps, pos)
    ev_ms = eng.timer_stop()
    dev.sync()
    wall_ms = (time.perf_counter() - t0) * 1e3
    ms_per_step = wall_ms / args.steps
    tokens = eng.decode_tokens()
    assert len(tokens) == (need if pos0 + need <= seq else pos), (len(tokens), need, pos)

    roofline, kernels = None, {}
    if not args.no_kprof:
        eng.set_graph_mode(False)   # per-launch event brackets need eager launches
        ksteps = 16
        kpos = min(pos, seq - ksteps)
        if kpos >= 0:
            for k in ("qkv", "attn", "wo", "w13", "w2", "cls"):
                eng.decode_begin(tokens[-1] if tokens else 1, kpos, [])
                avg_ms, n = eng.kprof(k, ksteps)
                b = bytes_.get(k)
                kernels[k] = {"avg_us": round(avg_ms * 1e3, 2), "launches": n,
                              "GBps": round(b / (avg_ms * 1e-3) / 1e9, 1) if b else None}
            a = kernels["w13"]["GBps"]
            traffic, traffic_src = pmc_traffic("gemv_rows<4, 2, 8, true, 5>") if args.config == "llama2-7B" else (None, None)
            kname = "gemv_rows_solo<4,CH,NORM,EPI_SWIGLU_PAIR>" if d <= 2048 else "gemv_rows<4,2,8,NORM,EPI_SWIGLU_PAIR>"
            roofline = {"bound": "hbm", "kernel": kname + " (rmsnorm + row-interleaved W1|W3 matvec + SiLU*gate)",
                        "achieved": a, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBPS, 4),
                        "traffic": traffic, "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": bytes_["w13"],
                        "avg_launch_us": kernels["w13"]["avg_us"]}

    cpu = None
    if not args.no_cpu_baseline and rank == 0:
        cpu = cpu_baseline(args.config, args.cpu_tokens, args.cpu_layers)

    tok_s = args.steps / (wall_ms * 1e-3)
    line = {
        "metric": "tokens/sec decode + matvec achieved-HBM-GB/s vs roofline, llama2-7B fp32 1xMI355X",
        "value": round(tok_s, 3), "unit": "tokens/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config} fp32 decode, weights resident in HBM, greedy, pos {pos0 + args.warmup}..{pos0 + need - 1}" + (" (wrapping at seq_len)" if pos0 + need > seq else ""),
                   "dim": d, "hidden_dim": h, "n_layers": L, "n_heads": H, "vocab_size": V, "seq_len": seq,
                   "sequences_in_flight": 1, "parallelism": "single GPU", "hipgraph": bool(args.graph),
                   "w13_layout": "row-interleaved copy per model (no placement tuning)"},
        "token_level": {"algorithmic_bytes_per_token": bytes_["token"],
                        "achieved_GBps": round(bytes_["token"] * tok_s / 1e9, 1),
                        "frac_of_8TBps": round(bytes_["token"] * tok_s / 1e9 / HBM_PEAK_GBPS, 4),
                        "event_ms_per_step": round(ev_ms / args.steps, 4)},
        "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu,
    }
    print(json.dumps(line), flush=True)
    eng.free(); model.free(); dev.close()


if __name__ == "__main__":
    main()
