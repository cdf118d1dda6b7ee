#!/usr/bin/env python3
"""bench.py -- decode throughput of the HIP path on llama2-7B-shaped synthetic fp32 weights.

  python bench.py --gpus 1 --steps K --warmup W
  python bench.py --gpus N ...          (N > 1 without a launcher: this script starts the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1: one decode step = one forward() (infer.rs:8-53) + greedy sample, chained on the device,
starting at pos 0 like generate() (BOS, then the Rama-BPE ids of 'once upon a time').  Two modes of
the SAME entry points are timed in one run:
  * parity mode (`value`): every op in the reference CPU path's own rounding order, on the model's
    chain-order weight copy (csrc/chain.hpp) -- logits bit-identical to engine/src/device/cpu.rs, the
    only way to stay within north_star's 1e-4 over a 200-token generation at llama2-7B depth (checked
    against the oracle in this run; 200 positions by tests/test_hip_parity_7b.py);
  * fast mode (`fast_mode`): fused multiply-adds and tree-shaped sums (csrc/kernels.hpp) -- closer to
    the exact logits than the reference itself, but up to 1.5e-4 from it at full depth.
`--mode tol` / `--mode all` adds the tolerance-mode experiment (chain-order matvecs with tree-summed
norms folded in and the fast attention: 0.72 of the roofline, but 1.4e-4 from the CPU path -- a sum of
squares in another order is enough to leave the bar; profiles/r04_tolerance_sweep_7b_200pos.jsonl).
Then stories15M / stories110M (BASELINE.json configs 2-3) in both modes (`other_configs`).
N > 1: the layer stack is pipeline-sharded over the N ranks with N sequences in flight; a step advances
every sequence by one token (rama_amd/pipeline.py, csrc/pipe.hip).

Prints ONE JSON line (see the driver contract): whole-job tokens/s, `roofline` for the dominant kernel
(W1|W3 SwiGLU matvec, 43 % of the bytes) and `cpu_baseline` (the oracle -- the C restatement of the
reference CPU path -- timed at full depth on the host cores, rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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
from rama_amd.sampler_const import TOPP_U_CPU as TOPP_U   # the reference re-seeds ChaCha20 (seed 100, cpu.rs:161-162) on every call: the draw is this constant (derived there)
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
METRIC_1GPU = "tokens/sec decode + matvec achieved-HBM-GB/s vs roofline, llama2-7B fp32 1xMI355X"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="llama2-7B", choices=list(SHAPES))
    ap.add_argument("--graph", type=int, default=1, help="replay each decode step from a hipGraph")
    ap.add_argument("--mode", default="both", choices=["both", "all", "tol", "parity", "fast"],
                    help="parity: every op in the reference's rounding order (bit-identical logits; the headline); fast: fused/tree sums; "
                         "tol: chain-order matvecs + folded tree-summed norms + fast attention (an experiment: 1.4e-4 from the CPU path at 7B); "
                         "both: parity + fast; all: the three of them")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="rama_set_tuning(KEY, VALUE) on every engine before timing (A/B runs under the profiler); repeatable")
    ap.add_argument("--settle-s", type=float, default=3.0,
                    help="seconds of untimed decoding over the same positions in front of the W warm-up steps (the part's clocks: r05_experiments.md 15); 0: none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kprof", action="store_true")
    ap.add_argument("--no-prefill", action="store_true", help="skip the prompt-ingestion (rama_prefill) figures")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the stories15M / stories110M lines")
    ap.add_argument("--no-sampled", action="store_true", help="skip the `-r 1` (top-p sampled) timings")
    ap.add_argument("--no-by-position", action="store_true", help="skip the timings at 200 / 1 000 / 1 900 positions of context")
    ap.add_argument("--no-generation-200", action="store_true", help="skip the README-length generation figures (positions 0..199: first use, cold, settled)")
    ap.add_argument("--no-trait-ops", action="store_true", help="skip the 1:1 Device-op path on tensor-by-tensor uploads")
    ap.add_argument("--no-placement-tuning", action="store_true", help="accepted and ignored (round-1 flag: the tuner is gone)")
    ap.add_argument("--pos0", type=int, default=0,
                    help="start the timed generation at this position over a pre-filled (zero) cache: long-context timing, "
                         "N = 1 only; the default 0 is generate()'s own start (BOS + prompt)")
    ap.add_argument("--cpu-tokens", type=int, default=16, help="tokens of the CPU baseline's sample (the first one warms up); every mode's logits are compared with the oracle's at each of them")
    ap.add_argument("--rank-timeout", type=float, default=900.0, help="seconds the self-started ranks of --gpus N may take")
    return ap.parse_args(argv)


LIB_SOURCES = ("rama_amd/csrc/*.hip", "rama_amd/csrc/*.hpp", "include/rama_hip.h")      # what librama_hip.so is compiled from


def source_hash():
    """first 16 hex digits of the SHA-256 over (relative path, contents) of every source file of librama_hip.so, in sorted order: the identity of a build that
    survives a rebuild (hipcc's output is not bit-reproducible: two builds of the same sources hash differently) and a fresh checkout"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for pat in LIB_SOURCES for f in glob.glob(str(REPO / pat)))
    for f in files:
        h.update(str(Path(f).relative_to(REPO)).encode() + b"\0")
        h.update(Path(f).read_bytes())
        h.update(b"\0")
    return h.hexdigest()[:16] if files else None


def library_stamp():
    """what is being run: `src_sha16` = source_hash() of the tree, `built_from_src_sha16` = the same as __graft_entry__.build() recorded it when it compiled
    the library that is loaded (rama_amd/BUILD_INFO.json; `stale_build` when the two differ: sources edited without a rebuild), the git head at that build
    (informational: the GPU box has no .git)"""
    out = {"src_sha16": source_hash(), "built_from_src_sha16": None, "git_head_at_build": None}
    try:
        info = json.loads((REPO / "rama_amd" / "BUILD_INFO.json").read_text())
        out["built_from_src_sha16"] = info.get("src_sha16")
        out["git_head_at_build"] = info.get("git_head")
    except (OSError, ValueError):
        pass
    if out["built_from_src_sha16"] != out["src_sha16"]:
        out["stale_build"] = True
    return out


def profile_stamp(path):
    """the library a committed profile was collected on (tools/collect_profiles.py and the parity test write `library` into their JSON; a CSV has a
    .meta.json beside it): -> src_sha16 or None (profiles of rounds 1-5 carry no stamp)"""
    try:
        pth = Path(path)
        if pth.suffix == ".json":
            j = json.loads(pth.read_text())
        else:
            j = json.loads(pth.with_suffix(".meta.json").read_text())
        return (j.get("library") or {}).get("src_sha16")
    except (OSError, ValueError, AttributeError):
        return None


def pmc_traffic(kernel_substr):
    """HBM read bytes per launch of the dominant kernel from the committed PMC pass (a run of
    its own: rocprofv3 --pmc FETCH_SIZE cannot ride along with a timed run), already carrying
    the guide's gfx950 correction (FETCH_SIZE counts 128-B requests at 64 B: x2)."""
    import glob
    files = sorted(glob.glob(str(REPO / "profiles" / "r*_bench_7b*_pmc_fetch_size.json")))
    for f in reversed(files):
        with open(f) as fh:
            rows = json.load(fh)["rows"]
        for r in rows:
            if kernel_substr in r["kernel"] and r["counter"] == "FETCH_SIZE":
                return r["hbm_read_bytes_corrected"], f"profiles/{Path(f).name}"
    return None, None


def pmc_token_traffic(shape, mode):
    """HBM read bytes per TOKEN (every launch of the decode loop) from the committed PMC pass of a small shape
    (tools/collect_profiles.py --config <shape>), gfx950 correction applied; None when there is no such pass."""
    import glob
    suffix = {"parity": "_parity", "tol": "_tol"}.get(mode, "")
    files = sorted(glob.glob(str(REPO / "profiles" / f"r*_bench_{shape}{suffix}_pmc_fetch_size.json")))
    for f in reversed(files):
        with open(f) as fh:
            v = json.load(fh).get("hbm_read_bytes_per_token_corrected")
        if v:
            return v, f"profiles/{Path(f).name}"
    return None, None


def _mem_available_gb() -> float:
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def cpu_baseline(shape_name, n_tokens, check_engines=None):
    """Time the oracle (reference-algorithm CPU restatement, OpenMP over rows/heads like the reference's
    rayon) at FULL depth on the same synthetic weights and token positions (BOS + prompt from position 0).
    With `check_engines` ({mode: rama_amd.Engine set to that mode}) the same positions run on the GPU in
    every mode and the logits are compared with the oracle's -- the oracle as the checker, never as the
    thing measured.  llama2-7B needs ~30 GB of host memory for the weights; a box without it gets an
    8-layer sample scaled by 4 (said so in `sample`)."""
    import numpy as np
    from oracle import oracle as O
    from oracle import synth as S
    check_engines = check_engines or {}
    d, h, L, H, V, seq, shared = SHAPES[shape_name]
    full = O.Config(d, h, L, H, H, V, seq, shared)
    n_tokens = min(n_tokens, seq)
    cseq = min(seq, max(64, n_tokens))        # the caches are sized for the sample's positions
    need_gb = 4e-9 * (L * (4 * d * d + 3 * d * h) + (1 if shared else 2) * V * d + 2 * L * cseq * d) + 2.0
    ls = L if _mem_available_gb() > need_gb + 4.0 else min(8, L)
    cfg = O.Config(d, h, ls, H, H, V, cseq, shared)
    spec = S.synth_spec(full)
    w = {}
    for name, shp in O.weight_shapes(cfg):
        if name.startswith("freq_cis"):
            continue
        tag, scale, bias = spec[name]
        w[name] = O.fill_synth(int(np.prod(shp)), 0, tag, scale, bias).reshape(shp)
    any_engine = next(iter(check_engines.values()), None)
    if any_engine is not None:      # the checkpoint's own RoPE tables (the first rows of the resident model's), so both sides read the same weights
        nrope = cfg.seq_len * (cfg.head_size // 2)
        w["freq_cis_real"] = any_engine.model.tensor("freq_cis_real", nrope).reshape(cfg.seq_len, -1)
        w["freq_cis_imag"] = any_engine.model.tensor("freq_cis_imag", nrope).reshape(cfg.seq_len, -1)
    else:
        w["freq_cis_real"], w["freq_cis_imag"] = S.rope_tables(cfg.seq_len, cfg.head_size)
    try:   # the GPU box shows 256 CPUs but a 1-GPU job owns a 16-CPU share
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    orc = O.Oracle(cfg, w, threads=max(1, min(16, avail)))
    toks = [1] + PROMPT
    t_layers = t_cls = 0.0
    token = 1
    chk = {m: {"checked_positions": 0, "worst_vs_oracle": 0.0, "bit_identical_to_oracle": True, "greedy_tokens_equal_oracle": True}
           for m in check_engines}
    for pos in range(n_tokens):
        t0 = time.perf_counter()
        orc.forward_range(token, pos, 0, ls, True, False)
        t1 = time.perf_counter()
        orc.forward_range(token, pos, ls, ls, False, True)
        t2 = time.perf_counter()
        if pos > 0:   # the first token pages the weights in
            t_layers += t1 - t0
            t_cls += t2 - t1
        lo = orc.s["logits"]
        nxt = O.argmax(lo)
        if ls == L:
            for m, eng in check_engines.items():
                eng.set_tuning("ref_order", REF_ORDER[m])
                try:
                    eng.forward(token, pos)
                    lg = eng.logits()
                finally:
                    eng.set_tuning("ref_order", 0)
                c = chk[m]
                c["worst_vs_oracle"] = max(c["worst_vs_oracle"], float(np.abs(lg - lo).max()))
                c["bit_identical_to_oracle"] = c["bit_identical_to_oracle"] and bool(np.array_equal(lg.view(np.uint32), lo.view(np.uint32)))
                c["greedy_tokens_equal_oracle"] = c["greedy_tokens_equal_oracle"] and int(np.flatnonzero(lg == lg.max())[-1]) == nxt
                c["checked_positions"] += 1
        token = toks[pos + 1] if pos + 1 < len(toks) else nxt
    n = max(n_tokens - 1, 1)
    per_token = (t_layers / n) * (L / ls) + t_cls / n
    threads = O.lib().oracle_get_threads()
    out = {"value": round(1.0 / per_token, 4), "unit": "tokens/s", "cores": threads, "kind": "port",
           "sample": f"{shape_name} shape, " + (f"all {L} layers" if ls == L else f"{ls} of {L} layers (layer time scaled x{L / ls:g})")
                     + f" + classifier, {n} tokens after 1 warm-up = positions 1-{n} (BOS + prompt from position 0: the short-context best case of the CPU path too); oracle/rama_oracle.c "
                     f"(C restatement of engine/src/device/cpu.rs), OpenMP threads={threads}"}
    return out, {m: c for m, c in chk.items() if c["checked_positions"]}


def time_decode(eng, dev, seq, steps, warmup, pos0, prompt, temperature=0.0, settle_s=0.0):
    """`warmup` untimed + `steps` timed chained decode steps; a generation that reaches seq_len is followed
    by a new one (BOS + prompt at position 0), so a run may be longer than the model's context.
    temperature != 0: Device::sample's top-p path on the device (topp 0.9, the CPU backend's constant draw)"""
    eng.decode_sampler(temperature, 0.9, TOPP_U if temperature != 0.0 else 0.0)
    def run_steps(n, pos):
        while n > 0:
            if pos == seq:
                eng.decode_begin(1, 0, prompt)
                pos = 0
            m = min(n, seq - pos)
            eng.decode_steps(m)
            n -= m
            pos += m
        return pos
    # the part's clocks first: a process that has just uploaded a model finds the GPU in a state in which every matvec runs ~2 % slower for the first second
    # or so of decoding (profiles/r05_experiments.md 15: 204-205 tok/s in some runs, 209 after three or more untimed passes, in every run).  Whole untimed
    # passes over the SAME positions until `settle_s` seconds of decoding have gone by; the W warm-up steps and the K timed ones then follow as ever.
    t_settle = time.perf_counter()
    while settle_s > 0 and time.perf_counter() - t_settle < settle_s:
        eng.decode_begin(1, pos0, prompt if pos0 == 0 else [])
        run_steps(warmup + steps, pos0)
        dev.sync()
        eng.decode_tokens()
    eng.decode_begin(1, pos0, prompt if pos0 == 0 else [])
    pos = run_steps(warmup, pos0)
    dev.sync()      # N = 1: no other rank to meet; everything runs on the context's stream, which this drains
    t0 = time.perf_counter()
    eng.timer_start()
    pos = run_steps(steps, pos)
    ev_ms = eng.timer_stop()
    dev.sync()
    wall_ms = (time.perf_counter() - t0) * 1e3
    tokens = eng.decode_tokens()
    need = warmup + steps
    assert len(tokens) == (need if pos0 + need <= seq else pos), (len(tokens), need, pos)
    return wall_ms, ev_ms, pos, tokens


def time_prefill(dev, model, mode, n_positions, V):
    """prompt positions per second of rama_prefill (the forced positions of generate(), mod.rs:187-194, as token
    batches: fp32 MFMA GEMMs, 128 positions per weight pass, in fast mode; chain-order kernels in the reference's
    rounding order, 16 per pass, in parity mode), best of 3 after one warm-up; reported beside `value`, never as it"""
    import ctypes as C
    import numpy as np
    import rama_amd
    from rama_amd._lib import check
    eng = rama_amd.Engine(dev, model)
    eng.set_tuning("ref_order", REF_ORDER[mode])
    toks = [1] + [int(v) for v in np.random.default_rng(0).integers(2, V, n_positions - 1)]
    arr = (C.c_int32 * n_positions)(*toks)
    best = 1e9
    for i in range(4):
        t0 = time.perf_counter()
        check(dev.lib.rama_prefill(dev.ctx, C.byref(model.ccfg), C.byref(model.weights), C.byref(eng.state), arr, n_positions, 0))
        dev.sync()
        if i:
            best = min(best, time.perf_counter() - t0)
    eng.set_tuning("ref_order", 0)
    eng.free()
    return {"positions": n_positions, "ms": round(best * 1e3, 2), "prompt_tok_s": round(n_positions / best, 1)}


def deep_context_check(dev, name, modes, graph, points):
    """the launch configurations `by_position` times, against the ORACLE at those positions: the shape's width with 2 of its layers (the oracle then needs
    1.6 GB at llama2-7B instead of 27), both caches filled with the same random rows on both sides, one forward() per point and mode.
    {mode: {p: {bit_identical_to_oracle, worst_vs_oracle}}} -- parity mode must be bit-identical; bar / fast are 2-layer distances (the full-depth
    ones over the whole context: profiles/r06_tolerance_sweep_7b_2048pos*.jsonl).  Part of the cpu_baseline leg: the oracle as the checker."""
    import numpy as np
    import rama_amd
    from oracle import oracle as O
    from oracle import synth as S
    d, h, L, H, V, seq, shared = SHAPES[name]
    L2, V2 = min(L, 2), min(V, 640)
    cfg = O.Config(d, h, L2, H, H, V2, seq, False)
    rope = S.rope_tables(seq, d // H)
    w = S.synth_weights(cfg, 13, rope=rope)
    orc = O.Oracle(cfg, w)
    model = rama_amd.Model.synth(dev, rama_amd.Config(d, h, L2, H, H, V2, seq, False), 13, rope=rope)
    rng = np.random.default_rng(13)
    kc = rng.standard_normal(L2 * seq * d, dtype=np.float32)
    vc = rng.standard_normal(L2 * seq * d, dtype=np.float32)
    out = {m: {} for m in modes}
    engs = {m: rama_amd.Engine(dev, model) for m in modes}
    try:
        for p in points:
            pos = min(p, seq - 1)
            orc.s["key_cache"][:] = kc; orc.s["value_cache"][:] = vc
            lo = orc.forward(7, pos).copy()
            for m, eng in engs.items():
                eng.set_tuning("ref_order", REF_ORDER[m])
                for k_, v_ in TUNE:
                    eng.set_tuning(k_, v_)
                eng.set_graph_mode(bool(graph))
                try:
                    eng.set_buffer("key_cache", kc); eng.set_buffer("value_cache", vc)
                    eng.forward(7, pos)
                    lg = eng.logits()
                finally:
                    eng.set_graph_mode(False)
                    eng.set_tuning("ref_order", 0)
                out[m][str(p)] = {"bit_identical_to_oracle": bool(np.array_equal(lg.view(np.uint32), lo.view(np.uint32))),
                                  "worst_vs_oracle": float(np.abs(lg - lo).max())}
    finally:
        for e in engs.values():
            e.free()
        model.free()
    return {"sample": f"{name} width, {L2} of {L} layers, vocabulary {V2}, both caches filled with the same random rows, one forward() per point vs oracle/rama_oracle.c",
            "modes": out}


def generation_200(dev, model, modes, graph, seq, prompt, n=200, cold=True):
    """The README's own bench is a 200-token generation (README.md:80-83): tokens/s over positions 0..n-1 of ONE generation -- BOS + the prompt, then greedy --
    on a run state created for it: `first_use` (the engine's first generation: graph captures and lazily made weight copies inside), `cold` (the same engine again,
    at once -- with cold=True this is called right behind the model's upload, before anything else has run: what a process that has just uploaded its model
    sees) and `settled` (after ~2 s of untimed decoding: the part's clocks, profiles/r05_experiments.md 15).  {mode: {first_use, cold, settled: tok_s}}"""
    import rama_amd
    n = min(n, seq)
    out = {}
    for mode in modes:
        eng = rama_amd.Engine(dev, model)
        eng.set_tuning("ref_order", REF_ORDER[mode])
        for k_, v_ in TUNE:
            eng.set_tuning(k_, v_)
        eng.set_graph_mode(bool(graph))
        eng.decode_sampler(0.0)

        def once():
            eng.decode_begin(1, 0, prompt)
            dev.sync()
            t0 = time.perf_counter()
            eng.decode_steps(n)
            dev.sync()
            dt = time.perf_counter() - t0
            eng.decode_tokens()
            return round(n / dt, 2)
        r = {"first_use": once(), "cold": once()} if cold else {}
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < 2.0:
            once()
        r["settled"] = max(once(), once())
        r["positions"] = f"0..{n - 1}"
        eng.set_tuning("ref_order", 0)
        eng.free()
        out[mode] = r
    return out


def by_position(dev, model, modes, graph, seq, points=(200, 1000, 1900), steps=32, warmup=4):
    """tokens/s with the context already `p` positions long (README.md:80-83 generates 200 tokens; the cache holds 2 048): the attention of
    cpu.rs:23-52 grows with the position, the matvecs do not.  The cache rows in front are zeros -- the launches and their traffic are those of
    a real context.  {mode: {p: {tok_s, ms_per_step, positions}}}"""
    import rama_amd
    out = {}
    for mode in modes:
        eng = rama_amd.Engine(dev, model)
        eng.set_tuning("ref_order", REF_ORDER[mode])
        for k_, v_ in TUNE:
            eng.set_tuning(k_, v_)
        eng.set_graph_mode(bool(graph))
        d = {}
        for p in points:
            pos0 = max(0, min(p, seq - steps) - warmup)
            wall_ms, _, _, _ = time_decode(eng, dev, seq, steps, warmup, pos0, PROMPT if pos0 == 0 else [])
            d[str(p)] = {"tok_s": round(steps / (wall_ms * 1e-3), 2), "ms_per_step": round(wall_ms / steps, 4),
                         "positions": f"{pos0 + warmup}..{pos0 + warmup + steps - 1}"}
        eng.set_tuning("ref_order", 0)
        eng.free()
        out[mode] = d
    return out


def trait_ops_path(dev, name, model, modes, graph, n_fwd=8):
    """The boundary as the reference drives it: weights uploaded TENSOR BY TENSOR (hbm.rs:55-90; here filled on the device with the
    resident model's values), then per token either forward() composed from the 1:1 Device ops (infer.rs:8-53 op for op, ~1 700 calls per token
    through ctypes at llama2-7B: 32 apply_position calls per layer) or the fused entry rama_forward on the same tensors -- which adopts them and
    runs the resident model's kernels.  Host-driven loops, no sampling; logits compared bit for bit with the resident model's."""
    import numpy as np
    import rama_amd
    from rama_amd._lib import check
    d, h, L, H, V, seq, shared = SHAPES[name]
    cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
    ws = rama_amd.TransformerWeights.synth(cfg, 0, dev)
    wv = rama_amd.TransformerWeightsView.from_gpu_ws(ws)
    rs = rama_amd.RunState.from_config(cfg, dev)
    rsv = rama_amd.RunStateView.from_rs(rs)
    ref = rama_amd.Engine(dev, model)
    out = {"calls_per_token_1to1": 2 + L * (17 + H) + 3, "host": "python/ctypes"}
    try:
        for mode in modes:
            check(dev.lib.rama_set_tuning(dev.ctx, b"ref_order", REF_ORDER[mode]))
            for k_, v_ in TUNE:
                check(dev.lib.rama_set_tuning(dev.ctx, k_.encode(), v_))
            same, worst = True, 0.0
            for path, fwd in (("ops", rama_amd.forward), ("fused_entry", rama_amd.forward_fused)):
                check(dev.lib.rama_set_graph_mode(dev.ctx, 1 if (graph and path == "fused_entry") else 0))
                token = 1
                for pos in range(3):      # warm-up (the chain-order copies are made here) and the check against the resident model
                    fwd(cfg, wv, rsv, token, pos, dev)
                    ref.forward(token, pos)
                    lg = dev.download(rsv.logits)
                    rl = ref.logits()
                    same = same and np.array_equal(lg.view(np.uint32), rl.view(np.uint32))
                    worst = max(worst, float(np.abs(lg - rl).max()))
                    token = int(np.flatnonzero(lg == lg.max())[-1])
                dev.sync()
                t0 = time.perf_counter()
                for pos in range(3, 3 + n_fwd):
                    fwd(cfg, wv, rsv, token, pos, dev)
                dev.sync()
                out[f"{mode}_{path}_tok_s"] = round(n_fwd / (time.perf_counter() - t0), 2)
            # (parity mode: the same bits whatever the path; fast mode: W1 and W3 as two tensors sum in another order than the resident model's
            # row-interleaved copy -- a few 1e-6 apart)
            out[f"{mode}_logits_bit_identical_to_resident_model"] = bool(same)
            out[f"{mode}_worst_vs_resident_model"] = worst
            check(dev.lib.rama_set_graph_mode(dev.ctx, 0))
    finally:
        check(dev.lib.rama_set_tuning(dev.ctx, b"ref_order", 0))
        ref.free(); rs.free(); ws.free()
    return out


def kernel_times(eng, cfg_seq, pos, tokens, bytes_, ksteps=16):
    """per-launch device time of every kernel class over `ksteps` eager decode steps (events carried by the dispatch;
    `sample`: event records around Device::sample's launches); us_per_step = avg_us x launches per step, so the classes
    add up to the step"""
    kernels = {}
    kpos = min(pos, cfg_seq - ksteps)
    if kpos < 0:
        return kernels
    for k in ("qkv", "attn", "wo", "w13", "w2", "cls", "norm", "sample"):
        eng.decode_begin(tokens[-1] if tokens else 1, kpos, [])
        avg_ms, n = eng.kprof(k, ksteps)
        b = bytes_.get(k)
        if n:
            kernels[k] = {"avg_us": round(avg_ms * 1e3, 2), "launches": n, "us_per_step": round(avg_ms * 1e3 * n / ksteps, 1),
                          "GBps": round(b / (avg_ms * 1e-3) / 1e9, 1) if b else None}
    return kernels


TUNE = []      # (key, value) pairs of --tune
REF_ORDER = {"fast": 0, "parity": 1, "tol": 2, "bar": 3}      # rama_set_tuning("ref_order", .)
MODE_TEXT = {
    "tol": "tolerance (experiment): chain-order matvecs in the reference CPU path's rounding order (cpu.rs:127-153), rmsnorm sums tree-shaped and folded "
           "into them, the fast path's attention -- 1.4e-4 from cpu.rs over 200 full-depth positions",
    "parity": "parity: every op in the reference CPU path's rounding order (chain-order weight copy), logits bit-identical to the oracle = cpu.rs with its two "
              "crate-owned summation orders fixed (wide::f32x4::reduce_add pairwise -- switchable: \"lane_reduce\"; rayon's softmax sum as ONE front-to-back sum); the "
              "reference's other admissible executions sit ~1e-4 from this one at this depth (profiles/r06_reference_self_spread.json)",
    "bar": "bar: parity mode up to position 127, the fast path's attention from 128 on (exact matvecs and norms) -- not bit-identical behind the switch, measured "
           "<= 1e-4 from the oracle over the whole 2 048-position context (profiles/r06_tolerance_sweep_7b_2048pos*.jsonl)",
    "fast": "fast: fused multiply-adds, tree-shaped sums",
}


def run_shape(dev, name, steps, warmup, pos0, graph, modes, kprof, sampled=False, settle_s=0.0, gen_head=None):
    """the requested modes of one shape on one resident model -> (cfg, model, bytes, {mode: {...}});
    sampled: also time the head mode with the README's `-r 1` (Device::sample's top-p path, on the device)"""
    import rama_amd
    d, h, L, H, V, seq, shared = SHAPES[name]
    cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
    model = rama_amd.Model.synth(dev, cfg, seed=0)
    bytes_ = rama_amd.algorithmic_bytes(cfg)
    out = {}
    pos0 = max(0, min(pos0, seq - 1))
    gen200 = {}
    if gen_head:      # the README's 200-token generation of the headline mode FIRST: its `cold` figure wants a part that has done nothing but upload the model
        gen200.update(generation_200(dev, model, [gen_head], graph, seq, PROMPT, cold=True))
        gen200.update(generation_200(dev, model, [m_ for m_ in modes if m_ != gen_head], graph, seq, PROMPT, cold=False))
    for mode in modes:
        eng = rama_amd.Engine(dev, model)
        eng.set_tuning("ref_order", REF_ORDER[mode])
        for k_, v_ in TUNE:
            eng.set_tuning(k_, v_)
        eng.set_graph_mode(bool(graph))
        wall_ms, ev_ms, pos, tokens = time_decode(eng, dev, seq, steps, warmup, pos0, PROMPT, settle_s=settle_s)
        tok_s = steps / (wall_ms * 1e-3)
        r = {"tok_s": round(tok_s, 3), "ms_per_step": round(wall_ms / steps, 4), "event_ms_per_step": round(ev_ms / steps, 4),
             "achieved_GBps": round(bytes_["token"] * tok_s / 1e9, 1),
             "frac_of_8TBps": round(bytes_["token"] * tok_s / 1e9 / HBM_PEAK_GBPS, 4),
             "positions": f"{pos0 + warmup}..{pos0 + warmup + steps - 1}" + (" (wrapping at seq_len)" if pos0 + warmup + steps > seq else ""),
             "tokens": tokens}
        if sampled and mode in sampled:
            # BASELINE config 3 as the README ran it: `-r 1` (README.md:80-83) = temperature 1, topp 0.9 (main.rs:48-50), Device::sample
            # (cpu.rs:155-179) on the device inside the chained loop.  Synthetic logits are flat: every entry is a top-p candidate, the worst case.
            s_wall, _, _, _ = time_decode(eng, dev, seq, steps, warmup, pos0, PROMPT, temperature=1.0)
            eng.decode_sampler(0.0)
            r["sampled_r1"] = {"tok_s": round(steps / (s_wall * 1e-3), 3), "ms_per_step": round(s_wall / steps, 4), "temperature": 1.0, "topp": 0.9,
                               "sampler": "device top-p (csrc/topp_sort.hpp), the CPU backend's constant draw u = %.7f" % TOPP_U}
        if kprof:
            eng.set_graph_mode(False)   # per-launch event brackets need eager launches
            r["kernels"] = kernel_times(eng, seq, pos, tokens, bytes_)
            r["kernels_sum_ms_per_step"] = round(sum(k["us_per_step"] for k in r["kernels"].values()) * 1e-3, 4)
        eng.set_tuning("ref_order", 0)
        eng.free()
        if mode in gen200:
            r["generation_200"] = gen200[mode]
        out[mode] = r
    return cfg, model, bytes_, out


def roofline_of(kernels, bytes_, mode, d):
    if not kernels or "w13" not in kernels:
        return None
    a = kernels["w13"]["GBps"]
    if mode == "tol":
        kname = "gemv_chain_kernel<W,D,XD,CEPI_SWIGLU,CNORM_TREE> (tree-summed rmsnorm + chain-order W1|W3 matvec in the reference's rounding order + SiLU*gate)"
        traffic, traffic_src = pmc_traffic("gemv_chain_kernel<1, 16, 4, 3, 2>") if d == 4096 else (None, None)
    elif mode == "parity":
        kname = "gemv_chain_kernel<W,D,XD,CEPI_SWIGLU> (chain-order W1|W3 matvec in the reference's rounding order + SiLU*gate)"
        kname = "gemv_chain_kernel<W,D,XD,CEPI_SWIGLU,CNORM_LEAD> (the FFN norm's exact sum by a leader workgroup of the launch + chain-order W1|W3 matvec in the reference's rounding order + SiLU*gate)"
        traffic, traffic_src = pmc_traffic("gemv_chain_kernel<1, 16, 4, 3, 3, 64") if d == 4096 else (None, None)      # [r5] the leader-norm instantiation
        if traffic is None and d == 4096:
            kname = "gemv_chain_kernel<W,D,XD,CEPI_SWIGLU> (chain-order W1|W3 matvec in the reference's rounding order + SiLU*gate)"
            traffic, traffic_src = pmc_traffic("gemv_chain_kernel<1, 16, 4, 3, 0>")
        if traffic is None and d == 4096:
            traffic, traffic_src = pmc_traffic("gemv_chain_kernel<1, 16, 4, 3>")
    else:
        kname = ("gemv_rows_solo<4,CH,NORM,EPI_SWIGLU_PAIR>" if d <= 2048 else "gemv_rows<4,2,8,NORM,EPI_SWIGLU_PAIR>") + " (rmsnorm + row-interleaved W1|W3 matvec + SiLU*gate)"
        traffic, traffic_src = pmc_traffic("gemv_rows<4, 2, 8, true, 5>") if d == 4096 else (None, None)
    return {"bound": "hbm", "kernel": kname, "achieved": a, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBPS, 4),
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": bytes_["w13"], "avg_launch_us": kernels["w13"]["avg_us"]}


def parity_200pos():
    """the committed per-position record of tests/test_hip_parity_7b.py (llama2-7B, all 32 layers, 200 positions): what the in-run
    check below samples at a few positions, over the README's whole generation length"""
    import glob
    files = sorted(glob.glob(str(REPO / "profiles" / "r*_parity_llama2_7b_200pos.json")))
    for f in reversed(files):
        with open(f) as fh:
            j = json.load(fh)
        if "worst_hip_tolerance_vs_oracle" in j:
            return {"source": f"profiles/{Path(f).name}", "positions": j["positions"],
                    "tol": j["worst_hip_tolerance_vs_oracle"], "parity": j["worst_hip_ref_order_vs_oracle"], "fast": j["worst_hip_vs_oracle"]}
    return None


def bar_whole_context():
    """the committed full-depth record of bar mode over the whole context (tools/tol_sweep.py 2048 llama2-7B '' bar): worst |dlogit| vs the oracle per 512 positions"""
    import glob
    for f in reversed(sorted(glob.glob(str(REPO / "profiles" / "r*_tol_curve_7b_2048pos_bar.json")))):
        try:
            j = json.loads(Path(f).read_text())
            for sm in j["summaries"]:
                if sm["config"] == "bar":
                    return {"source": f"profiles/{Path(f).name}", "positions": sm["positions"], "worst_vs_oracle": sm["worst_vs_oracle"], "positions_over_1e-4": sm["positions_over_1e-4"],
                            "greedy_tokens_equal": sm["greedy_tokens_equal"], "worst_by_512": sm["worst_by_512"], "library": (j.get("library") or {}).get("src_sha16")}
        except (OSError, ValueError, KeyError):
            continue
    return None


def single_gpu(args, local_rank):
    import rama_amd
    TUNE[:] = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in args.tune]
    dev = rama_amd.Hip(local_rank)
    modes = {"all": ["fast", "tol", "parity"], "both": ["fast", "parity"]}.get(args.mode, [args.mode])
    head = "parity" if "parity" in modes else modes[-1]
    cfg, model, bytes_, res = run_shape(dev, args.config, args.steps, args.warmup, args.pos0, args.graph, modes, not args.no_kprof,
                                        sampled=() if args.no_sampled else (head,), settle_s=args.settle_s,
                                        gen_head=None if (args.no_generation_200 or args.pos0) else head)
    d, h, L, H, V, seq, shared = SHAPES[args.config]
    for m_ in modes:
        if m_ != "fast" and "fast" in res:
            res[m_]["greedy_tokens_equal_fast_mode"] = res[m_]["tokens"] == res["fast"]["tokens"]

    def baseline_for(name, mdl, n_tokens):
        engines = {m_: rama_amd.Engine(dev, mdl) for m_ in modes}
        try:
            return cpu_baseline(name, n_tokens, engines)
        finally:
            for e in engines.values():
                e.free()

    prefill = None
    if not args.no_prefill:
        prefill = {m_: time_prefill(dev, model, m_, min(256, seq), V) for m_ in modes}
    cpu, check = (None, {}) if args.no_cpu_baseline else baseline_for(args.config, model, args.cpu_tokens)
    pos_modes = modes + (["bar"] if ("parity" in modes and "bar" not in modes) else [])      # bar mode = parity mode below position 128: it only shows at depth
    bypos = None if (args.no_by_position or seq < 512) else by_position(dev, model, pos_modes, args.graph, seq)
    bar0 = None
    if bypos and "bar" in pos_modes and not args.pos0:
        # bar mode with its switch at position 0 = exact matvecs + exact norms + the fast attention at EVERY position: the configuration the whole-context
        # sweep measured (<= 1e-4 over 2 048 positions, worst 9.75e-5; profiles/r06_tolerance_sweep_7b_2048pos.jsonl "tol+64"), at the headline's positions
        eng = rama_amd.Engine(dev, model)
        eng.set_tuning("ref_order", 3); eng.set_tuning("bar_pos", 0)
        for k_, v_ in TUNE:
            eng.set_tuning(k_, v_)
        eng.set_graph_mode(bool(args.graph))
        try:
            w_ms, _, _, toks0 = time_decode(eng, dev, seq, args.steps, args.warmup, 0, PROMPT, settle_s=min(args.settle_s, 1.0))
        finally:
            eng.set_tuning("bar_pos", 128); eng.set_tuning("ref_order", 0)
            eng.free()
        t0_ = args.steps / (w_ms * 1e-3)
        bar0 = {"tok_s": round(t0_, 3), "ms_per_step": round(w_ms / args.steps, 4), "frac_of_8TBps": round(bytes_["token"] * t0_ / 1e9 / HBM_PEAK_GBPS, 4),
                "positions": res[head]["positions"], "greedy_tokens_equal_parity_mode": toks0 == res["parity"]["tokens"] if "parity" in res else None,
                "what": "\"bar_pos\" = 0: the fast attention at every position, matvecs and norms exact -- within 1e-4 of the oracle over the whole context "
                        "(worst 9.75e-5 over 2 048 positions), NOT bit-identical; the reference's own admissible executions sit ~7e-5 apart (profiles/r06_reference_self_spread.json)",
                "evidence": "profiles/r06_tolerance_sweep_7b_2048pos.jsonl (configuration tol+64: the same arithmetic)"}
    deep = None
    if bypos and not args.no_cpu_baseline:
        deep = deep_context_check(dev, args.config, pos_modes, args.graph, (200, 1000, 1900))
        for m_ in pos_modes:
            for p_, v_ in deep["modes"][m_].items():
                bypos[m_][p_].update(v_)
    trait = None if args.no_trait_ops else trait_ops_path(dev, args.config, model, modes, args.graph)
    model.free()

    others = {}
    if not args.no_other_configs:
        for name in SHAPES:
            if name == args.config:
                continue
            oseq = SHAPES[name][5]
            osteps, owarm = min(200, oseq - 28), 28            # the README's 200-token generation (positions 28..227)
            _, om, ob, orr = run_shape(dev, name, osteps, owarm, 0, args.graph, modes, False, sampled=() if args.no_sampled else modes)
            # the oracle over the README's 200 positions: the CPU baseline's sample AND the checker of every mode's logits
            ocpu, ocheck = (None, {}) if args.no_cpu_baseline else baseline_for(name, om, min(200, oseq))
            om.free()
            # `tok_s` = the FASTEST mode whose logits stayed within north_star's 1e-4 of the oracle at every one of the README's 200 positions IN THIS RUN
            # (at these shallow shapes that is fast mode: ~1e-6); without the oracle leg the bit-identical mode stands
            ok_modes = [m_ for m_ in orr if m_ in ocheck and ocheck[m_]["checked_positions"] >= min(200, oseq) and ocheck[m_]["worst_vs_oracle"] <= 1e-4]
            best = max(ok_modes, key=lambda m_: orr[m_]["tok_s"]) if ok_modes else head
            od, oh, oL = SHAPES[name][0], SHAPES[name][1], SHAPES[name][2]
            phases = 5 * oL + 1      # dependent phases of a batch-1 token: per layer Wq|Wk|Wv -> attention -> Wo -> W1|W3 -> W2, then the classifier
            floor_us = phases * 1.3 + ob["token"] / 6.3e6      # 1.3 us per in-launch hand-off (profiles/r03_fused_stage.txt), the bytes at the measured 6.3 TB/s copy ceiling
            entry = {"steps": osteps, "warmup": owarm, "algorithmic_bytes_per_token": ob["token"],
                     "tok_s": orr[best]["tok_s"], "frac_of_8TBps": orr[best]["frac_of_8TBps"], "mode": best,
                     "mode_chosen_by": "fastest mode within 1e-4 of the oracle at all %d checked positions of this run" % min(200, oseq) if ok_modes else "no oracle leg in this run: the bit-identical mode",
                     "floor_tok_s": round(1e6 / floor_us, 1), "frac_of_floor": round(orr[best]["tok_s"] * floor_us / 1e6, 3),
                     "floor_model": f"batch-1 floor: {phases} dependent phases x 1.3 us hand-off + {ob['token']} B / 6.3 TB/s = {floor_us:.1f} us per token "
                                    "(a token is a chain of phases that each need the previous one's whole output; the 8 TB/s figure assumes none of that)"}
            for m_, r in orr.items():
                entry[m_ + "_mode"] = {k: v for k, v in r.items() if k not in ("tokens", "kernels")}
                tr, src = pmc_token_traffic(name, m_)
                entry[m_ + "_mode"]["traffic_per_token"] = tr           # whole-token HBM reads (PMC pass of its own), vs algorithmic_bytes_per_token
                entry[m_ + "_mode"]["traffic_source"] = src
                if m_ in ocheck:
                    entry[m_ + "_mode"].update(ocheck[m_])
            entry["cpu_baseline"] = ocpu
            others[name] = entry

    r = res[head]
    lib = library_stamp()
    line = {
        "metric": METRIC_1GPU,
        "value": r["tok_s"], "unit": "tokens/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "settle_s": args.settle_s,
        "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config} fp32 decode, weights resident in HBM, greedy, pos {r['positions']}",
                   "mode": MODE_TEXT[head],
                   "dim": d, "hidden_dim": h, "n_layers": L, "n_heads": H, "vocab_size": V, "seq_len": seq,
                   "sequences_in_flight": 1, "parallelism": "single GPU", "hipgraph": bool(args.graph)},
        "token_level": {"algorithmic_bytes_per_token": bytes_["token"], "achieved_GBps": r["achieved_GBps"],
                        "frac_of_8TBps": r["frac_of_8TBps"], "event_ms_per_step": r["event_ms_per_step"]},
        "roofline": roofline_of(r.get("kernels"), bytes_, head, d), "kernels": r.get("kernels", {}),
        "kernels_sum_ms_per_step": r.get("kernels_sum_ms_per_step"),
        "cpu_baseline": cpu,
        "library": lib,
    }
    if r.get("generation_200"):
        line["generation_200"] = r["generation_200"]      # the headline mode over the README's whole generation (positions 0..199): first use, cold, settled
    full = parity_200pos() if args.config == "llama2-7B" else None
    for m_ in modes:
        block = {k: v for k, v in res[m_].items() if k != "tokens"}
        if m_ in check:
            block.update(check[m_])
        if full:
            block["worst_vs_oracle_200_positions"] = full[m_]
            block["worst_vs_oracle_200_positions_source"] = full["source"]
        if m_ != head:
            block["roofline"] = roofline_of(block.get("kernels"), bytes_, m_, d)
        line[("tolerance" if m_ == "tol" else m_) + "_mode"] = block
    if head in check:
        line["worst_vs_oracle"] = check[head]["worst_vs_oracle"]
        line["checked_positions"] = check[head]["checked_positions"]
    if full:
        line["worst_vs_oracle_200_positions"] = full[head]
        line["worst_vs_oracle_200_positions_source"] = full["source"]
    if bypos:
        for m_ in modes:
            line[("tolerance" if m_ == "tol" else m_) + "_mode"]["by_position"] = bypos[m_]
        line["by_position"] = bypos[head]      # the headline mode with 200 / 1 000 / 1 900 positions of context in front
        if "bar" in bypos and "bar" not in modes:
            line["bar_mode"] = {"mode": MODE_TEXT["bar"], "by_position": bypos["bar"], "whole_context_vs_oracle": bar_whole_context(), "fast_attention_everywhere": bar0}
        if deep:
            line["by_position_oracle_check"] = deep["sample"]
    if trait:
        line["trait_ops_path"] = trait
    if prefill:
        line["prefill"] = prefill      # prompt ingestion (rama_prefill), the same resident model; not part of `value`
    if others:
        line["other_configs"] = others
    # provenance: the committed files this line quotes (roofline.traffic, worst_vs_oracle_200_positions, bar mode's whole-context record) were collected on
    # the library whose hash they carry; a file of another build -- or of a round that wrote no stamp -- is said so here
    quoted = {}
    rl = line.get("roofline") or {}
    if rl.get("traffic_source"):
        quoted[rl["traffic_source"]] = profile_stamp(REPO / rl["traffic_source"])
    if full:
        quoted[full["source"]] = profile_stamp(REPO / full["source"])
    bw = (line.get("bar_mode") or {}).get("whole_context_vs_oracle")
    if bw:
        quoted[bw["source"]] = bw.get("library")
    line["profiles_quoted"] = quoted
    line["profile_matches_build"] = bool(quoted) and not lib.get("stale_build") and all(v is not None and v == lib["src_sha16"] for v in quoted.values())
    print(json.dumps(line), flush=True)
    dev.close()


# ------------------------------------------------------------------ N > 1 without a launcher: start the ranks here

def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(argv, n, timeout, python=sys.executable, script=None, extra_env=None):
    """Start `n` ranks of this script as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, a free rendezvous port), BEFORE anything in this process has touched a GPU.  Rank 0's stdout is
    passed through (the one JSON line); any rank failing or the timeout expiring ends the others and gives a
    non-zero exit code.  Returns the exit code."""
    script = script or str(Path(__file__).resolve())
    port = _free_port()
    import threading
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RAMA_SELF_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL / NCCL log to STDOUT unless told otherwise: a debug level passed through would land in the one JSON line
        if env.get("NCCL_DEBUG", "").upper() in ("INFO", "TRACE") or env.get("RAMA_NCCL_DEBUG", "").upper() in ("INFO", "TRACE"):
            env.setdefault("NCCL_DEBUG_FILE", f"/tmp/rama_rccl_{os.getpid()}_rank%h_%p.log")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([python, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    # rank 0's stdout is drained WHILE it runs: a child that writes more than the pipe holds (64 KiB) would otherwise block in
    # write() until the timeout
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + timeout
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0:
                        print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
                        rc = rc or code or 1
            if rc or time.time() > deadline:
                if not rc:
                    print(f"bench.py: ranks still running after {timeout:.0f} s", file=sys.stderr)
                    rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:       # the exact children started above, nothing else
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    out = b"".join(chunks).decode(errors="replace")
    if rc == 0:
        sys.stdout.write(out)
        sys.stdout.flush()
    else:
        sys.stderr.write(out)
    return rc


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and not os.environ.get("RAMA_BENCH_SKIP_DEVICE_CHECK"):
        # N ranks need N visible devices: say so HERE, before anything touches a GPU (counting devices does not initialise one), instead of letting the ranks
        # fail one by one inside RCCL's bootstrap
        try:
            import torch
            visible = torch.cuda.device_count()
        except Exception:      # no torch: the ranks will find out
            visible = None
        if visible is not None and visible < args.gpus:
            print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible devices, this node shows {visible} "
                  f"(RAMA_FORCE_PIPELINE=1 python bench.py --gpus 1 rehearses the pipeline path on one)", file=sys.stderr)
            raise SystemExit(3)
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # no launcher around us: become one.  Nothing has touched a GPU yet (rama_amd is imported further down).
        raise SystemExit(spawn_ranks(sys.argv[1:], args.gpus, args.rank_timeout))
    if world != args.gpus:
        args.gpus = world

    import rama_amd

    # RAMA_FORCE_PIPELINE=1 rehearses the N > 1 code path (process group, the stage objects, the
    # grouped exchanges' bookkeeping) with a single rank on a 1-GPU box
    if args.gpus > 1 or os.environ.get("RAMA_FORCE_PIPELINE"):
        from rama_amd.pipeline import run_pipeline_bench
        d, h, L, H, V, seq, shared = SHAPES[args.config]
        cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
        line = run_pipeline_bench(args, cfg, rank, world, local_rank)
        if rank == 0:
            print(json.dumps(line), flush=True)
        return
    single_gpu(args, local_rank)


if __name__ == "__main__":
    main()
