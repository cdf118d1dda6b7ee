#!/usr/bin/env python3
"""Collect the two rocprofv3 summaries bench.py's numbers are checked against (run on the GPU box):

  1. kernel trace + stats of the default bench workload   -> <out>/rNN_bench_7b_kernel_stats.csv
  2. a SEPARATE counter pass (--pmc FETCH_SIZE, kernel trace only, eager launches)
                                                           -> <out>/rNN_bench_7b_pmc_fetch_size.json
     with the guide's gfx950 correction applied (FETCH_SIZE is in KB and counts 128-B requests at
     64 B: bytes = value * 1024 * 2).

This script never touches the GPU itself; rocprofv3 gets `python3 bench.py ...` directly after `--`.

    python tools/collect_profiles.py --round 1 --out gpurun_out/profiles
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
ap = argparse.ArgumentParser()
ap.add_argument("--round", type=int, default=1)
ap.add_argument("--out", default=str(REPO / "gpurun_out" / "profiles"))
ap.add_argument("--mode", default="fast", choices=["fast", "parity", "tol"], help="which decode mode of bench.py to profile")
ap.add_argument("--config", default="llama2-7B", help="shape of bench.py to profile (files are named ..._bench_7b... for llama2-7B, ..._bench_<shape>... otherwise)")
a = ap.parse_args()
out = Path(a.out).resolve()
out.mkdir(parents=True, exist_ok=True)
tag = f"r{a.round:02d}"
suffix = {"parity": "_parity", "tol": "_tol"}.get(a.mode, "")
mode_args = ["--mode", a.mode, "--no-other-configs", "--no-prefill", "--no-by-position", "--no-trait-ops", "--no-generation-200", "--config", a.config]
short = "7b" if a.config == "llama2-7B" else a.config
env = dict(os.environ, TMPDIR="/tmp")
bench = str(REPO / "bench.py")


sys.path.insert(0, str(REPO))
from bench import library_stamp  # noqa: E402   (hashes rama_amd/librama_hip.so; touches no GPU)
LIBRARY = library_stamp()


def run(cmd, workdir):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return workdir


# 1. kernel trace + stats
d1 = out / "_trace"
shutil.rmtree(d1, ignore_errors=True)
cmd1 = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", str(d1), "-o", "bench", "--",
        "python3", bench, "--steps", "32", "--warmup", "4", "--graph", "0", "--no-cpu-baseline", "--no-kprof", "--no-sampled"] + mode_args     # long graph replays crash rocprofv3 (ROCm 7.2): eager launches, same kernels
run(cmd1, d1)
stats = glob.glob(str(d1 / "**" / "*kernel_stats.csv"), recursive=True)
assert stats, "rocprofv3 wrote no kernel_stats.csv"
shutil.copy(stats[0], out / f"{tag}_bench_{short}{suffix}_kernel_stats.csv")
(out / f"{tag}_bench_{short}{suffix}_kernel_stats.meta.json").write_text(json.dumps({"library": LIBRARY, "command": " ".join(cmd1[:9]) + " -- python3 bench.py " + " ".join(cmd1[cmd1.index(bench) + 1:])}) + "\n")

# 2. counters, in a pass of their own
d2 = out / "_pmc"
shutil.rmtree(d2, ignore_errors=True)
cmd2 = ["rocprofv3", "--kernel-trace", "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", str(d2), "-o", "pmc", "--",
        "python3", bench, "--steps", "16", "--warmup", "2", "--settle-s", "0", "--no-cpu-baseline", "--no-kprof", "--no-sampled", "--graph", "0"] + mode_args      # (counters serialise the launches: no untimed settle passes under them)
run(cmd2, d2)
cc = glob.glob(str(d2 / "**" / "*counter_collection.csv"), recursive=True)
assert cc, "rocprofv3 wrote no counter_collection.csv"
acc = defaultdict(lambda: [0, 0.0])
with open(cc[0]) as f:
    for r in csv.DictReader(f):
        if r.get("Counter_Name") != "FETCH_SIZE":
            continue
        k = acc[r["Kernel_Name"]]
        k[0] += 1
        k[1] += float(r["Counter_Value"])
rows = [{"kernel": name, "counter": "FETCH_SIZE", "launches": n, "avg_value_KB": round(tot / n, 3),
         "hbm_read_bytes_corrected": int(round(tot / n * 1024 * 2))} for name, (n, tot) in sorted(acc.items())]
# whole-token traffic: every launch of the decode loop (one argmax launch per token; model set-up kernels left out)
setup = ("fill_synth", "tile_weights", "interleave_rows", "chain_weights", "rocclr", "rope_table")
tokens = sum(n for name, (n, tot) in acc.items() if "argmax_kernel" in name)
per_token = int(round(sum(tot for name, (n, tot) in acc.items() if not any(k in name for k in setup)) * 1024 * 2 / tokens)) if tokens else None
with open(out / f"{tag}_bench_{short}{suffix}_pmc_fetch_size.json", "w") as f:
    json.dump({"command": " ".join(cmd2[:5]) + f" -- python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-kprof --graph 0 --mode {a.mode} --config {a.config}",
               "library": LIBRARY, "tokens": tokens, "hbm_read_bytes_per_token_corrected": per_token,
               "note": "FETCH_SIZE is reported in KB; on gfx950 it counts 128-B requests at 64 B, i.e. exactly half of a wide "
                       "coalesced read (MI355X_MICROARCH.md, HBM section): hbm_read_bytes_corrected = value * 1024 * 2",
               "rows": rows}, f, indent=1)
shutil.rmtree(d1, ignore_errors=True)
shutil.rmtree(d2, ignore_errors=True)
for r in rows:
    if "true, 5>" in r["kernel"] or "swiglu" in r["kernel"] or "4, 3>" in r["kernel"]:
        print(r)
print("wrote", sorted(p.name for p in out.iterdir()))
