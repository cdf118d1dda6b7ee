#!/usr/bin/env python3
"""The C++ CLI (rama_amd/bin/engine, the mirror of engine/src/main.rs) at the llama2-7B shape: a synthetic v0 checkpoint written to a scratch
directory by rama_model_save, a 32 000-entry tokenizer file, then the CLI on its three paths -- RAMA_PATH=ops (forward() composed from the 1:1
Device ops on tensor-by-tensor uploads: the reference's own structure, compiled host), fused, chained -- in parity mode (the default) and in
fast mode, [r6] and in bar mode (RAMA_REF_ORDER=3).  Prints one JSON line per run.  RAMA_CLI_PATHS=chained restricts the paths.  Usage: python tools/cli_7b_bench.py [scratch dir] [steps]"""
import json, os, re, struct, subprocess, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import rama_amd
from bench import SHAPES

scratch = Path(sys.argv[1] if len(sys.argv) > 1 else "/tmp/rama_cli7b")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
scratch.mkdir(parents=True, exist_ok=True)
ckpt, tokp = scratch / "l7b.bin", scratch / "tok.bin"
d, h, L, H, V, seq, shared = SHAPES["llama2-7B"]
if not ckpt.exists():
    dev = rama_amd.Hip(0)
    m = rama_amd.Model.synth(dev, rama_amd.Config(d, h, L, H, H, V, seq, shared), seed=0)
    t0 = time.time(); m.save(ckpt); print(f"# wrote {ckpt} ({ckpt.stat().st_size / 1e9:.1f} GB) in {time.time() - t0:.0f} s", file=sys.stderr)
    m.free(); dev.close()
entries = [("<unk>", 0.0), ("<s>", 0.0), ("</s>", 0.0)] + [(f"t{i} ", -float(i)) for i in range(3, V)]
with open(tokp, "wb") as f:
    f.write(struct.pack("<I", max(len(s.encode()) for s, _ in entries)))
    for s, score in entries:
        b = s.encode(); f.write(struct.pack("<fi", score, len(b))); f.write(b)
modes = (("parity", None), ("bar", "3"), ("fast", "0"))      # [r6] bar: RAMA_REF_ORDER=3
paths = tuple(os.environ.get("RAMA_CLI_PATHS", "ops,fused,chained").split(","))
for mode, ro in modes:
    for path in paths:
        env = dict(os.environ, RAMA_PATH=path)
        env.pop("RAMA_REF_ORDER", None)
        if ro is not None:
            env["RAMA_REF_ORDER"] = ro
        t0 = time.time()
        r = subprocess.run([str(REPO / "rama_amd" / "bin" / "engine"), "-m", str(ckpt), "-t", str(tokp), "-p", "", "-s", str(steps), "-r", "0"],
                           capture_output=True, text=True, env=env, timeout=1200)
        mt = re.search(r"avg tok/s: ([0-9.eE+-]+)", r.stdout)
        print(json.dumps({"shape": "llama2-7B", "mode": mode, "path": path, "steps": steps, "rc": r.returncode, "cli_avg_tok_s": float(mt.group(1)) if mt else None,
                          "wall_s_incl_load": round(time.time() - t0, 1), "stderr": r.stderr[-200:] if r.returncode else ""}), flush=True)
