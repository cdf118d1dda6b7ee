#!/bin/bash
# A/B of one tuning key on one box (parity mode, llama2-7B, quick): tools/ab_tune.sh <name> KEY V1 V2 ...   -> gpurun_out/<name>_KEY_V.json
name=$1; key=$2; shift 2
for v in "$@"; do
  python bench.py --mode parity --steps 64 --warmup 8 --no-cpu-baseline --no-kprof --no-prefill --no-other-configs --no-sampled --no-by-position --no-trait-ops --tune $key=$v > gpurun_out/${name}_${key}_$v.json 2> gpurun_out/${name}_${key}_$v.err || exit 1
  python - <<PY
import json
d = json.load(open("gpurun_out/${name}_${key}_$v.json"))
print("$key=$v", d["value"], "tok/s", d["ms_per_step"], "ms")
PY
done
