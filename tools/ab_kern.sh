#!/bin/bash
# per-class kernel timings (bench.py's event-bracketed launches) of parity mode under several tunings, one box:
# tools/ab_kern.sh <name> "k=v k=v" "k=v" ...   ("-" = the defaults)
name=$1; shift
i=0
for t in "$@"; do
  args=""; if [ "$t" != "-" ]; then for kv in $t; do args="$args --tune $kv"; done; fi
  python bench.py --mode parity --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-other-configs --no-sampled --no-by-position --no-trait-ops $args > gpurun_out/${name}_$i.json 2> gpurun_out/${name}_$i.err || exit 1
  python - <<PY
import json
d = json.load(open("gpurun_out/${name}_$i.json"))
k = d["kernels"]
print("%-32s %7.2f tok/s | " % ("$t", d["value"]) + " ".join("%s %.2f" % (n, v["avg_us"]) for n, v in k.items()))
PY
  i=$((i+1))
done
