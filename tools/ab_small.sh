#!/bin/bash
# stories15M / stories110M decode in one mode under tuning sets, one box: tools/ab_small.sh <name> <mode> "k=v" ...  ("-" = defaults)
name=$1; mode=$2; shift 2
for cfg in stories15M stories110M; do
  for t in "$@"; do
    args=""; if [ "$t" != "-" ]; then for kv in $t; do args="$args --tune $kv"; done; fi
    python bench.py --config $cfg --mode $mode --steps 200 --warmup 28 --no-cpu-baseline --no-kprof --no-prefill --no-other-configs --no-sampled --no-by-position --no-trait-ops $args > gpurun_out/${name}.json 2> gpurun_out/${name}.err || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/${name}.json')); print('%-12s %-24s %9.1f tok/s  %.2f us/token' % ('$cfg', '$t', d['value'], d['ms_per_step'] * 1e3))"
  done
done
