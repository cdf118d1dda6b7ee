#!/bin/bash
# interleaved A/B of tuning sets on one box (parity mode, llama2-7B, hipGraph, 128 steps): tools/ab_multi.sh <name> <rounds> "k=v k=v" "k=v" ...  ("-" = defaults)
name=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do
  i=0
  for t in "$@"; do
    args=""; if [ "$t" != "-" ]; then for kv in $t; do args="$args --tune $kv"; done; fi
    python bench.py --mode parity --steps 128 --warmup 8 --no-cpu-baseline --no-kprof --no-prefill --no-other-configs --no-sampled --no-by-position --no-trait-ops $args > gpurun_out/${name}_${i}_$r.json 2> gpurun_out/${name}.err || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/${name}_${i}_$r.json')); print('round $r  %-40s %8.2f tok/s  %.4f ms' % ('$t', d['value'], d['ms_per_step']))"
    i=$((i+1))
  done
done
