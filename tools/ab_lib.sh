#!/bin/bash
# interleaved A/B of the product library against rama_amd/librama_hip_exp.so (make -C rama_amd/csrc exp EXPFLAGS=...) on one box:
# tools/ab_lib.sh <name> <rounds> [bench args...]
name=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do
  for lib in base exp; do
    if [ $lib = exp ]; then export RAMA_HIP_LIB=$PWD/rama_amd/librama_hip_exp.so; else unset RAMA_HIP_LIB; fi
    python bench.py --mode parity --steps 128 --warmup 8 --no-cpu-baseline --no-prefill --no-other-configs --no-sampled --no-by-position --no-trait-ops "$@" > gpurun_out/${name}_${lib}_$r.json 2> gpurun_out/${name}.err || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/${name}_${lib}_$r.json')); k=d.get('kernels') or {}
print('round $r  %-5s %8.2f tok/s  %.4f ms | ' % ('$lib', d['value'], d['ms_per_step']) + ' '.join('%s %.2f' % (n, v['avg_us']) for n, v in k.items()))"
  done
done
