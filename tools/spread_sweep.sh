#!/bin/bash
# parity-mode tokens/s at several positions with the spread attention switched on from position X on (tools; run on the GPU box)
for P in ${POSITIONS:-200 400 600 800 1000}; do
  for X in ${THRESHOLDS:-128 1024}; do
    python3 bench.py --pos0 $P --steps ${STEPS:-48} --warmup 4 --mode parity --no-cpu-baseline --no-other-configs --no-prefill --no-sampled --no-by-position --no-trait-ops --no-kprof --tune spread_pos=$X 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pos0 $P spread_pos $X', d['value'], d['ms_per_step'], d['config']['workload'][-16:])"
  done
done
