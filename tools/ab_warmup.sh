# does the state of the part before the timed region matter?  bench.py --settle-s S: S seconds of untimed decoding over the same positions first
for r in 1 2 3 4; do for w in 0 3; do python bench.py --settle-s $w --mode parity --steps 128 --warmup 8 --no-cpu-baseline --no-kprof --no-prefill --no-other-configs --no-sampled --no-by-position --no-trait-ops > gpurun_out/wu.json 2>/dev/null && python -c "
import json; d=json.load(open('gpurun_out/wu.json')); print('settle_s $w', d['value'])" || exit 1; done; done
