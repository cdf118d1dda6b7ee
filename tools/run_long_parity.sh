cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py --pos0 1850 --steps 64 --warmup 8 --mode parity --no-cpu-baseline --no-other-configs --no-prefill --no-sampled --no-by-position --no-trait-ops > $GRAFT_REPO_ROOT/gpurun_out/long_parity.json 2> $GRAFT_REPO_ROOT/gpurun_out/long_parity.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -o lp -- python3 $GRAFT_REPO_ROOT/bench.py --pos0 1850 --steps 32 --warmup 4 --graph 0 --mode parity --no-cpu-baseline --no-kprof --no-other-configs --no-prefill --no-sampled --no-by-position --no-trait-ops > /dev/null 2>&1
cp $(find /tmp/lp -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/long_parity_kernel_stats.csv
