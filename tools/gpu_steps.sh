#!/bin/bash
# Run GPU steps one after another on the box: tools/gpu_steps.sh "name|seconds|command" ...
# Each step runs under `timeout -k 10`; its output goes to gpurun_out/<name>.log.  A step that fails with an ordinary error does not stop the
# following ones; a step that TIMED OUT or was KILLED does (nothing further is started on a GPU that may be wedged).
mkdir -p gpurun_out
for spec in "$@"; do
    name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
    echo "=== $name (limit ${secs}s): $cmd" | tee -a gpurun_out/steps.log
    start=$(date +%s)
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "=== $name rc=$rc after $(( $(date +%s) - start ))s" | tee -a gpurun_out/steps.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name timed out or was killed: stopping" | tee -a gpurun_out/steps.log; exit $rc; fi
done
exit 0
