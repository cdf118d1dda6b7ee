#!/bin/bash
# kernel-trace statistics of one bench.py run (on the GPU box): tools/kstats.sh <name> <bench.py args...>
# -> gpurun_out/<name>_kernel_stats.csv   (rocprofv3 gets `python3 bench.py ...` directly after `--`)
set -e
name=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/_ks_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_ks_$name -o t -- python3 $repo/bench.py --steps 32 --warmup 4 --graph 0 --no-cpu-baseline --no-kprof --no-prefill --no-by-position --no-trait-ops "$@" > $out/${name}_bench.log 2>&1
cp $(find /tmp/_ks_$name -name '*kernel_stats.csv' | head -1) $out/${name}_kernel_stats.csv
python3 - $repo $out/${name}_kernel_stats.meta.json "$@" <<'PY'
import json, sys
sys.path.insert(0, sys.argv[1])
from bench import library_stamp
open(sys.argv[2], "w").write(json.dumps({"library": library_stamp(), "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 32 --warmup 4 --graph 0 --no-cpu-baseline --no-kprof --no-prefill --no-by-position --no-trait-ops " + " ".join(sys.argv[3:])}) + "\n")
PY
python3 - $out/${name}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print(f'{float(r["AverageNs"])/1e3:9.2f} us x{int(r["Calls"]):6d}  {float(r["Percentage"]):5.1f}%  {r["Name"][:110]}')
PY
