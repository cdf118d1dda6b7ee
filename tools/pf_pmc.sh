#!/bin/bash
# MfmaUtil of the prefill kernels (on the GPU box): tools/pf_pmc.sh <name> <n_positions> [shape]
# -> gpurun_out/<name>_pmc_mfma_util.json   (counters in a pass of their own: --kernel-trace + --pmc only)
set -e
name=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/_pmc_$name
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d /tmp/_pmc_$name -o t -- python3 $repo/tools/prefill_once.py "$@" > $out/${name}_pmc.log 2>&1
python3 - $(find /tmp/_pmc_$name -name '*counter_collection.csv' | head -1) $out/${name}_pmc_mfma_util.json "$@" <<'PY'
import csv, json, sys
from collections import defaultdict
acc = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "MfmaUtil":
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
rows = [{"kernel": k, "launches": len(v), "MfmaUtil_percent": round(sum(v) / len(v), 2)} for k, v in acc.items() if "mfma" in k]
rows.sort(key=lambda r: -r["launches"] * r["MfmaUtil_percent"])
doc = {"command": "rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -- python3 tools/prefill_once.py " + " ".join(sys.argv[3:]) +
                  "   (counters in a pass of their own)",
       "note": "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMD_NUM) * 100, rocprofv3's derived counter: the share of SIMD cycles "
               "with the matrix pipe busy. gemm_mfma_rows<PT, RT, EPI, JN, LD, STAGGER, MIX>: <8,3,2,..> = Wq|Wk|Wv, <8,2,3,..> = W1|W3 + SiLU*gate, "
               "<8,2,0,..> = Wo and W2",
       "rows": rows}
json.dump(doc, open(sys.argv[2], "w"), indent=1)
for r in rows: print(r)
PY
