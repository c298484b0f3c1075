#!/bin/bash
# usage: scratch/kstats.sh <script.py> [args]  -> per-kernel avg us
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf $R/gpurun_out/ks
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks -o k -- python3 $R/$@ > /dev/null 2>&1
cd $R
python3 - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/ks/k_kernel_stats.csv")))[:12]:
    if "at::native" in r["Name"] or "Cijk" in r["Name"]: continue
    print(r["Name"][:44].ljust(44), r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
