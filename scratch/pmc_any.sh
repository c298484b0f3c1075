#!/bin/bash
# usage: scratch/pmc_any.sh "<counters>" <kernel-substring> <script.py> [args]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; P="$1"; K=$2; shift; shift; cd /tmp
rm -rf $R/gpurun_out/pa
rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pa -o p -- python3 $R/$@ > $R/gpurun_out/pa.log 2>&1
cd $R
python3 - "$K" <<'PY'
import csv, collections, sys, glob
K = sys.argv[1]
for f in glob.glob("gpurun_out/pa/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if K not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    for c in agg: print("%-32s %16.0f  (%d launches)" % (c, agg[c] / len(n[c]), len(n[c])))
PY
tail -2 $R/gpurun_out/pa.log
