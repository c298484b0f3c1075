#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; pat=$1; shift
cd /tmp; rm -rf $R/gpurun_out/pmc_tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_tmp -o p -- python3 $R/scratch/shot_bench.py > /dev/null 2>&1
cd $R
python3 - "$pat" <<PY
import csv,collections,sys
pat=sys.argv[1]
rows=list(csv.DictReader(open("gpurun_out/pmc_tmp/p_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in rows:
    k=r["Kernel_Name"][:30]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k,v in agg.items():
    if pat in k: print(k, "launches", len(n[k]), {a: round(b/len(n[k])) for a,b in v.items()})
PY
