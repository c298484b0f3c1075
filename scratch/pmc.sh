#!/bin/bash
# usage: scratch/pmc.sh <stage> <kernel-substring> [counters...]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; stage=$1; pat=$2; shift 2
cd /tmp; rm -rf $R/gpurun_out/pmc_tmp
REPS=2 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_tmp -o p -- python3 $R/scratch/vc_bench.py $stage > /dev/null 2>&1
cd $R
python3 - "$pat" <<PY
import csv,collections,sys
pat=sys.argv[1]
rows=list(csv.DictReader(open("gpurun_out/pmc_tmp/p_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in rows:
    k=r["Kernel_Name"][:48]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k,v in agg.items():
    if pat in k: print(k, "launches", len(n[k]), {a: round(b/len(n[k])) for a,b in v.items()})
tr=list(csv.DictReader(open("gpurun_out/pmc_tmp/p_kernel_trace.csv")))
d=collections.defaultdict(list)
for r in tr: d[r["Kernel_Name"][:48]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items():
    if pat in k or "vote_frames" in k: print(k, "us:", [round(x) for x in v])
PY
