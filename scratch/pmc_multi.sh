#!/bin/bash
# usage: scratch/pmc_multi.sh <kernel-substring> <script.py> [args] ; counter groups from $PMC_GROUPS (';'-separated) -> per-launch averages
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; K=$1; shift
IFS=';' read -ra GROUPS_ <<< "$PMC_GROUPS"
i=0
for P in "${GROUPS_[@]}"; do
  i=$((i+1)); cd /tmp; rm -rf $R/gpurun_out/pm_$i
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pm_$i -o p -- python3 $R/$@ > $R/gpurun_out/pm_$i.log 2>&1
done
cd $R
python3 - "$K" $i <<'PY'
import csv, collections, sys, glob
K, n = sys.argv[1], int(sys.argv[2])
for i in range(1, n + 1):
    for f in glob.glob("gpurun_out/pm_%d/**/*counter_collection.csv" % i, recursive=True):
        agg = collections.defaultdict(float); cnt = collections.defaultdict(set); dur = {}
        for r in csv.DictReader(open(f)):
            if K not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]].add(r["Dispatch_Id"])
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for c in agg: print("%-36s %18.0f  (%d launches, %.1f us)" % (c, agg[c] / len(cnt[c]), len(cnt[c]), sum(dur.values()) / max(1, len(dur))))
PY
