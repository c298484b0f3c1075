#!/bin/bash
# usage: scratch/pmc_sq.sh <kernel-name-substring> <script.py> [args]  -> SQ counters of that kernel (two passes, 8 SQ slots each)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; K=$1; shift; cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf $R/gpurun_out/sq_$i
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/sq_$i -o p -- python3 $R/$@ > $R/gpurun_out/sq_$i.log 2>&1
done
cd $R
python3 - "$K" <<'PY'
import csv, collections, json, sys, glob
K = sys.argv[1]
out = collections.OrderedDict()
for i in (1, 2):
    for f in glob.glob("gpurun_out/sq_%d/**/*counter_collection.csv" % i, recursive=True):
        agg = collections.defaultdict(float); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if K not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
        for c in agg: out[c] = agg[c] / max(1, len(n[c])); out["launches"] = len(n[c])
json.dump({K: out}, open("gpurun_out/sq_%s.json" % K, "w"), indent=1)
for c, v in out.items(): print("%-24s %16.0f" % (c, v))
PY
