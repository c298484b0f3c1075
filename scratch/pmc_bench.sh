#!/bin/bash
# HBM traffic counters for every kernel of the bench step (separate passes: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-scenes 0 --no-reference-order --no-native-arith > /dev/null 2>&1
done
cd $R
python3 - <<PY
import csv, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open("gpurun_out/pmc_%s/p_counter_collection.csv" % c)))
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in agg:
        out.setdefault(k, {})[c + "_KB_per_launch"] = agg[k] / len(n[k])
        out[k]["launches"] = len(n[k])
keep = {k: v for k, v in out.items() if not k.startswith("Cijk") and "at::native" not in k and "rocclr" not in k}
json.dump(keep, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
for k, v in sorted(keep.items()): print(k, v)
PY
