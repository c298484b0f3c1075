#!/bin/bash
# HBM traffic counters for every kernel of the bench step (separate passes: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-scenes 0 --no-reference-order --no-native-arith --no-evidence --no-f16x2 --no-two-streams > /dev/null 2>&1
done
cd $R
python3 - <<PY
import csv, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open("gpurun_out/pmc_%s/p_counter_collection.csv" % c)))
    # a kernel whose dispatches fall into two duration classes (the gathering MLP kernel: the tuple encoder's launch and the
    # ~20 x shorter scale-head launch) is reported as two entries, "<name>#large" and "<name>#small"
    dur = collections.defaultdict(dict)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cls = {}
    for k, d in dur.items():
        mx = max(d.values())
        if "reslayer_split_kernel" in k and min(d.values()) < 0.25 * mx:
            for i, t in d.items():
                cls[(k, i)] = k + ("#large" if t >= 0.25 * mx else "#small")
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        k = cls.get((k, r["Dispatch_Id"]), k)
        agg[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in agg:
        out.setdefault(k, {})[c + "_KB_per_launch"] = agg[k] / len(n[k])
        out[k]["launches"] = len(n[k])
keep = {k: v for k, v in out.items() if not k.startswith("Cijk") and "at::native" not in k and "rocclr" not in k}
json.dump(keep, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
for k, v in sorted(keep.items()): print(k, v)
PY
