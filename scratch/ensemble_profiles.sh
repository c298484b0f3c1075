#!/bin/bash
# Round profiles of the ensemble workload (BASELINE configs[2]) on the GPU box: kernel stats + launch order of one step, PMC traffic
# passes (FETCH_SIZE / WRITE_SIZE separately), the bench line with the CPU baseline.  usage: bash scratch/ensemble_profiles.sh [r4]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; RND=${1:-r4}; TAG=${RND}_ensemble
ARGS="--workload ensemble --steps 5 --warmup 2 --cpu-scenes 0 --no-counters --single-stream"
cd /tmp; rm -rf $R/gpurun_out/prof_ens
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ens -o $TAG -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_ens.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_ens_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_ens_$c -o p -- python3 $R/bench.py --workload ensemble --steps 2 --warmup 1 --cpu-scenes 0 --no-counters --single-stream > /dev/null 2>&1
done
cd $R
mkdir -p gpurun_out/profiles_new
cp gpurun_out/prof_ens/${TAG}_kernel_stats.csv gpurun_out/profiles_new/
python3 scratch/step_trace.py gpurun_out/prof_ens/${TAG}_kernel_trace.csv > gpurun_out/profiles_new/${TAG}_step_trace.txt
python3 - "$TAG" <<'PY'
import csv, collections, json, sys
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open("gpurun_out/pmc_ens_%s/p_counter_collection.csv" % c)))
    dur = collections.defaultdict(dict)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cls = {}
    for k, d in dur.items():                      # a kernel launched for the tuples AND for the kept pairs: two duration classes
        mx = max(d.values())
        if "reslayer_split_kernel" in k and min(d.values()) < 0.25 * mx:
            for i, t in d.items():
                cls[(k, i)] = k + ("#large" if t >= 0.25 * mx else "#small")
    agg = collections.defaultdict(float); n = collections.defaultdict(set); us = collections.defaultdict(float)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        k = cls.get((k, r["Dispatch_Id"]), k)
        agg[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in agg:
        out.setdefault(k, {})[c + "_KB_per_launch"] = agg[k] / len(n[k])
        out[k]["launches"] = len(n[k])
keep = {k: v for k, v in out.items() if not k.startswith("Cijk") and "at::native" not in k and "rocclr" not in k}
json.dump(keep, open("gpurun_out/profiles_new/%s_pmc_traffic.json" % tag, "w"), indent=1)
PY
python3 bench.py --workload ensemble --steps 50 --cpu-scenes 2 > gpurun_out/ens_cur.json 2> gpurun_out/ens_cur.err
tail -1 gpurun_out/ens_cur.json > gpurun_out/profiles_new/${TAG}_bench_n1.json
tail -1 gpurun_out/ens_cur.json | cut -c1-600
tail -3 gpurun_out/profiles_new/${TAG}_step_trace.txt
