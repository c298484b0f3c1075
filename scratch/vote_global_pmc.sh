#!/bin/bash
# L2-atomic evidence for the global-atomic centre vote: timing of mode 1 (LDS slabs) and mode 2 (global atomics)
# at two grid sizes, then rocprofv3 counter passes (own passes, --kernel-trace only) on vote_center_global_kernel.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/vg; rm -rf $O; mkdir -p $O
for case in "0.236 0.102 0.266 64" "2.0 0.4 0.4 8"; do
  for m in 1 2; do python3 $R/scratch/vote_global.py $m $case 10 2>&1 | grep "^mode" | tee -a $O/timing.txt; done
done
i=0
for case in "0.236 0.102 0.266 64" "2.0 0.4 0.4 8"; do
  i=$((i+1))
  for P in "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum" "TCC_EA0_ATOMIC_LEVEL_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TA_FLAT_ATOMIC_WAVEFRONTS_sum SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    n=$(echo $P | cut -d" " -f1)
    cd /tmp; rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/c${i}_$n -o p -- python3 $R/scratch/vote_global.py 2 $case 3 > $O/c${i}_$n.log 2>&1; cd $R
  done
done
python3 - <<'PY'
import csv, glob, collections, json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "vg")
out = {}
for d in sorted(glob.glob(O + "/c*_*")):
    if not os.path.isdir(d): continue
    case = os.path.basename(d).split("_")[0]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); n = collections.defaultdict(set); dur = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "vote_center_global" not in k and "grid_zero" not in k and "grid_argmax_partial" not in k: continue
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
            dur.setdefault(k, {})[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for (k, c), v in agg.items():
            e = out.setdefault(case, {}).setdefault(k, {})
            e[c] = v / len(n[(k, c)]); e["launches"] = len(n[(k, c)]); e["avg_us"] = sum(dur[k].values()) / len(dur[k])
json.dump(out, open(O + "/counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
