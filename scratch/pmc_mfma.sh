#!/bin/bash
# usage: scratch/pmc_mfma.sh <tag> <script.py> [args]  -> matrix-pipe counters of reslayer_split_kernel launches -> gpurun_out/mfma_<tag>.json
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=$1; shift; cd /tmp
rm -rf $R/gpurun_out/mf
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_${MOPS:-BF16} --output-format csv -d $R/gpurun_out/mf -o p -- python3 $R/$@ > $R/gpurun_out/mf.log 2>&1
cd $R
python3 - "$TAG" <<'PY'
import csv, collections, json, sys, glob
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/mf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "reslayer_split" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for f in glob.glob("gpurun_out/mf/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "reslayer_split" in k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k in agg:
    L = len(n[k]); c = {m: v / L for m, v in agg[k].items()}
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
    us = sum(dur[k]) / max(1, len(dur[k]))
    out[k] = dict(launches=L, counters_per_launch=c, avg_us=us, shader_cycles_per_xcd=cyc, effective_clock_ghz=cyc / us / 1e3,
                  mfma_busy_frac_of_simd_cycles=c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc,
                  wait_frac_of_wave_cycles=c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"])
json.dump(out, open("gpurun_out/mfma_%s.json" % tag, "w"), indent=1)
for k, v in out.items(): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "counters_per_launch"})
PY
