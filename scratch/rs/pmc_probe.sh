#!/bin/bash
# usage (on the GPU box): scratch/rs/pmc_probe.sh <variant> [<variant> ...]   -> table of matrix-pipe counters per kernel and variant
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf /tmp/pp_$v; cd /tmp
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d /tmp/pp_$v -o p -- $R/scratch/rs/bin/probe_$v 1280000 3 > /tmp/pp_$v.log 2>&1
  cd $R
  python3 - "$v" <<'PY'
import csv, collections, sys, glob
v = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(list)
order = []
for f in glob.glob("/tmp/pp_%s/**/*counter_collection.csv" % v, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("reslayer_split_kernel", "rs")
        if k not in order: order.append(k)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for f in glob.glob("/tmp/pp_%s/**/*kernel_trace.csv" % v, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("reslayer_split_kernel", "rs")
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in order:
    if "rs" not in k: continue
    L = len(n[k]); c = {m: x / L for m, x in agg[k].items()}
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    us = sum(dur[k]) / max(1, len(dur[k]))
    wc = c["SQ_WAVE_CYCLES"]
    print("%-8s %-26s %8.1f us  clk %.2f GHz  mfma_busy %.3f  busy*clk %.2f | of wave cycles: wait_any %.2f wait_inst %.2f active %.2f | valu insts/mfma %.2f"
          % (v, k, us, cyc / us / 1e3, c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / us / 1e3,
             c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc,
             c["SQ_INSTS_VALU"] / (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0)))
PY
done
