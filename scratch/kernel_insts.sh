#!/bin/bash
# Dynamic instruction mix per kernel of a bench step (per wavefront, and the kernel's share of all vector instructions issued):
# usage: bash scratch/kernel_insts.sh [bench flags]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/ki && mkdir -p /tmp/ki
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/ki/a -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-scenes 0 --single-stream --no-reference-order --no-native-arith --no-evidence --no-f16x2 --no-counters "$@" > /tmp/ki/a.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob("/tmp/ki/a/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
tot = sum(c["SQ_INSTS_VALU"] for c in acc.values())
for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])[:24]:
    w = c["SQ_WAVES"] or 1
    print("%-58s launches %3d waves/launch %8.0f | per wave VALU %7.0f SALU %6.0f LDS %6.0f | %4.1f %% of all VALU" % (
        k[:58], len(disp[k]), w / len(disp[k]), c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_LDS"] / w, 100 * c["SQ_INSTS_VALU"] / tot))
PY
tail -1 /tmp/ki/a.log | cut -c1-200
