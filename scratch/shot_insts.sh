#!/bin/bash
# Dynamic instruction mix of the SHOT kernels (per wavefront): rocprofv3 counter pass over scratch/shot_stage.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/si && mkdir -p /tmp/si
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/si/a -- python3 $R/scratch/${SCRIPT:-shot_stage.py} "$@" > /tmp/si/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/si/b -- python3 $R/scratch/${SCRIPT:-shot_stage.py} "$@" > /tmp/si/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("/tmp/si/*/*/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "shot" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (f, r["Dispatch_Id"]) not in seen and r["Counter_Name"] in ("SQ_WAVES", "SQ_INSTS_VMEM_RD"):
            seen.add((f, r["Dispatch_Id"])); n[(k, r["Counter_Name"])] += 1
for k, c in acc.items():
    w = c.get("SQ_WAVES", 0) or 1
    print("%-40s waves/launch %.0f  per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f" % (
        k[:40], w / max(1, n[(k, "SQ_WAVES")]), c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_LDS"] / w,
        c["SQ_INSTS_VMEM_RD"] / w * n[(k, "SQ_WAVES")] / max(1, n[(k, "SQ_INSTS_VMEM_RD")]), c["SQ_INSTS_VMEM_WR"] / w * n[(k, "SQ_WAVES")] / max(1, n[(k, "SQ_INSTS_VMEM_RD")])))
PY
tail -2 /tmp/si/a.log
