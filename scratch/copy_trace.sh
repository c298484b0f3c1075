#!/bin/bash
# which copies / small torch kernels does one bench step launch, and how long do they take (kernel trace of 3 steps)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf $R/gpurun_out/ct
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/ct -o t -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-scenes 0 --no-reference-order > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/ct/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = after the last sample_tuples_kernel launch
idx = [i for i, r in enumerate(rows) if "sample_tuples_kernel" in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
agg = collections.OrderedDict()
for r in last:
    n = r["Kernel_Name"].split("(")[0][:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
print("kernels of the last step: %d launches, %.3f ms busy, span %.3f ms" % (len(last), sum(v[1] for v in agg.values()) / 1e3, (int(last[-1]["End_Timestamp"]) - t0) / 1e6))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%4d %9.1f us  %s" % (c, d, n))
m = glob.glob("gpurun_out/ct/**/t_memory_copy_trace.csv", recursive=True)
if m:
    cp = [r for r in csv.DictReader(open(m[0])) if int(r["Start_Timestamp"]) >= t0]
    print("memory copies in the last step:", len(cp))
    for r in cp[:30]:
        print("   ", r.get("Direction"), r.get("Bytes", r.get("Size")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
