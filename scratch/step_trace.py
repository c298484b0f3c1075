"""Kernels of ONE steady-state step (the last complete pass of a rocprofv3 --kernel-trace run of bench.py), in launch order:
usage: python scratch/step_trace.py <kernel_trace.csv> > profiles/rN_step_trace.txt"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
starts = [i for i, r in enumerate(rows) if name(r).startswith("sample_tuples_kernel")]
lo, hi = starts[-2], starts[-1]
t0 = int(rows[lo]["Start_Timestamp"])
print("# launch order of one steady-state step (between the last two sample_tuples_kernel launches): start us, duration us, kernel")
tot = 0.0
for r in rows[lo:hi]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print("%10.1f %9.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, name(r)))
foreign = [name(r) for r in rows[lo:hi] if "at::native" in name(r) or name(r).startswith("Cijk") or "rocblas" in name(r)]
print("# %d launches, %.3f ms of kernel time, %.3f ms from first start to last end; PyTorch / BLAS kernels in the step: %d %s"
      % (hi - lo, tot / 1e3, (int(rows[hi - 1]["End_Timestamp"]) - t0) / 1e6, len(foreign), foreign))
