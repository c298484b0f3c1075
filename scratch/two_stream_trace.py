"""Launch order of two consecutive steps of the two-stream headline loop from a rocprofv3 --kernel-trace run of bench.py: per kernel
its stream (queue), start, duration, and how much of it ran while a kernel of the OTHER stream was running.
usage: python scratch/two_stream_trace.py <kernel_trace.csv> > profiles/rN_two_stream_trace.txt"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
qcol = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
starts = [i for i, r in enumerate(rows) if name(r).startswith("sample_tuples_kernel")]
# the last window in which two different queues alternate: take the 5th- and 3rd-last step starts (the timed two-stream loop
# precedes the single-stream comparison loops in the run; search backwards for a pair of consecutive steps on different queues)
# steps of the two-stream loop alternate between two queues: take two consecutive ones from the middle of the longest such run
q = [rows[i][qcol] for i in starts]
best, cur = (0, 0), 0
for a in range(1, len(q)):
    cur = cur + 1 if (q[a] != q[a - 1] and (a < 2 or q[a] == q[a - 2] or cur == 0)) else 0
    if cur > best[0]:
        best = (cur, a)
if best[0] < 4:
    sys.exit("no run of steps alternating between two queues found")
pick = best[1] - best[0] // 2 - 1
lo, hi = starts[pick], starts[pick + 2]
win = rows[lo:hi]
t0 = int(win[0]["Start_Timestamp"])
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qcol]) for r in win]
print("# two consecutive steps of the two-stream loop (queue = HIP stream): start us, duration us, us of it beside a kernel of the other queue, queue, kernel")
tot = ov_tot = 0.0
for r in win:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qcol]
    ov = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2, q2 in iv if q2 != q)
    tot += (e - s) / 1e3
    ov_tot += ov / 1e3
    print("%10.1f %9.1f %9.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, ov / 1e3, q, name(r)))
span = (max(int(r["End_Timestamp"]) for r in win) - t0) / 1e3
print("# %d launches on %d queues, %.3f ms of kernel time in a %.3f ms window (%.3f ms of it overlapped with the other queue): %.3f ms per step"
      % (len(win), len({r[qcol] for r in win}), tot / 1e3, span / 1e3, ov_tot / 2e3, span / 2e3))
