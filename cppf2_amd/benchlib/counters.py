"""Hardware counters measured IN THE RUN: before bench.py touches the GPU, rank 0 of a one-rank run starts fresh child processes of
itself under rocprofv3 (the program itself after `--`; counters in their own passes with --kernel-trace only, as
MI355X_MICROARCH.md prescribes): FETCH_SIZE, WRITE_SIZE (HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE KB: the guide's gfx950
correction) and one pass of SQ counters (VALU / LDS / matrix-pipe activity per kernel).  No rocprofv3, a failed or timed-out pass or
--no-counters: the fields are null with the reason -- never a number read from profiles/."""
import os
import sys

from . import launch

XCDS = 8                       # MI355X_MICROARCH.md chip table: GRBM_GUI_ACTIVE is reported summed over the 8 XCDs
MAX_CLOCK_GHZ = 2.4            # same table
VALU_ISSUE_CYCLES = 2.0        # one wave64 VALU instruction per 2 cycles per SIMD-32 (cycle-constants table: v_fma_f32)
N_CU = [256]                   # set by bench.py from the device properties once the GPU is initialised (set_device_cus)


def set_device_cus(n):
    N_CU[0] = int(n)


STAGE_KERNEL = {"sample_tuples": "sample_tuples_kernel", "shot_frames": "shot_cov_kernel", "shot352": "shot_hist_kernel",
                "encode_tuples": "encode_shot_kernel<5, 16>", "decode_bins": "decode_bins_kernel<32>",
                "vote_frames": "vote_frames_kernel", "vote_center": "vote_center_persist_kernel", "backvote_filter": "backvote_kernel",
                "rot_bins": "rot_bins_lut_kernel<2>", "assemble_pose": "assemble_pose_kernel"}


COUNTER_PASSES = {"FETCH_SIZE": ["FETCH_SIZE"], "WRITE_SIZE": ["WRITE_SIZE"],
                  "SQ": ["SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU",
                         "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"]}
COUNTERS = {"reason": "not collected"}


def kernel_key(name):
    """'void reslayer_split_kernel<4, true, ...>(float const*, ...)' -> 'reslayer_split_kernel<4, true, ...>'"""
    n = name.replace("void ", "")
    depth = 0
    for i, ch in enumerate(n):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return n[:i].strip()
    return n.strip()


def stop_group(proc, grace_s=5.0):
    """SIGTERM, then SIGKILL, to the process group `proc` leads (it was started with start_new_session=True), and reap `proc`.
    Only ever the exact group this module started."""
    import signal
    import subprocess
    for sig in (signal.SIGTERM, signal.SIGKILL):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            break
        try:
            proc.wait(timeout=grace_s)
            break
        except subprocess.TimeoutExpired:
            continue
    try:
        proc.wait(timeout=grace_s)
    except subprocess.TimeoutExpired:
        pass


def collect_counters(argv_workload, passes=("FETCH_SIZE", "WRITE_SIZE", "SQ"), timeout_s=600):
    """Runs `rocprofv3 --kernel-trace --pmc <pass> -- python3 bench.py <workload flags> --steps 2 --warmup 1 --counter-child` once per
    pass and returns {kernel key [#large | #small]: {counter: mean per launch, "launches": n, "avg_us": mean duration}} -- a kernel
    launched both for all tuples and for the kept pairs is split into two duration classes."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {"reason": "rocprofv3 not found on PATH or under /opt/rocm/bin"}
    out, fail = {}, []
    tmp = tempfile.mkdtemp(prefix="cppf_bench_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    for k_ in list(env):
        if k_ in launch.SET_HERE:
            env.pop(k_)
    child = [sys.executable, launch.BENCH] + argv_workload + ["--steps", "2", "--warmup", "1", "--counter-child"]
    for pname in passes:
        d = os.path.join(tmp, pname)
        cmd = [exe, "--kernel-trace", "--pmc"] + COUNTER_PASSES[pname] + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + child
        # the pass runs in its own session (process group): on a timeout the WHOLE group -- rocprofv3 and the profiled python child
        # -- is stopped and waited for before anything else touches the GPU or the output directory (a child that outlived its
        # profiler would overlap the timed loops and keep writing into a directory being removed)
        try:
            errf = tempfile.TemporaryFile()
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=errf, start_new_session=True)
        except Exception as e:      # noqa: BLE001
            fail.append("%s: %r" % (pname, e))
            continue
        try:
            rc = proc.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            stop_group(proc)
            fail.append("%s: timed out after %d s; process group %d stopped and reaped" % (pname, timeout_s, proc.pid))
            break                                  # the later passes would time out the same way: do not spend 3 x the limit
        errf.seek(0)
        err_tail = errf.read().decode("utf-8", "replace")[-300:]
        errf.close()
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if rc != 0 or not files:
            fail.append("%s: rocprofv3 exit %d, %d counter files; %s" % (pname, rc, len(files), err_tail))
            continue
        rows = [row for f in files for row in csv.DictReader(open(f))]
        dur = collections.defaultdict(dict)
        for row in rows:
            dur[kernel_key(row["Kernel_Name"])][row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        cls = {}
        for k_, dd in dur.items():
            mx = max(dd.values())
            if "reslayer_split_kernel" in k_ and min(dd.values()) < 0.25 * mx:
                for i_, t_ in dd.items():
                    cls[(k_, i_)] = k_ + ("#large" if t_ >= 0.25 * mx else "#small")
        agg, cnt, us = collections.defaultdict(float), collections.defaultdict(set), collections.defaultdict(dict)
        for row in rows:
            k0 = kernel_key(row["Kernel_Name"])
            k_ = cls.get((k0, row["Dispatch_Id"]), k0)
            agg[(k_, row["Counter_Name"])] += float(row["Counter_Value"])
            cnt[k_].add(row["Dispatch_Id"])
            us[k_][row["Dispatch_Id"]] = dur[k0][row["Dispatch_Id"]] / 1e3
        for (k_, c_), v_ in agg.items():
            e = out.setdefault(k_, {})
            e[c_] = v_ / len(cnt[k_])
            e["launches"] = len(cnt[k_])
            e.setdefault("avg_us", {})[pname] = sum(us[k_].values()) / len(us[k_])
    shutil.rmtree(tmp, ignore_errors=True)
    if fail:
        out["reason"] = "; ".join(fail)
    return out


def counter_entry(name):
    """The counters of the kernel whose key starts with `name` (exact key first)."""
    if not name:
        return None
    if name in COUNTERS:
        return COUNTERS[name]
    hits = [v for k_, v in sorted(COUNTERS.items()) if isinstance(v, dict) and k_.startswith(name)]
    return hits[0] if hits else None


def hbm_bytes(entry):
    """HBM bytes per launch from one kernel's counters: 2 x FETCH_SIZE + WRITE_SIZE (KB), or None."""
    if not entry or "FETCH_SIZE" not in entry or "WRITE_SIZE" not in entry:
        return None
    return (2.0 * entry["FETCH_SIZE"] + entry["WRITE_SIZE"]) * 1024.0


# what limits each stage's kernel: "hbm" = streaming (bytes / time against 8 TB/s), "unit" = an execution unit (VALU or LDS: which
# one, and how busy, comes from the SQ counters), "latency" = one workgroup per scene or a chain of dependent phases
STAGE_BOUND = {"sample_tuples": "hbm", "encode_tuples": "hbm", "decode_bins": "hbm", "vote_frames": "hbm",
               "shot_frames": "unit", "shot352": "unit", "vote_center": "unit", "rot_bins": "unit",
               "backvote_filter": "latency", "assemble_pose": "latency"}


def pmc_traffic(stage):
    return hbm_bytes(counter_entry(STAGE_KERNEL.get(stage, "")))


def unit_activity(entry):
    """Activity of the VALU, LDS and matrix pipes over one launch, each as a fraction of what the unit can do in the launch's shader
    cycles, and never above 1 by construction of the counters (a value above 1 is reported as invalid, not as attainment):
      cycles      = GRBM_GUI_ACTIVE / XCDS, capped at MAX_CLOCK x the kernel's duration in the same pass (for kernels of a few
                    microseconds the GRBM window also covers dispatch time around the kernel: the cap keeps the fraction a lower bound)
      valu_issue  = SQ_INSTS_VALU x 2 cycles / (SIMDs x cycles): wave64 vector instructions against the SIMD-32's issue rate of one
                    per 2 cycles (what 157 TFLOP/s of float32 FMA is; float64, transcendental and DPP forms take longer, so this is a
                    LOWER bound of the time the VALU is occupied)
      valu_busy   = SQ_ACTIVE_INST_VALU x 2 / (SIMDs x cycles): the counter ticks about once per issued instruction on gfx950
                    (measured: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.02 .. 1.05 on the descriptor kernels), so rocprof's gfx9
                    formula (x 4: a SIMD-16 issuing over 4 cycles) reads up to 2 x too high on the SIMD-32 -- rounds 2-4 printed that
      lds_busy    = SQ_LDS_IDX_ACTIVE / (CUs x cycles); lds_bank_conflict likewise; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x cycles)
    CU count from the device properties (set_device_cus), 4 SIMDs per CU."""
    if not entry or "GRBM_GUI_ACTIVE" not in entry:
        return None
    cus = float(N_CU[0])
    simds = 4.0 * cus
    cyc = entry["GRBM_GUI_ACTIVE"] / float(XCDS)
    us = entry.get("avg_us", {}).get("SQ")
    capped = False
    if us and cyc > MAX_CLOCK_GHZ * 1e3 * us:
        cyc, capped = MAX_CLOCK_GHZ * 1e3 * us, True
    if cyc <= 0:
        return None
    vals = dict(valu_issue=VALU_ISSUE_CYCLES * entry.get("SQ_INSTS_VALU", 0.0) / simds / cyc,
                valu_busy=VALU_ISSUE_CYCLES * entry.get("SQ_ACTIVE_INST_VALU", 0.0) / simds / cyc,
                mfma_busy=entry.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simds / cyc,
                lds_busy=entry.get("SQ_LDS_IDX_ACTIVE", 0.0) / cus / cyc,
                lds_bank_conflict=entry.get("SQ_LDS_BANK_CONFLICT", 0.0) / cus / cyc)
    invalid = sorted(k_ for k_, v_ in vals.items() if not (0.0 <= v_ <= 1.0))
    out = {k_: (round(v_, 4) if k_ not in invalid else None) for k_, v_ in vals.items()}
    out.update(shader_clock_ghz=round(cyc / us / 1e3, 3) if us else None,
               cycles_source="MAX_CLOCK x duration (short kernel: GRBM window longer than the launch)" if capped else "GRBM_GUI_ACTIVE / XCDS",
               valu_insts_per_launch=entry.get("SQ_INSTS_VALU"), invalid=invalid or None)
    return out


def kernel_us(name, pass_name="FETCH_SIZE"):
    """The kernel's own average duration (microseconds) in one of this run's counter passes (rocprofv3's begin / end timestamps of
    the dispatch: no launch gap, no event overhead), or None."""
    e = counter_entry(name)
    if not e:
        return None
    a = e.get("avg_us", {})
    return a.get(pass_name) or a.get("WRITE_SIZE") or a.get("SQ")


def tuple_mlp_kernels(pieces=3, folded=True):
    """Counter keys of the tuple MLP's three launches: the gathered 360 -> 128 chain (MODE 3 = RS_ENCODE since round 5 -- the pair
    features are built in the kernel; with --separate-encode or f16x2 arithmetic MODE 0, whose key is shared with the ~20 x shorter
    scale-head launch on the kept pairs: `#large`), 128 -> 256 with the two 256-wide identity layers behind it, 256 -> 192 + bin draw."""
    first = ("reslayer_split_kernel<4, true, true, false, 3, 3>" if (folded and pieces == 3)
             else "reslayer_split_kernel<4, true, true, false, %d, 0>#large" % pieces)
    return (first, "reslayer_split_kernel<8, true, false, false, %d, 0>" % pieces, "reslayer_split_kernel<6, true, false, true, %d, 0>" % pieces)


def pmc_traffic_mlp(pieces=3, folded=True):
    """HBM bytes per step of the tuple MLP's three launches (tuple_mlp_kernels: the launches `launch_ms` times), from this run's
    counter passes.  None when the passes did not run."""
    tot = 0.0
    for k_ in tuple_mlp_kernels(pieces, folded):
        b_ = hbm_bytes(COUNTERS.get(k_) or COUNTERS.get(k_.replace("#large", "")))
        if b_ is None:
            return None
        tot += b_
    return tot


def ensemble_mlp_traffic():
    """HBM bytes per ensemble step of all matrix-core launches (every reslayer_split_kernel instantiation: both tuple MLPs, the
    point encoder, the DINO model's per-point Linear launches, the scale heads), from this run's counter passes."""
    passes = (COUNTERS.get("ensemble_select_kernel") or {}).get("launches")
    if not passes:
        return None
    tot = 0.0
    for k_, v in COUNTERS.items():
        if isinstance(v, dict) and k_.startswith("reslayer_split_kernel"):
            b_ = hbm_bytes(v)
            if b_ is None:
                return None
            tot += b_ * v["launches"]
    return tot / passes
