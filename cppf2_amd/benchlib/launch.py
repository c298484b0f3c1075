"""Process-level plumbing of bench.py: the command line, the TunableOp table of a rank, core pinning, and the self-launcher that
starts one fresh rank process per GPU when `python bench.py --gpus N` runs without a launcher environment.  Nothing here touches
the GPU (torch.cuda.device_count() does not initialise it) and nothing re-execs: children are started with Popen and waited for."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BENCH = os.path.join(ROOT, "bench.py")

SET_HERE = []          # environment variables this process set itself (not inherited by the ranks self_launch starts)


def use_tuned_gemms():
    """PyTorch TunableOp with the GEMM solutions recorded for this workload's shapes on gfx950
    (cppf2_amd/tunableop/gfx950_bench_shapes.csv, produced by one `PYTORCH_TUNABLEOP_TUNING=1 python bench.py` run;
    +4 % on the fp32 MLP).  No tuning happens at run time; a file recorded for other library versions is ignored by
    PyTorch (validator lines).  Must run before torch initialises; respects an explicit PYTORCH_TUNABLEOP_* setup."""
    if "PYTORCH_TUNABLEOP_ENABLED" in os.environ:
        return
    src = os.path.join(ROOT, "cppf2_amd", "tunableop", "gfx950_bench_shapes.csv")
    if not os.path.exists(src):
        return
    import atexit
    import shutil
    import tempfile
    # one directory per rank process (self_launch strips these variables from its children, so every rank gets here)
    d = tempfile.mkdtemp(prefix="cppf_tunableop_r%s_" % os.environ.get("RANK", "0"))
    atexit.register(shutil.rmtree, d, True)
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    shutil.copy(src, os.path.join(d, "gemm%d.csv" % dev))          # PyTorch appends the device ordinal to the name
    for k_, v_ in (("PYTORCH_TUNABLEOP_ENABLED", "1"), ("PYTORCH_TUNABLEOP_TUNING", "0"),
                   ("PYTORCH_TUNABLEOP_FILENAME", os.path.join(d, "gemm.csv"))):
        os.environ[k_] = v_
        SET_HERE.append(k_)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scenes-per-gpu", type=int, default=64)
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--tuples", type=int, default=20000)
    ap.add_argument("--rots", type=int, default=180)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-scenes", type=int, default=3, help="scenes of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--agreement-scenes", type=int, default=1 << 30,
                    help="scenes of rank 0's timed batch the CPU oracle re-computes end to end for oracle_agreement (default: all of "
                         "them; worker processes started before the GPU is initialised; 0 = skip; needs --cpu-scenes > 0)")
    ap.add_argument("--agreement-voxel-scenes", type=int, default=8,
                    help="scenes of the voxel-density loop (and instances of --workload ensemble) the oracle re-computes")
    ap.add_argument("--no-prior-variants", action="store_true",
                    help="skip the extra timed loops without the teacher prior (value_no_prior) and with the prior as a [T, 6, 32] "
                         "array (value_array_prior)")
    ap.add_argument("--no-launch-power", action="store_true",
                    help="skip the per-launch power / clock loops of the tuple MLP (roofline.power.mlp_launches)")
    ap.add_argument("--vote-mode", type=int, default=0)
    ap.add_argument("--array-prior", action="store_true",
                    help="pass the teacher prior as a [T, 6, 32] array instead of its generator (ops.BinPrior): same records, + 0.34 ms")
    ap.add_argument("--mlp-reserve-cus", type=int, default=None,
                    help="CUs the persistent tuple-MLP launches leave to the other stream in the two-stream loop "
                         "(default: one per shader engine = CUs / 8; 0 = none)")
    ap.add_argument("--eager-scale-head", action="store_true",
                    help="run the scale head on every tuple like the reference's forward (default: only on the pairs "
                         "that survive the back-vote filter, the only rows eval.py:272 ever reads)")
    ap.add_argument("--no-reference-order", action="store_true",
                    help="skip the second timed loop that measures the other scale-head placement (value_reference_order)")
    ap.add_argument("--mlp-arith", choices=("split", "split16", "native"), default=None,
                    help="arithmetic of the tuple MLP: split = float32 as 3 x bf16 on the bf16 matrix cores (default, exact "
                         "products, cppf_reslayer_split), split16 = float32 as 2 x fp16 (22-23 bits per operand, half the "
                         "matrix-core work, cppf_reslayer_split16), native = f32-input matrix cores (library GEMMs + "
                         "cppf_reslayer128)")
    ap.add_argument("--no-f16x2", action="store_true",
                    help="skip the extra timed loop with the MLP in f16x2 arithmetic (value_f16x2_mfma)")
    ap.add_argument("--materialize-tuples", action="store_true",
                    help="write the [T, 360] tuple rows (cppf_encode_tuples_shot) and let the MLP read them back, instead of "
                         "gathering them inside the first ResLayer's kernel (cppf_reslayer_split_gather)")
    ap.add_argument("--separate-encode", action="store_true",
                    help="write the 40 pair features + global indices per tuple with their own kernel (cppf_encode_tuples_shot_heads, "
                         "rounds 3-4) instead of building them inside the first ResLayer's kernel (cppf_reslayer_split_encode)")
    ap.add_argument("--no-native-arith", action="store_true",
                    help="skip the extra timed loop with the MLP on the f32-input matrix cores (value_f32_input_mfma)")
    ap.add_argument("--two-streams", action="store_true", help="(the default since round 4; kept for old command lines)")
    ap.add_argument("--single-stream", "--no-two-streams", dest="single_stream", action="store_true",
                    help="time the headline with every step on ONE HIP stream (rounds 1-3).  Default: consecutive steps "
                         "(independent scene batches) alternate between two streams with double-buffered state -- the product's "
                         "batch mode (eval.run_ensemble runs its two model passes the same way); the single-stream figure is "
                         "still measured and printed as value_single_stream")
    ap.add_argument("--no-evidence", action="store_true",
                    help="skip the untimed accuracy evidence (mlp_error_vs_f64, bin_flip_rate_vs_expf)")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the headline loop rotates its steps over, each with its own resident state (default 2)")
    ap.add_argument("--no-counters", action="store_true",
                    help="skip the rocprofv3 counter passes (roofline.traffic and the unit-activity fractions are then null)")
    ap.add_argument("--counter-child", action="store_true", help=argparse.SUPPRESS)      # set by collect_counters for its children
    ap.add_argument("--workload", choices=("shot", "ensemble", "dense64k"), default="shot",
                    help="shot (default, the headline): BASELINE configs[1], the SHOT model; ensemble: BASELINE configs[2], the "
                         "reference's real per-instance loop (eval.py:219-372) -- the DINO model AND the SHOT model vote every "
                         "instance, the pose with the smaller alignment loss is kept; value = instances/s; dense64k: BASELINE "
                         "configs[4], 65 536 pairs per scene, float16 feature table, uncertainty-weighted centre votes (extensions "
                         "the reference does not have; --tuples is ignored, --scenes-per-gpu defaults to 16)")
    ap.add_argument("--breakdown", action="store_true", help="also print the per-stage table to stderr")
    ap.add_argument("--cloud", choices=("synthetic", "voxel2mm"), default="synthetic",
                    help="synthetic (default; BASELINE's workload): cppf2_amd.synth surface samples, ~90 neighbours inside the 2 cm "
                         "SHOT support; voxel2mm: the same objects sampled the way eval.py:185-201 produces its clouds (one point per "
                         "2 mm voxel of a dense depth-map-like sampling: ~250 neighbours) -- what the descriptor stage costs on real "
                         "inputs.  The default run also times the descriptor stage on voxel2mm clouds (value_voxel_density)")
    ap.add_argument("--no-voxel-density", action="store_true",
                    help="skip the extra timed loop of the default run on voxel2mm clouds (value_voxel_density)")
    return ap.parse_args(argv)


def cdist_forced():
    return os.environ.get("CPPF_DIST_FORCE_COLLECTIVE", "0") not in ("", "0")


def core_slice(cores, local_rank, local_world):
    """The cores local rank r of W keeps, out of the process' allowed set `cores` (any iterable of ids: it may be non-contiguous --
    a cgroup cpuset, SMT siblings removed -- and smaller than W): the r-th of W near-equal runs of the SORTED ids (sizes differ by at
    most one, every id goes to exactly one rank); with fewer cores than ranks, rank r shares core r mod len(cores).  Pure function
    (tests/test_host_logic.py)."""
    cores = sorted(set(int(c) for c in cores))
    if not cores or local_world < 1 or not (0 <= local_rank < local_world):
        return []
    n = len(cores)
    if n < local_world:
        return [cores[local_rank % n]]
    lo, hi = local_rank * n // local_world, (local_rank + 1) * n // local_world
    return cores[lo:hi]


def pin_rank_to_cores(local_rank, local_world):
    """One run of the process' allowed cores per local rank (core_slice): the host threads of a rank -- launch loop, RCCL proxy,
    the allocator -- stay on neighbouring cores instead of migrating across all of them while eight ranks launch ~30 kernels per
    12 ms step each.  On the usual 8-GPU boards GPUs 0..3 / 4..7 hang off sockets 0 / 1 and core ids are socket-major, so runs in
    LOCAL_RANK order are NUMA-local as well; nothing here depends on that (a different topology costs locality, not correctness).
    CPPF_BENCH_NO_AFFINITY=1 leaves the affinity alone.  Returns {"cores": n, "first": id, "last": id, "contiguous": bool} or None."""
    if local_world <= 1 or os.environ.get("CPPF_BENCH_NO_AFFINITY"):
        return None
    try:
        mine = core_slice(os.sched_getaffinity(0), local_rank, local_world)
        if not mine:
            return None
        os.sched_setaffinity(0, mine)
        return {"cores": len(mine), "first": mine[0], "last": mine[-1], "contiguous": mine[-1] - mine[0] + 1 == len(mine)}
    except (AttributeError, OSError):
        return None


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) started without a launcher environment: run N FRESH rank processes (one per GPU,
    the environment torch.distributed.run would give them) and exit with their status; rank 0 prints the JSON line on the
    inherited stdout.  This process never initialises the GPU (device_count() does not) and never re-execs itself."""
    import socket
    import subprocess
    import torch
    n = args.gpus
    backend = os.environ.get("CPPF_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if have < n and backend == "nccl":
        print("bench.py: --gpus %d but %d GPU(s) are visible; one rank per GPU over RCCL needs %d (CPPF_BENCH_BACKEND=gloo "
              "is the dry-run switch that lets ranks share a GPU)" % (n, have, n), file=sys.stderr)
        return 2
    if have < 1:
        print("bench.py needs a GPU", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return run_ranks([[sys.executable, BENCH] + sys.argv[1:]] * n, have, port)


LAUNCH_TIMEOUT_S = 1500.0


def run_ranks(cmds, visible_gpus, port, timeout_s=None, out=None):
    """Starts one fresh process per entry of `cmds` (rank = its index; the environment torch.distributed.run would give it), waits
    for them WITHOUT polling (blocking waitpid), and returns the exit status of the run: 0 if every rank exited 0.  On the first
    failing rank -- or when the run exceeds `timeout_s` (CPPF_BENCH_LAUNCH_TIMEOUT_S, default 1500 s) -- the other ranks, which
    would otherwise wait in a collective for their own timeout, are stopped by PID (SIGTERM, then SIGKILL after 10 s), and EVERY
    rank's exit status and the tail of its stderr are printed, so that the first real multi-GPU run explains itself.  Rank 0's
    stdout is this process' stdout (the JSON line); every rank's stderr goes to a file of its own and rank 0's is forwarded at the
    end."""
    import signal
    import subprocess
    import tempfile
    import time
    n = len(cmds)
    timeout_s = float(os.environ.get("CPPF_BENCH_LAUNCH_TIMEOUT_S", LAUNCH_TIMEOUT_S)) if timeout_s is None else float(timeout_s)
    out = out or sys.stderr
    d = tempfile.mkdtemp(prefix="cppf_ranks_")
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % max(1, visible_gpus)), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        for k_ in SET_HERE:                     # every rank sets up its own TunableOp table (its own device ordinal and directory)
            env.pop(k_, None)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        errs.append(os.path.join(d, "rank%d.err" % r))
        with open(errs[-1], "w") as ef:
            procs.append(subprocess.Popen(cmds[r], env=env, cwd=os.getcwd(), stderr=ef,
                                          stdout=None if r == 0 else subprocess.DEVNULL))
    live = {p_.pid: (r, p_) for r, p_ in enumerate(procs)}
    status = [None] * n
    why = None

    class _Timeout(Exception):
        pass

    def on_alarm(*_):
        raise _Timeout()
    old = signal.signal(signal.SIGALRM, on_alarm)
    signal.setitimer(signal.ITIMER_REAL, timeout_s)
    try:
        while live:
            try:
                pid, st = os.waitpid(-1, 0)          # sleeps until a child exits
            except ChildProcessError:
                break
            if pid not in live:
                continue
            r, p_ = live.pop(pid)
            status[r] = p_.returncode = os.waitstatus_to_exitcode(st)
            if status[r] != 0 and why is None:
                why = "rank %d exited with status %d" % (r, status[r])
                break
    except _Timeout:
        why = "the run exceeded %.0f s" % timeout_s
    finally:
        signal.setitimer(signal.ITIMER_REAL, 0)
        signal.signal(signal.SIGALRM, old)
    if live:                                        # a failed run: stop exactly the PIDs that are left
        for r, p_ in live.values():
            p_.terminate()
        t_end = time.monotonic() + 10.0
        for r, p_ in live.values():
            try:
                status[r] = p_.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p_.kill()
                status[r] = p_.wait()
            status[r] = "stopped by the launcher (%s)" % status[r]

    def tail(path, lines):
        try:
            with open(path, errors="replace") as f:
                return "".join(f.readlines()[-lines:])
        except OSError:
            return ""
    if why is not None:
        print("bench.py: multi-rank run FAILED: %s.  Per rank:" % why, file=out)
        for r in range(n):
            print("---- rank %d: exit %s; last lines of its stderr:\n%s" % (r, status[r], tail(errs[r], 15).rstrip()), file=out)
        rc = next((s_ for s_ in status if isinstance(s_, int) and s_ > 0), 1)
    else:
        out.write(tail(errs[0], 200))              # rank 0's diagnostics of a good run (self-check messages, warnings)
        rc = 0
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    return rc
