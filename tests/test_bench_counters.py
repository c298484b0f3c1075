"""bench.py's in-run counter passes (cppf2_amd/benchlib/counters.py), on a box without a GPU: the rocprofv3 child is replaced by a stub
that writes the CSV rocprofv3 writes, and the parsing / normalisation (kernel keys, duration classes of a kernel launched at two
sizes, HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE KB, the VALU / LDS / matrix-pipe fractions -- never above 1 -- ) is checked on known
numbers; a pass that hangs is stopped as a whole process group."""
import csv
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    sys.path.insert(0, ROOT)
    mod = importlib.import_module("cppf2_amd.benchlib.counters")
    monkeypatch.setattr(mod, "N_CU", [256])
    return mod


class FakePopen:
    """Stands in for the rocprofv3 process: runs `action(cmd, cwd, env)` (which writes the CSV) and exits with `rc`."""
    action, rc, err = None, 0, b""

    def __init__(self, cmd, cwd=None, env=None, stdout=None, stderr=None, start_new_session=False):
        assert start_new_session, "a counter pass must lead its own process group"
        self.pid = 999999
        if FakePopen.action is not None:
            FakePopen.action(cmd, cwd, env)
        if stderr is not None and FakePopen.err:
            stderr.write(FakePopen.err)

    def wait(self, timeout=None):
        return FakePopen.rc


def test_kernel_key_strips_arguments_but_keeps_template_parentheses(bench):
    k = bench.kernel_key
    assert k("void reslayer_split_kernel<4, true, true, false, 3, 0>(float const*, long, int)") == "reslayer_split_kernel<4, true, true, false, 3, 0>"
    assert k("shot_eig_kernel(long, float const*, double const*)") == "shot_eig_kernel"
    assert k("void foo<(anonymous namespace)::Bar>(int)") == "foo<(anonymous namespace)::Bar>"
    assert k("__amd_rocclr_copyBuffer") == "__amd_rocclr_copyBuffer"


def test_counter_passes_are_parsed_and_normalised(bench, monkeypatch, tmp_path):
    calls = []
    big, small = "void reslayer_split_kernel<4, true, true, false, 3, 0>(float const*)", "void vote_center_persist_kernel<false>(int)"

    def fake_run(cmd, cwd=None, env=None):
        assert cmd[1:3] == ["--kernel-trace", "--pmc"] and "--" in cmd and cmd[cmd.index("--") + 1] == sys.executable
        assert "--counter-child" in cmd and cwd == "/tmp" and env.get("TMPDIR") == "/tmp"
        counters = cmd[3:cmd.index("--output-format")]
        d = cmd[cmd.index("-d") + 1]
        os.makedirs(os.path.join(d, "host"), exist_ok=True)
        calls.append(counters)
        with open(os.path.join(d, "host", "p_counter_collection.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
            for c in counters:
                val = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 500.0, "GRBM_GUI_ACTIVE": 8.0e6, "SQ_ACTIVE_INST_VALU": 2.56e8,
                       "SQ_INSTS_VALU": 1.28e8, "SQ_LDS_IDX_ACTIVE": 6.4e7, "SQ_LDS_BANK_CONFLICT": 2.56e7,
                       "SQ_VALU_MFMA_BUSY_CYCLES": 5.12e8}.get(c, 1.0)
                # the gathering kernel twice at full size (2 ms) and twice on the kept pairs (0.1 ms); the vote kernel twice
                for i, (name, dur, scale) in enumerate([(big, 2_000_000, 1.0), (big, 100_000, 0.1), (big, 2_000_000, 1.0),
                                                        (big, 100_000, 0.1), (small, 500_000, 1.0), (small, 500_000, 1.0)]):
                    # counters arrive split over several rows per dispatch (one per XCD / instance): two halves here
                    for half in (0, 1):
                        w.writerow([i + 1, name, c, val * scale / 2, 1_000_000 * i, 1_000_000 * i + dur])
    monkeypatch.setattr(FakePopen, "action", staticmethod(fake_run))
    monkeypatch.setattr(FakePopen, "rc", 0)
    monkeypatch.setattr(subprocess, "Popen", FakePopen)
    monkeypatch.setattr("shutil.which", lambda name: "/fake/rocprofv3")
    out = bench.collect_counters(["--scenes-per-gpu", "4"])
    assert "reason" not in out and [c[0] for c in calls] == ["FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU"]
    key = "reslayer_split_kernel<4, true, true, false, 3, 0>"
    assert set(out) == {key + "#large", key + "#small", "vote_center_persist_kernel<false>"}
    assert out[key + "#large"]["launches"] == 2 and out[key + "#small"]["launches"] == 2
    assert bench.hbm_bytes(out[key + "#large"]) == (2 * 1000.0 + 500.0) * 1024.0
    assert bench.hbm_bytes(out[key + "#small"]) == pytest.approx((2 * 100.0 + 50.0) * 1024.0)
    vc = out["vote_center_persist_kernel<false>"]
    act = bench.unit_activity(vc)
    # 8e6 GUI cycles over 8 XCDs = 1e6 shader cycles in 500 us = 2 GHz; VALU issue 2 cycles x 1.28e8 instructions / 1024 SIMDs / 1e6 =
    # 0.25, VALU busy 2 x 2.56e8 / 1024 / 1e6 = 0.5; LDS 6.4e7 / 256 / 1e6 = 0.25
    assert act["shader_clock_ghz"] == 2.0 and act["valu_issue"] == 0.25 and act["valu_busy"] == 0.5 and act["lds_busy"] == 0.25
    assert act["lds_bank_conflict"] == 0.1 and act["mfma_busy"] == 0.5 and act["invalid"] is None
    # a fraction above 1 is reported as invalid, not as attainment; a short kernel's cycles are capped at max clock x duration
    hot = dict(vc, SQ_INSTS_VALU=1.0e9)
    a2 = bench.unit_activity(hot)
    assert a2["valu_issue"] is None and a2["invalid"] == ["valu_issue"] and a2["valu_busy"] == 0.5
    tiny = dict(vc, GRBM_GUI_ACTIVE=8.0e7)         # 1e7 "cycles" in 500 us = 20 GHz: the window is longer than the launch
    a3 = bench.unit_activity(tiny)
    assert a3["shader_clock_ghz"] == 2.4 and a3["cycles_source"].startswith("MAX_CLOCK") and a3["valu_issue"] == pytest.approx(0.2083, abs=1e-4)
    # the CU count comes from the device properties
    bench.set_device_cus(128)
    assert bench.unit_activity(vc)["lds_busy"] == 0.5
    bench.COUNTERS.clear()
    bench.COUNTERS.update(out)
    assert bench.counter_entry("vote_center_persist_kernel") is vc and bench.counter_entry("nothing_kernel") is None
    assert bench.kernel_us("vote_center_persist_kernel") == 500.0 and bench.kernel_us("nothing_kernel") is None
    assert bench.hbm_bytes({"FETCH_SIZE": 1.0}) is None and bench.unit_activity({}) is None


def test_missing_profiler_or_failed_pass_gives_a_reason_not_a_number(bench, monkeypatch):
    monkeypatch.setattr("shutil.which", lambda name: None)
    monkeypatch.setattr(os.path, "exists", lambda p, _e=os.path.exists: False if p == "/opt/rocm/bin/rocprofv3" else _e(p))
    assert "rocprofv3 not found" in bench.collect_counters([])["reason"]
    monkeypatch.setattr("shutil.which", lambda name: "/fake/rocprofv3")
    monkeypatch.setattr(FakePopen, "action", None)
    monkeypatch.setattr(FakePopen, "rc", 3)
    monkeypatch.setattr(FakePopen, "err", b"boom")
    monkeypatch.setattr(subprocess, "Popen", FakePopen)
    out = bench.collect_counters([], passes=("FETCH_SIZE",))
    assert "rocprofv3 exit 3" in out["reason"] and "boom" in out["reason"]
    bench.COUNTERS.clear()
    bench.COUNTERS.update(out)
    assert bench.pmc_traffic_mlp() is None and bench.pmc_traffic_mlp(3, False) is None and bench.pmc_traffic("vote_center") is None
    assert bench.tuple_mlp_kernels(3, True)[0].endswith("3, 3>") and bench.tuple_mlp_kernels(2, True)[0].endswith("2, 0>#large")


def test_a_hanging_pass_is_stopped_as_a_whole_process_group(bench, monkeypatch, tmp_path):
    """on a timeout only rocprofv3 used to be killed; its profiled child kept the GPU busy under the timed loops.  Now the
    pass leads its own session and the whole group is terminated and reaped before collect_counters returns."""
    import time
    stub = tmp_path / "rocprofv3"
    pidfile = tmp_path / "pids"
    stub.write_text("#!%s\nimport os, subprocess, sys, time\n"
                    "c = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])\n"
                    "open(%r, 'w').write('%%d %%d' %% (os.getpid(), c.pid))\ntime.sleep(600)\n" % (sys.executable, str(pidfile)))
    stub.chmod(0o755)
    monkeypatch.setattr("shutil.which", lambda name: str(stub))
    t0 = time.time()
    out = bench.collect_counters([], passes=("FETCH_SIZE", "WRITE_SIZE"), timeout_s=3)
    assert time.time() - t0 < 30
    assert "timed out after 3 s" in out["reason"] and "WRITE_SIZE" not in out["reason"]      # later passes are not attempted
    parent, child = (int(x) for x in pidfile.read_text().split())
    for pid in (parent, child):
        for _ in range(50):                       # reaped / gone (a zombie of the grandchild is collected by init)
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                break
            try:
                if open("/proc/%d/stat" % pid).read().split()[2] == "Z":
                    break
            except OSError:
                break
            time.sleep(0.1)
        else:
            raise AssertionError("process %d of the timed-out pass is still running" % pid)
