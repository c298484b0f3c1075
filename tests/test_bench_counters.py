"""bench.py's in-run counter passes, on a box without a GPU: the rocprofv3 child is replaced by a stub that writes the CSV rocprofv3
writes, and the parsing / normalisation (kernel keys, duration classes of a kernel launched at two sizes, HBM bytes = 2 x FETCH_SIZE
+ WRITE_SIZE KB, rocprof's VALUBusy / LDS / matrix-pipe fractions) is checked on known numbers."""
import csv
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    monkeypatch.setenv("PYTORCH_TUNABLEOP_ENABLED", "0")          # (bench.py sets up a TunableOp table on import otherwise)
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_kernel_key_strips_arguments_but_keeps_template_parentheses(bench):
    k = bench.kernel_key
    assert k("void reslayer_split_kernel<4, true, true, false, 3, 0>(float const*, long, int)") == "reslayer_split_kernel<4, true, true, false, 3, 0>"
    assert k("shot_eig_kernel(long, float const*, double const*)") == "shot_eig_kernel"
    assert k("void foo<(anonymous namespace)::Bar>(int)") == "foo<(anonymous namespace)::Bar>"
    assert k("__amd_rocclr_copyBuffer") == "__amd_rocclr_copyBuffer"


def test_counter_passes_are_parsed_and_normalised(bench, monkeypatch, tmp_path):
    calls = []
    big, small = "void reslayer_split_kernel<4, true, true, false, 3, 0>(float const*)", "void vote_center_persist_kernel<false>(int)"

    def fake_run(cmd, cwd=None, env=None, stdout=None, stderr=None, timeout=None):
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
                val = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 500.0, "GRBM_GUI_ACTIVE": 8.0e6, "SQ_ACTIVE_INST_VALU": 1.28e8,
                       "SQ_LDS_IDX_ACTIVE": 6.4e7, "SQ_LDS_BANK_CONFLICT": 2.56e7, "SQ_VALU_MFMA_BUSY_CYCLES": 5.12e8}.get(c, 1.0)
                # the gathering kernel twice at full size (2 ms) and twice on the kept pairs (0.1 ms); the vote kernel twice
                for i, (name, dur, scale) in enumerate([(big, 2_000_000, 1.0), (big, 100_000, 0.1), (big, 2_000_000, 1.0),
                                                        (big, 100_000, 0.1), (small, 500_000, 1.0), (small, 500_000, 1.0)]):
                    # counters arrive split over several rows per dispatch (one per XCD / instance): two halves here
                    for half in (0, 1):
                        w.writerow([i + 1, name, c, val * scale / 2, 1_000_000 * i, 1_000_000 * i + dur])

        class R:
            returncode = 0
            stderr = b""
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
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
    # 8e6 GUI cycles over 8 XCDs = 1e6 shader cycles in 500 us = 2 GHz; VALU 4 x 1.28e8 / 1024 / 1e6 = 0.5; LDS 6.4e7 / 256 / 1e6 = 0.25
    assert act["shader_clock_ghz"] == 2.0 and act["valu_busy"] == 0.5 and act["lds_busy"] == 0.25
    assert act["lds_bank_conflict"] == 0.1 and act["mfma_busy"] == 0.5
    bench.COUNTERS.clear()
    bench.COUNTERS.update(out)
    assert bench.counter_entry("vote_center_persist_kernel") is vc and bench.counter_entry("nothing_kernel") is None
    assert bench.hbm_bytes({"FETCH_SIZE": 1.0}) is None and bench.unit_activity({}) is None


def test_missing_profiler_or_failed_pass_gives_a_reason_not_a_number(bench, monkeypatch):
    monkeypatch.setattr("shutil.which", lambda name: None)
    monkeypatch.setattr(os.path, "exists", lambda p, _e=os.path.exists: False if p == "/opt/rocm/bin/rocprofv3" else _e(p))
    assert "rocprofv3 not found" in bench.collect_counters([])["reason"]

    class R:
        returncode = 3
        stderr = b"boom"
    monkeypatch.setattr("shutil.which", lambda name: "/fake/rocprofv3")
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: R())
    out = bench.collect_counters([], passes=("FETCH_SIZE",))
    assert "rocprofv3 exit 3" in out["reason"] and "boom" in out["reason"]
    bench.COUNTERS.clear()
    bench.COUNTERS.update(out)
    assert bench.pmc_traffic_mlp() is None and bench.pmc_traffic("vote_center") is None
