"""CPU tests of round 6's bench additions: the oracle worker pool behind `oracle_agreement` (fresh `bench.py --oracle-worker`
children, results identical to the oracle run in-process), the C + OpenMP restatement of get_topk_dir's sphere-bin count (bit-equal
to the NumPy definition and to the reference's own counts in tests/golden/small.npz, for any thread count), the OpenMP mode of the C
SHOT oracle (identical outputs), the product-prior restatement, and the power / clock telemetry summary."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import cppf_oracle as O          # noqa: E402
from oracle import pipeline_oracle as PO     # noqa: E402
from oracle import shot_oracle as S          # noqa: E402
from oracle import vote_oracle as V          # noqa: E402


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_for_agreement_tests", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def small():
    return dict(np.load(os.path.join(GOLDEN, "small.npz")))


@pytest.mark.parametrize("name", ["up", "right"])
@pytest.mark.parametrize("threads", [1, 3, 0])
def test_c_sphere_counts_equal_numpy_definition_and_reference_golden(small, name, threads):
    vmask = small["small_%s_vmask" % name]
    ref = small["small_%s_cand" % name].reshape(-1, 3)                     # the reference's own candidates
    w = np.broadcast_to(small["small_imp_pair_wt"][vmask, None], (int(vmask.sum()), 36)).reshape(-1, 1)
    for bmm in (100000, 1000, 37):
        _, _, want = O.get_topk_dir(ref, small["sphere_pts"], bmm, 1.0, w, return_counts=True)
        d, c, got = O.get_topk_dir(ref, small["sphere_pts"], bmm, 1.0, w, topk=5, return_counts=True, impl="c", threads=threads)
        assert np.array_equal(got, want), (bmm, np.abs(got - want).max())
        if bmm == 100000:
            assert np.array_equal(got, small["small_%s_counts" % name])    # = eval.get_topk_dir's own output
            assert np.array_equal(c, small["small_%s_top5_counts" % name])


def test_c_sphere_counts_random_and_edge_cases():
    rng = np.random.RandomState(3)
    sph = O.sphere_bins(1.0)
    pred = rng.randn(20000, 3).astype(np.float32)
    pred /= np.linalg.norm(pred, axis=1, keepdims=True)
    pred[:50] = sph[rng.randint(0, len(sph), 50)]                           # exactly on bin centres
    wt = rng.uniform(0.5, 40.0, (20000, 1))
    for bmm in (100000, 8192, 8193, 1):
        if bmm == 1 and pred.shape[0] > 300:
            p_, w_ = pred[:300], wt[:300]
        else:
            p_, w_ = pred, wt
        _, _, want = O.get_topk_dir(p_, sph, bmm, 1.0, w_, return_counts=True)
        _, _, got = O.get_topk_dir(p_, sph, bmm, 1.0, w_, return_counts=True, impl="c", threads=2)
        assert np.array_equal(got, want)
    # float32 weights / no weights / no rows keep the NumPy path
    _, _, a = O.get_topk_dir(pred[:100], sph, 100000, 1.0, None, return_counts=True, impl="c")
    _, _, b = O.get_topk_dir(pred[:100], sph, 100000, 1.0, None, return_counts=True)
    assert np.array_equal(a, b)
    assert V.sphere_counts(np.zeros((0, 3), np.float32), sph, np.zeros((0,)), 0.5, 100).sum() == 0


def test_shot_oracle_threads_do_not_change_outputs():
    from cppf2_amd import synth
    pc = synth.make_scene(0, 3, 700)["pc"]
    a = S.compute_ex(pc, 0.02, 0.02, pcl_arithmetic=True, threads=1)
    b = S.compute_ex(pc, 0.02, 0.02, pcl_arithmetic=True, threads=0)
    c = S.compute_ex(pc, 0.02, 0.02, pcl_arithmetic=False, threads=3)
    d = S.compute_ex(pc, 0.02, 0.02, pcl_arithmetic=False, threads=1)
    for x, y in list(zip(a, b)) + list(zip(c, d)):
        assert np.array_equal(x, y, equal_nan=True)


def test_product_prior_is_bin_prior_dense(bench):
    import torch
    from cppf2_amd import ops, synth
    sc = synth.make_scene(0, 1, 256)
    idx = O.sample_tuples(0, 1, 500, 5, 256).astype(np.int64)
    got = bench.product_prior(sc["pc_canon"], idx)
    coords = torch.from_numpy(sc["pc_canon"])[torch.from_numpy(idx[:, :2]).reshape(-1)].reshape(-1, 6)
    pos = (coords.clamp(-0.5, 0.5) + 0.5) * 31.0                            # workloads.Step.__init__
    want = ops.BinPrior(pos.contiguous(), bench.PRIOR_INV_SIGMA).dense(32).numpy()
    assert got.dtype == np.float32 and np.array_equal(got, want)


def test_oracle_pool_results_equal_in_process_oracle(bench):
    import torch
    args = bench.launch.parse(["--points", "400", "--tuples", "1500", "--rots", "36", "--scenes-per-gpu", "3",
                               "--agreement-voxel-scenes", "1"])
    pool = bench.start_oracle_pool(args, 0, 1)
    assert len(pool.procs) >= 1 and len(pool.tasks) == 4
    out = pool.join(timeout=600)
    assert pool.error is None, pool.error
    assert sorted(out) == ["headline", "voxel2mm"] and sorted(out["headline"]) == [0, 1, 2] and sorted(out["voxel2mm"]) == [0]
    assert pool.extra["workers_agree_on_weights"] and pool.extra["workers"] == len(pool.procs)
    # the same scene in this process: identical records
    from cppf2_amd import synth
    w = bench.seeded_weights(0)
    assert bench.weights_sha(w) == pool.extra["worker_weights_sha"]["shot"]
    ang = torch.arange(36).float() / 36 * 2 * np.pi
    trig = (torch.cos(ang).numpy(), torch.sin(ang).numpy())
    sc = synth.make_scene(0, 1, 400)
    o = PO.run_scene_full(w, sc["pc"], 0, 1, 1500, res=2e-3, num_rots=36, trig=trig,
                          prior_fn=lambda idx: bench.product_prior(sc["pc_canon"], idx))
    r = out["headline"][1]
    assert r["argmax"] == o["argmax"] and r["up_idx"] == o["up_idx"] and r["right_idx"] == o["right_idx"]
    assert np.array_equal(r["T_est"], o["T_est"]) and np.array_equal(r["R_est"], o["R_est"])
    assert np.array_equal(r["bins"], o["bins"].astype(np.uint8)) and r["kept"] == int(o["pairs_mask"].sum())
    # rank != 0, a multi-rank run, --cpu-scenes 0 and counter children start no workers
    for argv, rank, world in ((["--cpu-scenes", "0"], 0, 1), ([], 1, 2), ([], 0, 2), (["--counter-child"], 0, 1), (["--agreement-scenes", "0"], 0, 1)):
        p = bench.start_oracle_pool(bench.launch.parse(argv), rank, world)
        assert not p.procs and p.join() == {}


def test_oracle_pool_reports_a_failed_worker(bench, monkeypatch):
    args = bench.launch.parse(["--points", "64", "--tuples", "100", "--rots", "36", "--scenes-per-gpu", "1", "--no-voxel-density"])
    monkeypatch.setattr(bench.OraclePool, "worker_cap", staticmethod(lambda: 1))
    pool = bench.OraclePool(args)
    pool.add("headline", "shot", "synthetic", [0])
    pool.start(seed=0, points=64, tuples=100, rots=36, prior=True, cos=[1.0], sin=[0.0], scene0=0, batch=1, desc_seed=17)   # 1-entry trig: the worker raises
    out = pool.join(timeout=300)
    assert out == {} and pool.error and "worker 0 rc" in pool.error


def test_telemetry_summary_is_windowed_and_robust():
    from cppf2_amd.benchlib import telemetry as T
    data = {"bdf": "0000:0d:00.0", "power_file": "power1_input", "power_cap_uw": 1400000000, "power_cap_max_uw": 1400000000,
            "freq_labels": {"freq1": "sclk"}, "error": None, "smi_error": None,
            "samples": [(t / 100.0, int((300 + (1000 if 1.0 <= t / 100.0 < 2.0 else 0)) * 1e6), int((2400 - (400 if 1.0 <= t / 100.0 < 2.0 else 0)) * 1e6))
                        for t in range(300)],
            "smi": [(0.95 + 0.1 * i, {"acc_ppt_pwr": 10 * i, "acc_counter": 100 * i, "per_ppt_pwr": 50, "active_ppt_pwr": True}, [2000 + i] * 8) for i in range(12)]}
    out = T.summarize(data, [("idle", 0.0, 0.9), ("loop", 1.0, 1.995), ("empty", 5.0, 6.0)])
    assert out["available"] and out["power_cap_w"] == 1400.0
    assert out["windows"]["idle"]["power_w"]["mean"] == 300.0 and out["windows"]["idle"]["sclk_mhz"]["max"] == 2400.0
    lp = out["windows"]["loop"]
    assert lp["power_w"]["mean"] == 1300.0 and abs(lp["power_frac_of_cap"]["mean"] - 1300 / 1400) < 1e-3 and lp["sclk_mhz"]["mean"] == 2000.0
    assert lp["throttle"]["acc_ppt_pwr"] > 0 and lp["throttle"]["active_ppt_pwr"] is True and lp["sclk_mhz_all_xcds"]["min"] >= 2000
    assert out["windows"]["empty"]["samples"] == 0 and "power_w" not in out["windows"]["empty"]
    assert T.summarize({"error": "no amdgpu hwmon directory for x", "bdf": "x"}, [])["available"] is False
    # without sysfs (this container) the object still reports why, and windows are harmless
    t = T.Telemetry.start()
    with t.window("w"):
        pass
    r = t.finish()
    assert isinstance(r, dict) and "available" in r
    json.dumps(r)


def test_telemetry_child_protocol_without_a_card():
    p = subprocess.run([sys.executable, "-m", "cppf2_amd.benchlib.telemetry", "--sample"], cwd=ROOT, input="card 0000:ff:00.0\nstop\n",
                       capture_output=True, text=True, timeout=60)
    assert p.returncode == 0
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["bdf"] == "0000:ff:00.0" and d["error"] and d["samples"] == []
