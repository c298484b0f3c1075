"""The N > 1 path on one GPU (scenes shard, one gather of the 160-byte records).

* records do not depend on the world size: scenes 0..7 as one batch == two 4-scene "ranks" (global scene id keys every
  RNG stream; byte-identical records);
* bench.py --gpus 2 as two fresh processes sharing GPU 0 (gloo), started by tests/conftest.py before this process
  touched the GPU: rank 0's JSON line reports both ranks' scenes and the gathered records.
Needs an MI355X: run with `pytest -m gpu`.
"""
import json
import os
import sys
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args(scenes_per_gpu, points=2048, tuples=8000, rots=90):
    return types.SimpleNamespace(scenes_per_gpu=scenes_per_gpu, points=points, tuples=tuples, rots=rots, seed=0,
                                 vote_mode=0, eager_scale_head=False)


def test_records_do_not_depend_on_the_world_size():
    from cppf2_amd.benchlib import workloads as bench
    dev = torch.device("cuda")
    whole = bench.Step(_args(8), 0, 1, dev)
    whole.run()
    want = whole.pipe.results_to_numpy()
    parts = []
    for rank in range(2):
        st = bench.Step(_args(4), rank, 2, dev)
        assert st.scene0 == 4 * rank
        st.run()
        parts.append(st.pipe.results_to_numpy())
    got = np.concatenate(parts)
    # every byte of every record, `scale` included: since round 3 the scale head too is the library's kernels (split matrix-core
    # layers + cppf_reslayer_tail), whose arithmetic per row does not depend on the row count or on which workgroup runs the row
    # -- no BLAS library picks a tiling by the batch size any more (rounds 1-2 allowed 1 ulp on `scale` for that reason)
    for f in got.dtype.names:
        assert np.array_equal(got[f], want[f], equal_nan=True), f
    assert got.tobytes() == want.tobytes()
    # the same block of scenes at the same batch size: byte-identical, whatever the world size says
    again = bench.Step(_args(4), 1, 4, dev)
    again.run()
    assert again.pipe.results_to_numpy().tobytes() == parts[1].tobytes()
    # and the poses are the synthetic ground truth's (teacher prior): the batch really ran
    for b in range(8):
        assert np.linalg.norm(want["t"][b] - whole.scenes[b]["t"]) < 5e-3


def test_two_streams_give_the_single_stream_records():
    """Consecutive steps alternating between two HIP streams (bench.py's two_streams loop), each pipeline with its own buffers: the
    records of both must be byte-identical to a sequential run -- the library keeps no global device state, every entry point
    takes its stream, the SHOT scratch buffer is per (device, stream).  (Regression guard for DESIGN.md section 11: the
    rotation-vote kernel once produced different votes when it shared a CU with the MLP workgroups of the other stream.)"""
    from cppf2_amd.benchlib import workloads as bench
    dev = torch.device("cuda")
    a = _args(16, points=4096, tuples=20000, rots=180)
    # the two pipelines hold DIFFERENT scene batches (like bench.py's headline loop since round 5): an aliased buffer would show
    steps = [bench.Step(a, 0, 1, dev), bench.Step(a, 0, 1, dev, scene_shift=16)]
    want = []
    for s in steps:
        s.run()
        torch.cuda.synchronize()
        want.append(s.pipe.results.clone())
    assert not torch.equal(want[0], want[1]) and steps[1].scene0 == 16
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    from cppf2_amd import ops
    for trial in range(4):
        # (trials 2, 3: the bench's batch-mode setting -- the persistent MLP launches leave one CU per shader engine to the
        # other stream, cppf_mlp_reserve_cus -- so the two pipelines' kernels really run side by side)
        ops.mlp_reserve_cus(ops.batch_mode_reserved_cus(dev) if trial >= 2 else 0)
        try:
            for i in range(6):
                with torch.cuda.stream(streams[i & 1]):
                    steps[i & 1].run()
            torch.cuda.synchronize()
        finally:
            ops.mlp_reserve_cus(0)
        for s, w in zip(steps, want):
            assert torch.equal(s.pipe.results, w), trial


def test_rotation_vote_beside_the_wide_mlp_kernels_of_another_stream():
    """The kernel-level form of the guard above (scratch/rot_race_probe3.py): both rotation votes launched while a second stream
    runs a 256-wide / a 256 -> 192 MLP kernel (one 448- / 416-register wavefront per SIMD) give the counts of the solo launch.
    Before the packed-float32 erratum forms were built out (profiles/r3_pk_op_sel_erratum.md) 10 launches of 10 differed."""
    from cppf2_amd.benchlib import workloads as bench
    from cppf2_amd import models, ops
    dev = torch.device("cuda")
    st = bench.Step(_args(64, points=4096, tuples=20000, rots=180), 0, 1, dev)
    st.run()
    torch.cuda.synchronize()
    pipe = st.pipe
    idx = ops.sample_tuples(4096, 20000, 5, 0, tuple(range(64)), dev)
    g = torch.Generator(device="cpu").manual_seed(1)
    w = [(torch.randn(n, k, generator=g) / k ** 0.5).to(dev) for n, k in ((256, 256), (256, 256), (192, 256), (192, 256), (192, 192))]
    x = torch.randn(400000, 256, device=dev)
    wide = (models.pack_split(w[0], None, w[1], 256), torch.zeros(256, device=dev))
    proj = (models.pack_split(w[2], w[3], w[4], 256), torch.zeros(192, device=dev), torch.zeros(192, device=dev))
    out192 = torch.empty(400000, 192, device=dev)
    hogs = {"256 -> 256": lambda: ops.reslayer_split(x, wide[0], wide[1], None, 256),
            "256 -> 192": lambda: ops.reslayer_split(x, proj[0], proj[1], proj[2], 192, out=out192)}
    pipe.rot_bins(st.pts, idx)
    torch.cuda.synchronize()
    want = pipe.counts.clone()
    side = torch.cuda.Stream()
    for name, hog in hogs.items():
        for rep in range(6):
            with torch.cuda.stream(side):
                hog()
            pipe.rot_bins(st.pts, idx)
            torch.cuda.synchronize()
            assert torch.equal(pipe.counts, want), (name, rep)


def _job(name):
    from conftest import bench_job
    return bench_job(name)


def test_bench_agreement_pool_telemetry_and_prior_variants():
    """The default run's round-6 legs on the GPU box at a small batch: `oracle_agreement` covers EVERY scene of the batch (worker
    pool) with all-equal records and the workers' weights identical to the GPU run's; the voxel-density loop has its own; the f16x2
    loop too; `roofline.power` carries per-window power / clock, the three MLP launch forms, the library GEMM yardstick and the
    power-limit verdict; value_no_prior / value_array_prior are measured; the CPU baseline states its stages' threads."""
    rc, out, err = _job("agreement")
    assert rc == 0, err[-3000:]
    j = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][0])
    assert j["ok"] and not j["problems"]
    ag = j["oracle_agreement"]
    assert ag["scenes"] == 4 == ag["centre_argmax_equal"] == ag["up_bin_equal"] == ag["right_bin_equal"] == ag["kept_count_equal"] == ag["translation_bit_equal"]
    assert ag["match_5deg5cm"] == 1.0 and ag["defects"] == [] and ag["weights_identical_to_the_gpu_run"] and ag["workers_agree_on_weights"]
    assert ag["bin_draws"] == 4 * 20000 * 6 and ag["bin_draws_differing"] <= 4 and "pool_error" not in ag
    va = j["voxel_density"]["oracle_agreement"]
    assert va["scenes"] == 2 == va["centre_argmax_equal"] == va["up_bin_equal"] == va["right_bin_equal"] and va["defects"] == []
    assert j["voxel_density"]["streams"] == 2 and j["voxel_density"]["records_identical_to_single_stream"] is True
    assert j["value_voxel_density_single_stream"] > 0 and j["value_no_prior"] > 0 and j["value_array_prior"] > 0
    fo = j["f16x2_agreement"]["oracle"]
    assert fo["scenes"] == 4 == fo["centre_argmax_equal"] and fo["defects"] == []
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["shot_single_thread_s_per_scene"] >= cb["shot_all_cores_s_per_scene"] > 0
    assert cb["all_cores_pool"]["scenes"] == 6 and "get_topk_dir (C + OpenMP, oracle/vote_oracle.c)" in cb["threads_per_stage"]
    pw = j["roofline"]["power"]
    if pw.get("available"):                     # (a box without readable hwmon files reports why instead)
        assert pw["power_cap_w"] > 0 and {"idle", "two_stream_loop", "single_stream_loop"} <= set(pw["windows"])
        assert len(pw["mlp_launches"]) == 3 and all(l_["ms"] > 0 for l_ in pw["mlp_launches"])
        assert [l_["op"] for l_ in pw["mlp_launches"]] == ["reslayer_split_encode", "reslayer_split", "reslayer_split_decode"]
        idle = pw["windows"]["idle"].get("power_w", {}).get("mean")
        for l_ in pw["mlp_launches"]:           # 1.5 s of back-to-back launches each: ~150 samples, far above idle power
            if idle is not None and l_.get("samples", 0) >= 20:
                assert l_["power_w"]["mean"] > idle + 100.0, l_
        assert pw["library_bf16_gemm"]["tflops"] > 100
        if "mlp_power_limit" in pw:
            assert isinstance(pw["mlp_power_limit"]["confirmed"], bool)
        f_ = j["roofline"].get("frac_executed_vs_library_bf16_gemm")
        assert f_ is None or 0.3 < f_ < 1.5
    else:
        assert pw.get("reason")


def test_bench_two_ranks_on_one_gpu():
    """Plain `python bench.py --gpus 2`: bench.py starts both ranks itself (fresh processes, gloo because they share GPU 0)."""
    rc, out, err = _job("two_ranks")
    assert rc == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints the line, rank 1 nothing
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 1 and j["scaling"] == "weak" and j["unit"] == "scenes/s"
    assert j["records_gathered"] == 8 and j["config"]["scenes_per_gpu"] == 4
    assert j["collective"]["backend"] == "gloo" and j["collective"]["world"] == 2 and j["collective"]["records_gathered"] == 8
    assert j["value"] > 0 and abs(j["value"] - 8 / (j["ms_per_step"] / 1e3)) < 1e-6 * j["value"]
    assert j["cpu_baseline"] is None                       # rank 0 at N = 1 only
    assert j["roofline"]["frac"] > 0 and "per_kernel" in j["roofline"] and j["roofline"]["pipeline_frac"] > 0
    assert j["pose_5deg5cm_vs_gt"] == 1.0


def test_bench_measures_its_counters_in_the_run():
    """roofline.traffic and the per-kernel unit activity come from rocprofv3 passes bench.py runs itself (fresh children, before
    it initialises the GPU) -- not from files under profiles/."""
    rc, out, err = _job("counters")
    assert rc == 0, err[-3000:]
    j = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][0])
    r = j["roofline"]
    # (the job ran alone on the GPU: tests/conftest.py runs the bench jobs one after another before the first test)
    assert j["ok"], j["problems"]
    assert j["counter_children_started"] == 3
    assert r["traffic"] is not None and r["traffic"] > 0, (r.get("hbm", {}).get("traffic_source"), err[-1500:])
    assert "counter passes of this run" in r["hbm"]["traffic_source"]
    # what `frac` means: algorithmic float32 flops over the bf16 pipe's peak; the executed fraction and the reading
    # against the f32-input matrix instruction beside it, as scalars of `roofline`
    assert r["bound"] == "mfma" and r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.0 < r["frac"] < r["frac_executed"] <= 1.0
    assert r["frac_executed"] == pytest.approx(6.0 * r["frac"], rel=0.05)          # 6 products per float32 product (+ K padding)
    assert r["frac_vs_f32_mfma_peak"] == pytest.approx(r["frac"] * 2500.0 / 157.0, rel=1e-6)
    # EVERY per-kernel fraction is a fraction
    fr = {k: e["frac"] for k, e in r["per_kernel"].items()}
    assert all(f is None or 0.0 < f <= 1.0 for f in fr.values()), fr
    assert sum(f is not None for f in fr.values()) >= 7, fr
    for k in ("shot_frames", "shot352", "vote_center", "rot_bins"):
        e = r["per_kernel"][k]
        assert e["bound"] in ("valu", "lds") and e["frac"] is not None and e["activity"]["invalid"] is None, (k, e)
        assert e["activity"]["valu_issue"] <= e["activity"]["valu_busy"] * 1.001 + 1e-3, e      # issue rate: a lower bound of busy
    vc = r["per_kernel"]["vote_center"]
    assert vc["work"]["votes_per_s"] > 0 and vc["pmc_MB"] is not None, vc
    # the tuple encode has no kernel of its own since round 5 (built inside the MLP's first launch); the streaming kernels' fractions
    # are on the kernel's own duration (rocprofv3 dispatch timestamps of this run), not on the HIP-event gap of the stage
    enc = r["per_kernel"]["encode_tuples"]
    assert enc["bound"] == "fused" and enc["frac"] is None and enc["kernel"] is None
    dec = r["per_kernel"]["decode_bins"]
    assert dec["bound"] == "hbm" and "rocprofv3 dispatch timestamps" in dec["frac_kind"] and 0 < dec["kernel_ms"] <= dec["event_ms"] * 1.5
    assert r["hbm"]["kernel"] in ("decode_bins", "sample_tuples")
    assert r["per_kernel"]["assemble_pose"]["bound"] == "latency"
    busy = [v.get("mfma_busy") for v in r["mfma_busy_per_launch"].values()]
    assert all(b is not None and 0.0 < b <= 1.0 for b in busy), r["mfma_busy_per_launch"]


def test_bench_eight_ranks_dry_run_equals_one_rank():
    """SURVEY 8e on the hardware there is: eight fresh rank processes (bench.py's own launcher) sharing GPU 0 over gloo, two
    scenes each; the 16 records gathered in global scene order are byte-identical to one rank's 16 (records_sha256)."""
    rc8, out8, err8 = _job("eight_ranks")
    assert rc8 == 0, err8[-3000:]
    rc1, out1, err1 = _job("one_rank_16")
    assert rc1 == 0, err1[-3000:]
    j8 = json.loads([ln for ln in out8.splitlines() if ln.startswith("{")][0])
    j1 = json.loads([ln for ln in out1.splitlines() if ln.startswith("{")][0])
    assert j8["n_gpus"] == 8 and j8["records_gathered"] == 16 and j8["collective"]["world"] == 8 and j8["config"]["scenes_per_gpu"] == 2
    assert j1["n_gpus"] == 1 and j1["records_gathered"] == 16
    assert j8["records_sha256"] == j1["records_sha256"]
    assert j8["pose_5deg5cm_vs_gt"] == 1.0 and j8["ok"] and j1["ok"]
    # the line of a multi-rank run diagnoses itself: backend, world, op, per-rank step times, the records' hash
    c8 = j8["collective"]
    assert c8["backend"] == "gloo" and c8["op"] == "all_gather" and c8["records_sha256"] == j8["records_sha256"]
    assert 0 < c8["ms_per_step_per_rank"]["min"] <= c8["ms_per_step_per_rank"]["max"] == pytest.approx(j8["ms_per_step"], rel=1e-6)
    # no rank of a multi-rank run starts profiler children (bench.py guards on world == 1), and says so
    assert j8["counter_children_started"] == 0 and j8["roofline"]["traffic"] is None
    assert "one rank only" in j8["roofline"]["hbm"]["traffic_source"]
    # rank 0 of eight was pinned to its run of the host cores (an eighth of them)
    cores = j8["host_cores_of_rank0"]
    assert cores is None or (cores["cores"] >= 1 and cores["first"] <= cores["last"])


def test_bench_eight_ranks_of_64_scenes_equal_one_rank_of_512():
    """The dry run at the geometry of the real 8-GPU job (BASELINE configs[3]: 512 scenes, 64 per rank): eight rank processes of 64
    scenes each sharing GPU 0 over gloo gather the 512 records one rank computes alone, byte for byte (records_sha256)."""
    rc8, out8, err8 = _job("eight_ranks_64")
    assert rc8 == 0, err8[-3000:]
    rc1, out1, err1 = _job("one_rank_512")
    assert rc1 == 0, err1[-3000:]
    j8 = json.loads([ln for ln in out8.splitlines() if ln.startswith("{")][0])
    j1 = json.loads([ln for ln in out1.splitlines() if ln.startswith("{")][0])
    assert j8["n_gpus"] == 8 and j8["records_gathered"] == 512 and j8["config"]["scenes_per_gpu"] == 64 and j8["collective"]["bytes_per_rank"] == 64 * 160
    assert j1["n_gpus"] == 1 and j1["records_gathered"] == 512
    assert j8["records_sha256"] == j1["records_sha256"] and j8["ok"] and j1["ok"]


def test_bench_gathers_through_rccl_in_a_one_rank_group():
    """The `nccl` branch of bench.py / cppf2_amd.dist on the hardware there is: init_process_group("nccl", device_id=...),
    all_gather_into_tensor of the device-resident records, barrier and all_reduce, world size 1.  bench.py itself asserts
    that the gathered records are byte-equal to the local ones."""
    rc, out, err = _job("rccl_one_rank")
    assert rc == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    c = j["collective"]
    assert c["backend"] == "nccl" and c["op"] == "all_gather_into_tensor" and c["world"] == 1
    assert c["records_gathered"] == 4 and c["gather_us"] > 0
    assert j["n_gpus"] == 1 and j["records_gathered"] == 4 and j["pose_5deg5cm_vs_gt"] == 1.0
    # the headline loop of that run was the two-stream batch mode with one CU per shader engine left to the other stream
    two = j["two_streams"]
    assert two["streams"] == 2 and two["records_identical_to_single_stream"] and two["pipelines_hold_different_scenes"]
    assert two["mlp_reserved_cus"] == torch.cuda.get_device_properties(0).multi_processor_count // 8


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 2` over RCCL with one visible GPU must not silently measure one rank."""
    from conftest import BENCH2
    rc, out, err = _job("refuse_two_gpus")
    if BENCH2.get("gpus", 1) >= 2:
        assert rc == 0 and json.loads([ln for ln in out.splitlines() if ln.startswith("{")][0])["n_gpus"] == 2
        return
    assert rc == 2 and "GPU(s) are visible" in err and not [ln for ln in out.splitlines() if ln.startswith("{")]


def test_batch_mode_object_alternates_streams_and_restores_the_grid():
    """cppf2_amd.pipeline.BatchMode (what bench.py's headline loop runs on): batches alternate between the states' own streams
    with the CU reservation in force inside the block and lifted after it; records equal the sequential run's."""
    from cppf2_amd.benchlib import workloads as bench
    from cppf2_amd.pipeline import BatchMode
    from cppf2_amd import ops, _lib
    dev = torch.device("cuda")
    a = _args(8, points=2048, tuples=6000, rots=90)
    steps = [bench.Step(a, 0, 1, dev), bench.Step(a, 0, 1, dev, scene_shift=8)]
    want = []
    for s in steps:
        s.run()
        torch.cuda.synchronize()
        want.append(s.pipe.results.clone())
    seen = []
    with BatchMode(steps, device=dev) as mode:
        assert mode.reserve_cus == ops.batch_mode_reserved_cus(dev) and len(mode.streams) == 2
        for i in range(6):
            with mode.next() as s:
                assert s is steps[i % 2] and torch.cuda.current_stream() == mode.streams[i % 2]
                seen.append(torch.cuda.current_stream().cuda_stream)
                s.run()
    torch.cuda.synchronize()              # (leaving the block made the calling stream wait for both)
    assert len(set(seen)) == 2
    for s, w in zip(steps, want):
        assert torch.equal(s.pipe.results, w)
    with pytest.raises(AssertionError):
        mode.next()                       # not active any more
