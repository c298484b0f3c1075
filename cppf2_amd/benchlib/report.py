"""What bench.py prints: the JSON line of each workload (rank 0) -- the driver's contract fields, `roofline`, `cpu_baseline`,
`collective` -- and the arithmetic behind the roofline figures (algorithmic bytes per stage; MLP flops).

`roofline` of the headline describes the dominant kernel, the tuple MLP (matrix-core bound).  Its `frac` is ALGORITHMIC work over
time over the pipe's peak:  2 M K N float32 flops of the network as the reference writes it / launch time / the dense bf16 MFMA peak
(the pipe the kernel runs on).  The emulation overhead -- 6 bf16 products per float32 product -- is NOT counted as useful work there;
`frac_executed` is the pipe's utilisation including it, `frac_vs_f32_mfma_peak` the same algorithmic rate against the peak of the
f32-input matrix instruction (a value above 1 means: faster than the dtype's own pipe could go)."""
import json
import os
import time

import numpy as np

from . import counters as C
from .counters import COUNTERS, STAGE_BOUND, STAGE_KERNEL, counter_entry, hbm_bytes, kernel_us, pmc_traffic, unit_activity
from .workloads import DenseStep, EnsembleStep, Step

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak (no sparsity)
F32_MFMA_PEAK_TFLOPS = 157.0    # f32-input MFMA: 256 CUs x 4 SIMDs x 64 flops/cycle x 2.4 GHz


GATHERED_TUPLES = False         # set by main(): the encode stage writes pair features + indices only
FUSED_DRAW = False              # set by main(): the bins are drawn inside the MLP's output layer, the decode stage starts from them
ENCODE_FOLDED = False           # set by main(): the pair features are built inside the tuple MLP's first launch (no encode kernel)


def algorithmic_bytes(stage, B, N, T, R, S, G):
    """Compulsory bytes one launch of the stage's kernel moves for B scenes (SURVEY.md 8d per-scene figures)."""
    if stage == "encode_tuples" and ENCODE_FOLDED:
        return (T * 5 * 4 + N * 12 + N * 12) * B          # indices, points, normals in -- read by the MLP's first launch; nothing out
    if stage == "encode_tuples" and GATHERED_TUPLES:
        return (T * 5 * 4 + N * 12 + N * 12 + T * 40 * 4 + T * 5 * 4) * B
    if stage == "decode_bins" and FUSED_DRAW:
        return (T * 6 * 4 + T * 8 + N * 12 + T * (12 + 24 + 4 + 24)) * B
    per_scene = {
        "sample_tuples": T * 5 * 4,
        "shot_frames": N * 12 * 2 + N * 12 + N * 19 * 8 * 2 + N * 56,   # points in+sorted, normals out, sums w+r, frames
        "shot352": N * 12 + N * 12 + N * 56 + N * 352 * 4,
        "encode_tuples": T * 5 * 4 + N * 12 + N * 12 + N * 64 * 4 + T * 360 * 4,
        "decode_bins": 2 * T * 6 * 32 * 4 + T * 6 * 4 + T * 8 + T * (8 + 12 + 24 + 4 + 24),   # logits + prior read
        # idx + tr + points, grid clear G*4, one 4-byte accumulator update per vote (V = T*R), argmax
        # read G*4.  (This implementation keeps the accumulator in LDS slabs, so its HBM traffic -- `traffic` -- is
        # well below this figure: the frames workspace and the per-slab re-reads of it.)
        "vote_center": T * 8 + T * 8 + N * 12 + G * 4 + T * R * 4 + G * 4,
        "vote_frames": T * 8 + T * 8 + N * 12 + T * 48,
        "backvote_filter": T * 8 + T * 8 + N * 12 + T * (1 + 4 + 4 + 8 + 4),
        "rot_bins": 2 * (T // 10) * (4 + 8 + 4 + 8 + 12) + 2 * S * 4,
        "assemble_pose": 160,
    }
    return per_scene.get(stage, 0) * B


def tuple_mlp_flops(model, B, T, N, nprod):
    """(executed MFMA flops, algorithmic float32 flops) of one pass' tuple MLP launches (the three reslayer_split launches; for
    the DINO model also the two per-point Linear launches), K padded to 16 in the executed figure."""
    def layer(k, n, proj):
        return 2.0 * n * ((k + 15) // 16 * 16) * (2 if proj else 1) + 2.0 * n * n, 2.0 * n * k * (2 if proj else 1) + 2.0 * n * n
    tail = [(128, 128, False)] * 4 + [(128, 256, True), (256, 256, False), (256, 256, False), (256, 192, True)]
    if model == "shot":
        layers = [(360, 128, True)] + tail
        ex = sum(layer(*l)[0] for l in layers) * B * T
        al = sum(layer(*l)[1] for l in layers) * B * T
        return nprod * ex, al
    # DINO: the first layer's products run over the 30 (32) coordinate columns per tuple; its descriptor columns (and
    # desc_pair_transform) are the per-point slot tables, desc_transform the per-point Linear in front of them
    layers = [(30, 128, True)] + tail
    ex = sum(layer(*l)[0] for l in layers) * B * T + (2.0 * 1024 * 256 + 2.0 * 256 * 1280) * B * N
    # algorithmic = the reference's network as written: 286-column rows, desc_transform on k gathered descriptors per tuple,
    # desc_pair_transform over their concatenation (train_dino.py:95-96)
    al = (sum(layer(*l)[1] for l in [(286, 128, True)] + tail) + 5 * 2.0 * 1024 * 256 + 2.0 * 1280 * 256) * B * T
    return nprod * ex, al


def mfma_roofline(executed, algorithmic, ms, nprod, traffic, launches, kernel="tuple_mlp", **extra):
    """The `roofline` object of a matrix-core-bound stage.  frac = ALGORITHMIC float32 flops / time / dense bf16 peak (the pipe the
    kernel runs on); the operand-splitting overhead is not counted as useful work.  frac_executed = the bf16 (fp16) MFMA work the
    kernel issues / time / the same peak = the pipe's utilisation; frac_vs_f32_mfma_peak = the algorithmic rate against the peak of
    the f32-input matrix instruction (157 TFLOP/s; above 1 = faster than the arithmetic type's own pipe)."""
    sec = ms / 1e3
    r = dict(bound="mfma", kernel=kernel, kernel_name="reslayer_split_kernel",
             achieved=algorithmic / 1e12 / sec, peak=BF16_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
             frac=algorithmic / 1e12 / sec / BF16_MFMA_PEAK_TFLOPS,
             frac_executed=executed / 1e12 / sec / BF16_MFMA_PEAK_TFLOPS,
             frac_vs_f32_mfma_peak=algorithmic / 1e12 / sec / F32_MFMA_PEAK_TFLOPS,
             frac_kind="algorithmic float32 flops (2 M K N per Linear) / launch time / dense bf16 MFMA peak",
             achieved_executed=executed / 1e12 / sec,
             executed_kind="bf16 MFMA flops issued: 6 exact products per float32 product, K padded to 16" if nprod == 6.0
                           else "fp16 MFMA flops issued: 3 products per float32 product, K padded to 16",
             f32_input_mfma_peak_tflops=F32_MFMA_PEAK_TFLOPS, traffic=traffic, launch_ms=ms, launches=launches)
    r.update(extra)
    return r


def pose_ok_vs_gt(rec, scenes):
    """Scenes whose recovered centre / up axis are within 5 cm / 5 degrees of the synthetic ground truth."""
    ok = 0
    for b, sc in enumerate(scenes):
        terr = np.linalg.norm(rec["t"][b] - sc["t"])
        cosang = abs(float(rec["R"][b][:, 1] @ sc["R"][:, 1]))
        ok += int(terr < 0.05 and np.degrees(np.arccos(min(cosang, 1.0))) < 5.0)
    return ok


def collective_info(backend, world, B, records, gather_ms, rank_ms=None, records_sha256=None):
    """The path's one collective as the line reports it -- with what a first multi-GPU run needs to diagnose itself:
    backend, world, the op, bytes per rank, the gather's HIP-event time, per-rank step times (min / max over ranks) and the SHA-256
    of the gathered records (equal for every world size)."""
    return {"backend": backend or "none (one rank: the local records are the result)", "world": world,
            "op": "all_gather_into_tensor" if backend == "nccl" else ("all_gather" if backend else None),
            "records_gathered": int(records), "bytes_per_rank": int(B * 160), "gather_us": round(1e3 * gather_ms, 2),
            "ms_per_step_per_rank": rank_ms, "records_sha256": records_sha256}


def stage_means(evs):
    out = {}
    for ev in evs:
        for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
            out[n1] = out.get(n1, 0.0) + e0.elapsed_time(e1) / len(evs)
    return out


def report_dense(args, step, dt, evs, world, backend, rank_ms=None):
    """The JSON line of --workload dense64k (rank 0): contract fields, per-stage times, the two extension kernels' rooflines."""
    from cppf2_amd import models as _models
    B, N, T, R = step.B, step.N, step.T, args.rots
    stage_ms = stage_means(evs)
    rec = step.pipe.results_to_numpy()
    all_rec = step.pipe.results_to_numpy(step.all_records)
    assert all_rec.shape[0] == B * world and np.array_equal(all_rec[:B].tobytes(), rec.tobytes())
    ok = pose_ok_vs_gt(rec, step.scenes)
    nprod = 6.0 if _models.MLP_ARITH == "split" else 3.0
    ex, al = tuple_mlp_flops("shot", B, T, N, nprod)
    enc_bytes = B * (T * 5 * 4 + N * 24 + N * 64 * 2 + T * 360 * 4)                 # indices + points + normals + f16 table in, rows out
    enc_ms, vc_ms, mlp_ms = stage_ms["encode_tuples_f16"], stage_ms["vote_center_weighted"], stage_ms["tuple_mlp"]
    enc_c = counter_entry("encode_shot_f16_kernel")
    vc_act = unit_activity(counter_entry("vote_center_persist_kernel<true>"))
    total = B * world * args.steps
    line = {
        "metric": "scenes/sec at %d pts x %d pairs (BASELINE configs[4]: dense pairs, fp16 table, weighted votes)" % (N, T),
        "value": total / dt, "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: dense pairs -- %d scenes/GPU x %d pts x %d tuples x %d rots, SHOT model, float16 "
                               "per-point feature table (tuple rows materialised from it), per-pair vote weights in [0, 4] on the "
                               "fixed-point centre accumulator (extensions the reference does not have); random-init weights + teacher "
                               "prior; MLP arithmetic %s; one HIP stream" % (B, N, T, R, _models.MLP_ARITH),
                   "scenes_per_gpu": B, "points": N, "tuples": T, "rots": R, "parallelism": "scene-sharded x%d" % world},
        "pairs_per_s": total * T / dt,
        "roofline": mfma_roofline(ex, al, mlp_ms, nprod, None, 3, algorithmic_f32_flops_per_step=al, executed_flops_per_step=ex,
                         per_kernel={
                             "encode_tuples_f16": dict(kernel="encode_shot_f16_kernel", bound="hbm", ms=round(enc_ms, 4), alg_MB=round(enc_bytes / 1e6, 1),
                                                       alg_GBs=round(enc_bytes / 1e9 / (enc_ms / 1e3), 1),
                                                       frac=round(enc_bytes / 1e9 / (enc_ms / 1e3) / HBM_PEAK_GBS, 4),
                                                       pmc_MB=None if hbm_bytes(enc_c) is None else round(hbm_bytes(enc_c) / 1e6, 1)),
                             "vote_center_weighted": dict(kernel="vote_center_persist_kernel<true>", bound="valu", ms=round(vc_ms, 4),
                                                          frac=None if vc_act is None else vc_act["valu_issue"],
                                                          frac_kind="VALU instruction issue rate / the SIMD-32 peak (counters.unit_activity)",
                                                          activity=vc_act,
                                                          work={"votes_per_launch": B * T * R, "votes_per_s": B * T * R / (vc_ms / 1e3)}),
                         },
                         counters=("this run's rocprofv3 passes" if "reason" not in COUNTERS else "null: " + str(COUNTERS.get("reason"))),
                         per_stage_ms={s_: round(stage_ms.get(s_, 0.0), 4) for s_ in DenseStep.STAGES}),
        "cpu_baseline": None, "cpu_baseline_note": "the extensions have no reference path to time; their oracle restatements are "
                                                   "checked in tests/test_gpu_parity.py",
        "pose_5deg5cm_vs_gt": ok / B,
        "collective": collective_info(backend, world, B, all_rec.shape[0], stage_ms.get("gather", 0.0), rank_ms,
                                      __import__("hashlib").sha256(all_rec.tobytes()).hexdigest()),
        "ok": True, "problems": [],
    }
    return line


def report_ensemble(args, step, dt, evs, world, backend, cpu_fn=None, rank_ms=None):
    """The JSON line of --workload ensemble (rank 0): the contract fields + roofline + cpu_baseline, per model.  cpu_fn(args, step,
    both, pick) -> (cpu_baseline, oracle_agreement): the CPU leg lives in bench.py (the only importer of oracle/)."""
    from cppf2_amd import models as _models
    B, N, T, R = step.B, step.N, step.T, args.rots
    stage_ms = stage_means(evs)
    step_times = sorted(ev[0][1].elapsed_time(ev[-1][1]) for ev in evs)
    shared = ["sample_tuples", "shot_frames", "shot352"]
    dino_ms = stage_ms.get("dino_point_transforms", 0.0) + sum(stage_ms.get("dino_" + n, 0.0) for n in EnsembleStep.PASS)
    shot_ms = stage_ms.get("shot_encoder", 0.0) + sum(stage_ms.get("shot_" + n, 0.0) for n in EnsembleStep.PASS)
    shared_ms = sum(stage_ms.get(n, 0.0) for n in shared)
    nprod = 6.0 if _models.MLP_ARITH == "split" else 3.0
    ex_d, al_d = tuple_mlp_flops("dino", B, T, N, nprod)
    ex_s, al_s = tuple_mlp_flops("shot", B, T, N, nprod)
    mlp_ms = stage_ms["dino_point_transforms"] + stage_ms["dino_tuple_mlp"] + stage_ms["shot_tuple_mlp"]
    rec = step.pipe.results_to_numpy(step.pipe.selected)
    all_rec = step.pipe.results_to_numpy(step.all_records)
    assert all_rec.shape[0] == B * world and np.array_equal(all_rec[:B].tobytes(), rec.tobytes())
    both = [step.pipe.results_to_numpy(step.pipe.result_slots[m]) for m in (0, 1)]
    ok = pose_ok_vs_gt(rec, step.scenes)
    pick = rec["pad_"][:, 0]
    cpu, agree = (None, None)
    if cpu_fn is not None and args.cpu_scenes > 0 and world == 1:
        cpu, agree = cpu_fn(args, step, both, pick)
    total = B * world * args.steps
    line = {
        "metric": "instances/sec, every instance through BOTH models (DINO + SHOT ensemble, BASELINE configs[2]) at %d pts x %d tuples" % (N, T),
        "value": total / dt, "unit": "instances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: DINO + SHOT ensemble, every instance voted by BOTH models (eval.py:219-372), %d "
                               "instances/GPU x %d pts x %d tuples x %d rots, 720 sphere bins, res 2 mm, bottle axes (y-only "
                               "alignment loss); random-init weights + teacher prior; DINOv2 descriptors = seeded unit vectors "
                               "[N, 1024] resident in HBM (inputs of the path); MLP arithmetic %s; one scene = one instance "
                               "through both models" % (B, N, T, R, _models.MLP_ARITH),
                   "scenes_per_gpu": B, "points": N, "tuples": T, "rots": R, "parallelism": "scene-sharded x%d" % world},
        "step_ms_sampled": {"min": round(step_times[0], 4), "median": round(step_times[len(step_times) // 2], 4),
                            "max": round(step_times[-1], 4), "n": len(step_times)},
        "per_model_ms": {"shared (sampler, normals + SHOT352)": round(shared_ms, 4),
                         "dino (point transforms .. alignment loss)": round(dino_ms, 4),
                         "shot (point encoder .. alignment loss)": round(shot_ms, 4),
                         "dino_over_shot": round(dino_ms / shot_ms, 4) if shot_ms > 0 else None},
        "picked": {"dino": int((pick == 0).sum()), "shot": int((pick == 1).sum()), "none": int((pick < 0).sum())},
        "roofline": mfma_roofline(ex_d + ex_s, al_d + al_s, mlp_ms, nprod, C.ensemble_mlp_traffic(), 8,
                         kernel="tuple_mlp (both models) + the DINO model's per-point Linear launches",
                         traffic_source=("rocprofv3 counter passes of this run: HBM bytes (2 x FETCH_SIZE + WRITE_SIZE) of every "
                                         "reslayer_split_kernel launch of a step" if "reason" not in COUNTERS
                                         else "null: " + str(COUNTERS.get("reason"))),
                         executed_flops_per_step={"dino": ex_d, "shot": ex_s},
                         algorithmic_f32_flops_per_step={"dino (the reference's row form)": al_d, "shot": al_s},
                         per_stage_ms={s_: round(stage_ms.get(s_, 0.0), 4) for s_ in EnsembleStep.STAGES}),
        "cpu_baseline": cpu, "oracle_agreement": agree, "pose_5deg5cm_vs_gt": ok / B,
        "collective": collective_info(backend, world, B, all_rec.shape[0], stage_ms.get("gather", 0.0), rank_ms,
                                      __import__("hashlib").sha256(all_rec.tobytes()).hexdigest()),
        "two_streams": getattr(step, "two", None),
        "value_single_stream": (step.two or {}).get("value_single_stream") if getattr(step, "two", None) else None,
        "ok": not (getattr(step, "two", None) and not step.two["records_identical_to_single_stream"]),
        "problems": (["two_streams: records differ from the single-stream ones"]
                     if (getattr(step, "two", None) and not step.two["records_identical_to_single_stream"]) else []),
    }
    return line


def per_kernel_table(stage_ms, B, N, T, R, S, G, stages):
    """roofline.per_kernel: every library stage with the bound that limits ITS kernel.
    * streaming kernels ("hbm"): algorithmic bytes / the KERNEL's duration / 8 TB/s.  The duration is rocprofv3's begin-to-end time of
      the dispatch in this run's FETCH_SIZE pass when the passes ran (the HIP-event gap of the stage also holds the launch gap, which
      is a fifth of a 0.1 ms kernel: `event_ms` keeps it); the counters' bytes sit beside the algorithmic ones;
    * descriptor / voting kernels ("valu" | "lds"): the busier unit's fraction from this run's SQ pass (counters.unit_activity: VALU
      = instruction issue rate against the SIMD-32's peak, never above 1) + their work in the domain's units per second;
    * latency-bound ones (one workgroup per scene, dependent phases): no fraction."""
    rows, per_kernel = [], {}
    for s in Step.STAGES:
        ms = stage_ms.get(s, 0.0)
        ab = algorithmic_bytes(s, B, N, T, R, S, G)
        gbs = (ab / 1e9) / (ms / 1e3) if ms > 0 and ab else 0.0
        rows.append((s, ms, ab / 1e6, gbs))
        if s not in stages:
            continue
        tr = pmc_traffic(s)
        act = unit_activity(counter_entry(STAGE_KERNEL.get(s, "")))
        e = dict(kernel=STAGE_KERNEL.get(s), ms=round(ms, 4), alg_MB=round(ab / 1e6, 2),
                 pmc_MB=None if tr is None else round(tr / 1e6, 2), activity=act)
        kind = STAGE_BOUND.get(s, "latency")
        if s == "encode_tuples" and ENCODE_FOLDED:
            e.update(bound="fused", frac=None, kernel=None,
                     frac_kind="no kernel: the pair features are built by the tuple MLP's first launch (cppf_reslayer_split_encode) "
                               "from the sampler's indices; the stage's event gap is host time between two launches")
        elif kind == "hbm":
            kus = kernel_us(STAGE_KERNEL.get(s, ""))
            kms = kus / 1e3 if kus else ms
            kgbs = (ab / 1e9) / (kms / 1e3) if kms > 0 and ab else 0.0
            e.update(bound="hbm", kernel_ms=round(kms, 4), event_ms=round(ms, 4), alg_GBs=round(kgbs, 1), frac=round(kgbs / HBM_PEAK_GBS, 4),
                     frac_kind="algorithmic bytes / kernel duration (%s) / 8 TB/s"
                               % ("rocprofv3 dispatch timestamps, this run's FETCH_SIZE pass" if kus else "HIP-event gap of the stage: no counter pass"),
                     frac_on_event_gap=round(gbs / HBM_PEAK_GBS, 4),
                     pmc_frac=None if (tr is None or kms <= 0) else round(tr / 1e9 / (kms / 1e3) / HBM_PEAK_GBS, 4))
        elif kind == "unit":
            if act is not None and act.get("valu_issue") is not None and act.get("lds_busy") is not None:
                unit, key = ("valu", "valu_issue") if act["valu_issue"] >= act["lds_busy"] else ("lds", "lds_busy")
                e.update(bound=unit, frac=act[key],
                         frac_kind=("vector instructions issued / the SIMD-32s' issue capacity over the launch's shader cycles"
                                    if unit == "valu" else "LDS-array cycles / shader cycles of the launch") + " (SQ counters of this run)")
            else:
                e.update(bound="valu", frac=None, frac_kind="no valid SQ counter pass in this run (%s)"
                                                            % (COUNTERS.get("reason") or (act or {}).get("invalid") or "--no-counters"))
            sec = ms / 1e3 if ms > 0 else float("nan")
            if s == "vote_center":
                e["work"] = {"votes_per_launch": B * T * R, "votes_per_s": B * T * R / sec}
            elif s == "rot_bins":
                tf = T // 10
                e["work"] = {"candidates_per_launch": 2 * B * tf * R, "candidates_per_s": 2 * B * tf * R / sec,
                             "exhaustive_equivalent_compare_accumulates_per_s": 2.0 * B * tf * R * S / sec,
                             "note": "the lookup table tests <= 8 bins per candidate (0.46 on average) where the "
                                     "reference's mm tests all %d" % S}
            elif s in ("shot_frames", "shot352"):
                e["work"] = {"points_per_launch": B * N, "points_per_s": B * N / sec}
        else:
            e.update(bound="latency", frac=None, frac_kind="one workgroup per scene / dependent phases: neither a "
                                                           "bandwidth nor an issue bound applies")
        per_kernel[s] = e
    return rows, per_kernel


def shot_mlp_layers(eager_scale_head):
    layers = [(360, 128, True)] + [(128, 128, False)] * 4 + [(128, 256, True), (256, 256, False), (256, 256, False), (256, 192, True)]
    if eager_scale_head:           # the scale head's matrix-core layers run inside the stage too (on every tuple)
        layers += [(256, 128, True), (128, 64, True)]
    return layers


def layer_flops(k, n, proj):
    """(executed per product with K padded to 16, algorithmic) flops per row of one ResLayer (train_shot.py:19-45)."""
    return 2.0 * n * ((k + 15) // 16 * 16) * (2 if proj else 1) + 2.0 * n * n, 2.0 * n * k * (2 if proj else 1) + 2.0 * n * n


def report_shot(run):
    """The JSON line of the headline workload (rank 0).  `run` = the measurements of bench.main(): args, step, world, backend, dt,
    dt_single, dt_other, dt_native, dt_f16, f16_agreement, two, intervals, intervals_single, stage_ms (kernel durations: single-stream
    loop), stage_ms_2s, step_times, affinity, cpu (cpu_baseline, oracle_agreement), evidence, voxel, rank_ms.  Returns (line, problems)."""
    from cppf2_amd import models as _models
    args, step, world, backend = run["args"], run["step"], run["world"], run["backend"]
    dt, dt_single, stage_ms, stage_ms_2s = run["dt"], run["dt_single"], run["stage_ms"], run["stage_ms_2s"]
    B, N, T, R, S = step.B, step.N, step.T, args.rots, step.pipe.S
    res = step.pipe.results_to_numpy()
    all_rec = step.pipe.results_to_numpy(step.all_records)
    assert all_rec.shape[0] == B * world and np.array_equal(all_rec[:B].tobytes(), res.tobytes())
    G = int(np.mean(res["ncell"]))
    mlp_stages = ("shot_encoder", "tuple_mlp", "scale_head")
    hip_stages = [s for s in Step.STAGES if s not in mlp_stages and s != "gather"]
    # the HBM roofline object describes the longest of the kernels that ARE bandwidth-bound (section 4 of DESIGN.md); the
    # voting and descriptor kernels (VALU / LDS bound) have their fractions in per_kernel
    hbm_bound = ("decode_bins", "sample_tuples") if ENCODE_FOLDED else ("decode_bins", "encode_tuples", "sample_tuples")
    dominant = max(hbm_bound, key=lambda s: stage_ms.get(s, 0.0))
    rows, per_kernel = per_kernel_table(stage_ms, B, N, T, R, S, G, hip_stages)
    dk = per_kernel[dominant]
    dom_ms = dk["kernel_ms"]
    dom_bytes = algorithmic_bytes(dominant, B, N, T, R, S, G)
    achieved = (dom_bytes / 1e9) / (dom_ms / 1e3)
    hip_only_ms = sum(stage_ms.get(s, 0.0) for s in hip_stages)
    path_bytes = sum(algorithmic_bytes(s, B, N, T, R, S, G) for s in hip_stages)
    step_ms = 1e3 * (dt_single if dt_single is not None else dt) / args.steps      # the step the stage times belong to
    counters_src = ("rocprofv3 counter passes of this run (2 x FETCH_SIZE + WRITE_SIZE, separate passes, fresh child processes)"
                    if "reason" not in COUNTERS else "null: " + str(COUNTERS.get("reason")))
    roofline = dict(bound="hbm", kernel=dominant, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=achieved / HBM_PEAK_GBS, traffic=pmc_traffic(dominant), launch_ms=dom_ms, launch_ms_kind=dk["frac_kind"],
                    kernel_name=STAGE_KERNEL.get(dominant), algorithmic_bytes_per_launch=dom_bytes,
                    algorithmic_model=("compulsory bytes of the stage as it runs here x %d scenes per launch (benchlib/report.py:"
                                       "algorithmic_bytes): SURVEY.md 8d's per-scene figures, except the tuple encode -- folded into "
                                       "the tuple MLP's first launch (indices + points + normals in, nothing out) or, with "
                                       "--separate-encode, in its gathered form (40 pair features + 5 global indices per tuple out: "
                                       "4.1 MB per scene instead of 8d's 30.35 MB of rows) -- and the decode behind the fused bin "
                                       "draw (bins in, vote parameters out)" % B),
                    traffic_source=counters_src,
                    # the whole path's algorithmic bytes (HIP stages; ~54 MB/scene) over the whole step
                    # (MLP included) and over the HIP stages alone, as fractions of the HBM peak
                    pipeline_bytes_per_step=path_bytes,
                    pipeline_frac=(path_bytes / 1e9) / (step_ms / 1e3) / HBM_PEAK_GBS,
                    hip_only_ms=hip_only_ms,
                    hip_only_frac=(path_bytes / 1e9) / (hip_only_ms / 1e3) / HBM_PEAK_GBS if hip_only_ms > 0 else None,
                    hip_only_scenes_per_s=B * world / (hip_only_ms / 1e3) if hip_only_ms > 0 else None,
                    mlp_ms=sum(stage_ms.get(s, 0.0) for s in mlp_stages),
                    per_kernel=per_kernel,
                    per_stage_ms={s: round(stage_ms.get(s, 0.0), 4) for s in Step.STAGES},
                    per_stage_ms_source=("the single-stream loop of this run (kernel durations; with two streams a stage's "
                                         "event time also counts the waits for the other stream's kernels: "
                                         "per_stage_ms_two_streams)" if dt_single is not None else "the headline loop"),
                    per_stage_ms_two_streams=({s: round(stage_ms_2s.get(s, 0.0), 4) for s in Step.STAGES}
                                              if dt_single is not None else None))
    if _models.MLP_ARITH in ("split", "split16"):
        # The dominant kernel of the step is the tuple MLP (cppf_reslayer_split, 3 launches back to back: the stage time is their
        # sum): matrix-core bound -> the line's `roofline`; the longest bandwidth-bound kernel's roofline stays under "hbm".
        nprod = 6.0 if _models.MLP_ARITH == "split" else 3.0
        pcs = 3 if nprod == 6.0 else 2
        layers = shot_mlp_layers(args.eager_scale_head)
        executed = nprod * sum(layer_flops(*l)[0] for l in layers) * B * T
        algorithmic = sum(layer_flops(*l)[1] for l in layers) * B * T
        hbm = {k_: roofline[k_] for k_ in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "launch_ms_kind",
                                           "kernel_name", "algorithmic_bytes_per_launch", "algorithmic_model", "traffic_source")}
        busy = {}
        for k_ in C.tuple_mlp_kernels(pcs, ENCODE_FOLDED):
            busy[k_] = unit_activity(COUNTERS.get(k_) or COUNTERS.get(k_.replace("#large", ""))) or {}
        roofline.update(mfma_roofline(
            executed, algorithmic, stage_ms["tuple_mlp"], nprod, C.pmc_traffic_mlp(pcs, ENCODE_FOLDED), 3,
            traffic_covers="the same 3 launches as launch_ms (counter passes of this run: 2 x FETCH_SIZE + WRITE_SIZE)",
            traffic_source=counters_src, mfma_busy_per_launch=busy,
            executed_flops_per_step=executed, algorithmic_f32_flops_per_step=algorithmic,
            algorithmic_model="tuple MLP of train_shot.py:48-73 at %d tuples: 2 M K N per Linear; executed = %d %s MFMA products per "
                              "float32 product (%s)" % (B * T, int(nprod), "bf16" if nprod == 6.0 else "fp16",
                                                        "3-way exact operand split" if nprod == 6.0 else "fp16 operand pairs, 22-23 bits"),
            hbm=hbm))
        for k_ in ("algorithmic_bytes_per_launch", "launch_ms_kind"):
            roofline.pop(k_, None)
    roofline["power"] = power = run.get("power")
    if power and power.get("mlp_launches"):
        # the claim DESIGN.md section 5 makes about the dominant kernel, decided by this run's telemetry: each launch form of the
        # tuple MLP, run back to back on its own, against the socket's power cap
        fr = [l_.get("power_frac_of_cap", {}).get("mean") for l_ in power["mlp_launches"]]
        ppt = [l_.get("throttle", {}).get("per_ppt_pwr") for l_ in power["mlp_launches"]]
        if all(f_ is not None for f_ in fr):
            power["mlp_power_limit"] = {
                "claim": "the tuple MLP's launches run at the chip's power limit (so roofline.frac is bounded by energy per flop of the "
                         "split arithmetic, not by stalls)",
                "criterion": "every launch form alone draws >= 95 % of the power cap (mean over ~1.5 s back to back)",
                "power_frac_of_cap_per_launch": fr, "ppt_throttle_residency_percent_per_launch": ppt,
                "confirmed": bool(min(fr) >= 0.95)}
        g_ = power.get("library_bf16_gemm")
        if g_ and g_.get("tflops") and roofline.get("achieved_executed"):
            # the matrix pipe's practical ceiling on this chip under this cap, measured in this run, beside the MLP's issued rate
            roofline["frac_executed_vs_library_bf16_gemm"] = round(roofline["achieved_executed"] / g_["tflops"], 4)
            roofline["frac_vs_library_bf16_gemm"] = round(roofline["achieved"] / g_["tflops"], 4)
    ok = pose_ok_vs_gt(res, step.scenes)
    cpu, agree = run.get("cpu") or (None, None)
    total_scenes = B * world * args.steps
    two, f16_agreement = run["two"], run["f16_agreement"]
    dt_other, dt_native, dt_f16 = run["dt_other"], run["dt_native"], run["dt_f16"]
    step_times = run["step_times"]
    sha = __import__("hashlib").sha256(all_rec.tobytes()).hexdigest()
    line = {
        "metric": "scenes/sec (1/2/4/8 GPU) at 4096 pts x 20k tuples; 5deg5cm match vs ref",
        "value": total_scenes / dt, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: SHOT model, %d scenes/GPU x %d pts x %d tuples x %d rots, "
                               "720 sphere bins, res 2 mm, bottle axes; %s clouds; random-init weights + teacher prior; scale head on %s; "
                               "%s; MLP arithmetic: %s; tuple rows %s"
                               % (B, N, T, R, step.cloud, "all tuples" if args.eager_scale_head else "the kept pairs only",
                                  "every step on one HIP stream" if args.single_stream else
                                  "consecutive steps (different scene batches) alternate between two HIP streams (double-buffered state)",
                                  "float32 operands split exactly into 3 x bf16, 6 exact products on the bf16 matrix cores, "
                                  "float32 accumulate (error vs float64 within 3 x a library float32 GEMM's: mlp_error_vs_f64, "
                                  "tests/test_mlp_split.py)"
                                  if _models.MLP_ARITH == "split" else
                                  ("float32 operands as fp16 pairs (22-23 significant bits), 3 products on the fp16 matrix "
                                   "cores, float32 accumulate (error vs float64 at the library float32 GEMMs' level; NOT exact "
                                   "products)" if _models.MLP_ARITH == "split16" else "f32-input matrix cores"),
                                  (("built and gathered inside the first ResLayer's kernel (pair features and descriptors: no per-tuple "
                                    "array between sampler and MLP)" if ENCODE_FOLDED else
                                    "gathered inside the first ResLayer's kernel (never written)") if GATHERED_TUPLES
                                   else "materialised ([T, 360] float32)")
                                  + ("; bins drawn in the epilogue of the logit head's output layer (logits never written)"
                                     if FUSED_DRAW else "")),
                   "scenes_per_gpu": B, "points": N, "tuples": T, "rots": R, "cloud": step.cloud,
                   "parallelism": "scene-sharded x%d" % world},
        # the same run with the scale head on every tuple (the reference's forward order), same loop protocol
        "value_reference_order" if not args.eager_scale_head else "value_kept_pairs_order":
            (total_scenes / dt_other) if dt_other else None,
        # the same run with the MLP on the f32-input matrix instruction (no operand splitting)
        "value_f32_input_mfma": (total_scenes / dt_native) if dt_native else None,
        # the same run with the MLP in f16x2 arithmetic (operands as fp16 pairs, 22-23 bits; not the headline: products are
        # not exact there -- its error against float64 is under mlp_error_vs_f64.split_f16x2)
        "value_f16x2_mfma": (total_scenes / dt_f16) if dt_f16 else None,
        "f16x2_agreement": f16_agreement,
        # the same path on clouds at the point density real inputs have (eval.py:185-201's 2 mm voxel grid: ~250 neighbours inside
        # the SHOT support instead of the synthetic clouds' ~90), single stream, same loop protocol; with its per-stage times
        # (round 6: `value` of the voxel-density object is the batch mode's -- two streams like the headline --, the single-stream
        # figure of rounds 4-5 is value_voxel_density_single_stream)
        "value_voxel_density": (run.get("voxel") or {}).get("value"),
        "value_voxel_density_single_stream": (run.get("voxel") or {}).get("value_single_stream"),
        "voxel_density": run.get("voxel"),
        # the headline loop WITHOUT the teacher prior (the reference has none, eval.py:225-235: what trained checkpoints run; with
        # the bench's random-init weights the votes then do not cluster, so only its speed means anything) and with the prior read
        # as a [T, 6, 32] array (the form of rounds 1-4: + 983 MB of reads per step)
        "value_no_prior": (total_scenes / run["dt_noprior"]) if run.get("dt_noprior") else None,
        "value_array_prior": (total_scenes / run["dt_arrayprior"]) if run.get("dt_arrayprior") else None,
        "prior_form_timed": ("array [T, 6, 32]" if getattr(args, "array_prior", False) else
                             "generator (ops.BinPrior: 24 bytes per tuple, evaluated in the bin draw's epilogue)"),
        # the headline's stream mode; value_single_stream = the same steps on ONE stream (rounds 1-3), same loop protocol
        "two_streams": two,
        "value_single_stream": (total_scenes / dt_single) if dt_single else None,
        "ms_per_step_single_stream": (1e3 * dt_single / args.steps) if dt_single else None,
        # intervals between consecutive step completions inside the timed loop (one HIP event per step), and the same for the
        # single-stream loop: the per-step spread of the headline
        "step_interval_ms": run["intervals"],
        "step_interval_ms_single_stream": run["intervals_single"],
        # first-event to last-event time of the sampled steps on their own stream: a step's RESIDENCE on its stream (with two
        # streams a step overlaps its neighbours, so this is about two step intervals -- not a per-step time)
        "step_residence_ms_sampled": {"min": round(step_times[0], 4), "median": round(step_times[len(step_times) // 2], 4),
                                      "max": round(step_times[-1], 4), "n": len(step_times)},
        "records_gathered": int(all_rec.shape[0]),
        # SHA-256 of the gathered records in global scene order: equal for every world size and stream mode (the records
        # depend on the global scene id only)
        "records_sha256": sha,
        "host_cores_of_rank0": run["affinity"],
        # the path's one collective: all_gather of the 160-byte scene records, HIP-event time of the stage
        "collective": collective_info(backend, world, B, all_rec.shape[0], stage_ms.get("gather", 0.0), run.get("rank_ms"), sha),
        "roofline": roofline, "cpu_baseline": cpu,
        "pose_5deg5cm_vs_gt": ok / B, "oracle_agreement": agree,
    }
    line.update(run.get("evidence") or {})
    # self-checks of the run: a comparison loop whose records differ from the headline's is a failed run, not a footnote
    problems = []
    if two is not None and not two["records_identical_to_single_stream"]:
        problems.append("two_streams: a pipeline's records differ from the single-stream records of its own scenes")
    if f16_agreement is not None and f16_agreement["scenes_with_equal_argmax_rotation_bins_kept_count"] != f16_agreement["scenes"]:
        problems.append("f16x2_agreement: a scene's arg-max / rotation bins / kept count differs from the headline arithmetic's")
    for name_, ag_ in (("oracle_agreement", agree), ("voxel_density.oracle_agreement", (run.get("voxel") or {}).get("oracle_agreement"))):
        if ag_ and ag_.get("defects"):
            problems.append("%s: scenes %s differ from the oracle even when it votes from the GPU's own bin draws" % (name_, ag_["defects"]))
        if ag_ and ag_.get("weights_identical_to_the_gpu_run") is False:
            problems.append("%s: the oracle workers' weights differ from the GPU run's" % name_)
    bad = [k_ for k_, e_ in per_kernel.items() if e_.get("frac") is not None and not (0.0 < e_["frac"] <= 1.0)]
    if bad:
        problems.append("roofline.per_kernel: fraction outside (0, 1] for " + ", ".join(bad))
    line["ok"] = not problems
    line["problems"] = problems
    if args.breakdown:
        import sys
        print("%-22s %10s %12s %10s" % ("stage", "ms/launch", "alg MB", "GB/s"), file=sys.stderr)
        for r in rows:
            print("%-22s %10.3f %12.1f %10.1f" % r, file=sys.stderr)
    return line, problems
