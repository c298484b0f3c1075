"""Writes profiles/<round>_summary.md from the committed rocprofv3 kernel stats and the bench lines of the round (round 4 on: the
bench line carries its own counters, so no separate PMC file is read).  usage: python scratch/profile_summary.py [r4]"""
import csv, json, os, sys
RND = sys.argv[1] if len(sys.argv) > 1 else "r5"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles")


def load(name):
    with open(os.path.join(P, name)) as f:
        return json.loads(f.readline())


def kernel_rows(stats_csv, per_step_kernel):
    stats = list(csv.DictReader(open(os.path.join(P, stats_csv))))
    steps = next(int(r["Calls"]) for r in stats if per_step_kernel in r["Name"])
    rows, foreign = [], []
    for r in stats:
        n = r["Name"]
        key = n.split("(")[0].replace("void ", "")
        calls = int(r["Calls"])
        torchy = n.startswith("Cijk") or "rocblas" in n or "at::native" in n or "elementwise" in n
        if torchy or "rocclr" in n:
            if calls % steps == 0 and calls >= steps and torchy:
                foreign.append((key[:80], calls / steps, float(r["AverageNs"]) / 1e3))
            continue
        if calls < steps:
            continue
        rows.append((key, calls / steps, float(r["AverageNs"]) / 1e3))
    rows.sort(key=lambda x: -x[1] * x[2])
    return steps, rows, foreign


b = load(RND + "_bench_n1.json")
steps, rows, foreign = kernel_rows(RND + "_kernel_stats.csv", "vote_worklist_kernel")
out = []
out.append("# Round profile summary %s (one MI355X)\n" % RND)
out.append("Sources: `%s_kernel_stats.csv` = `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --cpu-scenes 0 "
           "--single-stream --no-reference-order --no-native-arith --no-evidence --no-f16x2 --no-counters --no-voxel-density` (%d passes incl. priming); "
           "`%s_step_trace.txt` = launch order of one step of that run; `%s_two_stream_trace.txt` = two consecutive steps of the "
           "default (two-stream) loop; `%s_bench_n1.json` = the default `python bench.py` line, whose counters (HBM bytes, unit "
           "activity) come from rocprofv3 passes the bench ran itself.\n" % (RND, steps, RND, RND, RND))
out.append("## Headline\n")
out.append("| | |\n|---|---|")
out.append("| value (two streams) | %.0f scenes/s, %.3f ms per step |" % (b["value"], b["ms_per_step"]))
out.append("| value_single_stream | %.0f scenes/s, %.3f ms per step |" % (b["value_single_stream"], b["ms_per_step_single_stream"]))
out.append("| records identical across stream modes | %s |" % b["two_streams"]["records_identical_to_single_stream"])
for k in ("value_reference_order", "value_f32_input_mfma", "value_f16x2_mfma"):
    if b.get(k):
        out.append("| %s | %.0f scenes/s |" % (k, b[k]))
r = b["roofline"]
out.append("| value_voxel_density (clouds at 2 mm voxel spacing, %d stream%s) | %.0f scenes/s, %.3f ms per step; descriptor stage %.3f ms%s |"
           % (b["voxel_density"].get("streams", 1), "s" if b["voxel_density"].get("streams", 1) > 1 else "", b["value_voxel_density"],
              b["voxel_density"]["ms_per_step"], b["voxel_density"]["shot_stage_ms"],
              ("; single stream %.0f" % b["value_voxel_density_single_stream"]) if b.get("value_voxel_density_single_stream") else ""))
if b.get("value_no_prior"):
    out.append("| value_no_prior / value_array_prior (the reference has no prior; the prior as a [T, 6, 32] array) | %.0f / %.0f scenes/s |"
               % (b["value_no_prior"], b["value_array_prior"]))
out.append("| roofline.frac (tuple MLP: algorithmic float32 flops / time / bf16 MFMA peak) | %.0f TFLOP/s of %.0f = **%.4f**; launch_ms %.3f "
           "(rocprofv3: the three launches' durations in `%s_step_trace.txt`); HBM traffic %.2f GB per step |"
           % (r["achieved"], r["peak"], r["frac"], r["launch_ms"], RND, (r["traffic"] or 0) / 1e9))
out.append("| roofline.frac_executed (bf16 MFMA work issued, 6 products per float32 product) | %.0f TFLOP/s = %.3f |" % (r["achieved_executed"], r["frac_executed"]))
out.append("| roofline.frac_vs_f32_mfma_peak (same algorithmic rate / %.0f TFLOP/s of the f32-input matrix instruction) | %.3f |"
           % (r["f32_input_mfma_peak_tflops"], r["frac_vs_f32_mfma_peak"]))
out.append("| roofline.hbm (longest streaming kernel: %s) | %.0f GB/s algorithmic on the kernel's rocprofv3 duration = %.3f of 8 TB/s |"
           % (r["hbm"]["kernel"], r["hbm"]["achieved"], r["hbm"]["frac"]))
cb = b.get("cpu_baseline")
if cb:
    out.append("| cpu_baseline | %.3f scenes/s on %d cores (%s)%s |" % (cb["value"], cb["cores"], cb["kind"],
               ("; all-cores pool %.2f scenes/s over %d workers" % (cb["all_cores_pool"]["value"], cb["all_cores_pool"]["workers"]))
               if cb.get("all_cores_pool") else ""))
ag = b.get("oracle_agreement")
if ag:
    out.append("| oracle_agreement (every scene of the timed batch recomputed by the CPU oracle) | %d scenes: arg-max equal %d, up / right bins %d / %d, "
               "kept count %d, translation bits %d; 5deg5cm match %.2f; %d of %d bin draws differ; defects %s |"
               % (ag["scenes"], ag["centre_argmax_equal"], ag["up_bin_equal"], ag["right_bin_equal"], ag["kept_count_equal"],
                  ag["translation_bit_equal"], ag["match_5deg5cm"], ag["bin_draws_differing"], ag["bin_draws"], ag["defects"]))
pw = (b.get("roofline") or {}).get("power") or {}
if pw.get("available"):
    w = pw["windows"]
    out.append("| roofline.power: socket power / cap, shader clock, PPT throttle residency | cap %.0f W; idle %.0f W; two-stream loop %.0f W (%.3f), %.0f MHz, %s %%; "
               "single-stream loop %.0f W (%.3f) |"
               % (pw["power_cap_w"], w["idle"]["power_w"]["mean"], w["two_stream_loop"]["power_w"]["mean"],
                  w["two_stream_loop"]["power_frac_of_cap"]["mean"], w["two_stream_loop"]["sclk_mhz"]["mean"],
                  w["two_stream_loop"].get("throttle", {}).get("per_ppt_pwr"), w["single_stream_loop"]["power_w"]["mean"],
                  w["single_stream_loop"]["power_frac_of_cap"]["mean"]))
    for l_ in pw.get("mlp_launches", []):
        out.append("| roofline.power.mlp_launches[%d] (`%s`, alone, back to back) | %.3f ms; %.0f W = %.3f of the cap; %.0f MHz; PPT residency %s %% |"
                   % (l_["launch"], l_["op"], l_["ms"], l_["power_w"]["mean"], l_["power_frac_of_cap"]["mean"], l_["sclk_mhz"]["mean"],
                      l_.get("throttle", {}).get("per_ppt_pwr")))
    if pw.get("library_bf16_gemm"):
        g_ = pw["library_bf16_gemm"]
        out.append("| library bf16 GEMM under the same telemetry (hipBLASLt, 8192^3) | %.0f TFLOP/s; %.0f W = %.3f of the cap; %.0f MHz; the MLP issues %.3f of it |"
                   % (g_["tflops"], g_["power_w"]["mean"], g_["power_frac_of_cap"]["mean"], g_["sclk_mhz"]["mean"],
                      b["roofline"].get("frac_executed_vs_library_bf16_gemm") or float("nan")))
out.append("\n## Kernels of a single-stream step (rocprofv3 averages)\n")
out.append("| kernel | launches / step | avg us | us / step |\n|---|---|---|---|")
tot = 0.0
for key, per, us in rows:
    out.append("| `%s` | %.0f | %.1f | %.1f |" % (key, per, us, per * us))
    tot += per * us
out.append("| **sum** | %.0f | | **%.1f** |" % (sum(x[1] for x in rows), tot))
out.append("\nPyTorch / BLAS kernels launched once (or k times) per step: **%d** %s\n" % (len(foreign), foreign if foreign else ""))
out.append("## Stages: bound and fraction of that bound (from the bench line, counters of the same run)\n")
out.append("(`frac`: streaming kernels = algorithmic bytes / rocprofv3 kernel duration / 8 TB/s; descriptor and voting kernels = the busier of "
           "VALU instruction issue (instructions x 2 cycles / SIMD-32 capacity) and LDS-array cycles, from the SQ pass: never above 1.)\n")
out.append("| stage | kernel | event ms | kernel ms | bound | frac | alg MB | counters' MB | VALU issue / VALU busy / LDS / conflicts | work |\n|---|---|---|---|---|---|---|---|---|---|")
for s_, e in r["per_kernel"].items():
    a = e.get("activity") or {}
    w = e.get("work") or {}
    wtxt = ", ".join("%s %.3g" % (k_, v_) for k_, v_ in w.items() if isinstance(v_, (int, float)) and k_.endswith("_per_s"))
    out.append("| %s | `%s` | %.4f | %s | %s | %s | %s | %s | %s / %s / %s / %s | %s |" % (
        s_, e.get("kernel"), e["ms"], e.get("kernel_ms", ""), e.get("bound"), e.get("frac"), e.get("alg_MB"), e.get("pmc_MB"),
        a.get("valu_issue"), a.get("valu_busy"), a.get("lds_busy"), a.get("lds_bank_conflict"), wtxt))
out.append("\nMatrix-pipe activity of the three tuple-MLP launches: " + "; ".join(
    "`%s` MFMA-busy %s at %s GHz" % (k_, v_.get("mfma_busy"), v_.get("shader_clock_ghz")) for k_, v_ in r["mfma_busy_per_launch"].items()) + "\n")
for tag, title in (("ensemble", "Ensemble workload (configs[2])"), ("dense64k", "Dense-pair workload (configs[4])")):
    try:
        e = load("%s_%s_bench_n1.json" % (RND, tag))
    except OSError:
        continue
    out.append("## %s\n" % title)
    out.append("`%s`\n" % e["config"]["workload"])
    out.append("value %.0f %s (%.3f ms per step)%s; roofline frac %.4f (executed %.3f); pose_5deg5cm_vs_gt %.3f\n" % (
        e["value"], e["unit"], e["ms_per_step"],
        (", single stream %.0f" % e["value_single_stream"]) if e.get("value_single_stream") else "", e["roofline"]["frac"],
        e["roofline"]["frac_executed"], e["pose_5deg5cm_vs_gt"]))
    if "per_model_ms" in e:
        out.append("per model: %s\n" % json.dumps(e["per_model_ms"]))
    out.append("per stage (ms): %s\n" % json.dumps(e["roofline"]["per_stage_ms"]))
    if tag == "ensemble" and os.path.exists(os.path.join(P, RND + "_ensemble_kernel_stats.csv")):
        st, rws, fr = kernel_rows(RND + "_ensemble_kernel_stats.csv", "ensemble_select_kernel")
        out.append("kernels per step: %.0f launches, %.1f us; PyTorch / BLAS kernels per step: %d (`%s_ensemble_step_trace.txt`)\n"
                   % (sum(x[1] for x in rws), sum(x[1] * x[2] for x in rws), len(fr), RND))
# the same step on clouds at voxel-grid density
vs = os.path.join(P, RND + "_voxel2mm_kernel_stats.csv")
if os.path.exists(vs):
    st, rws, fr = kernel_rows(RND + "_voxel2mm_kernel_stats.csv", "vote_worklist_kernel")
    out.append("## The step on clouds at voxel-grid density (`bench.py --cloud voxel2mm`, ~250 neighbours inside the 2 cm support)\n")
    out.append("`%s_voxel2mm_kernel_stats.csv` / `%s_voxel2mm_step_trace.txt`: %.0f launches, %.1f us of kernel time per step; the kernels that move:\n"
               % (RND, RND, sum(x[1] for x in rws), sum(x[1] * x[2] for x in rws)))
    base = {k_: us for k_, per, us in rows}
    out.append("| kernel | avg us (voxel2mm) | avg us (synthetic) |\n|---|---|---|")
    for k_, per, us in rws:
        if k_ in base and abs(us - base[k_]) > 0.15 * base[k_] and us > 20:
            out.append("| `%s` | %.1f | %.1f |" % (k_, us, base[k_]))
    out.append("")
extra = os.path.join(P, RND + "_extra.md")
if os.path.exists(extra):
    out.append(open(extra).read())
open(os.path.join(P, RND + "_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
