"""Writes profiles/<round>_summary.md from the committed rocprofv3 kernel stats and the bench lines of the round (round 4 on: the
bench line carries its own counters, so no separate PMC file is read).  usage: python scratch/profile_summary.py [r4]"""
import csv, json, os, sys
RND = sys.argv[1] if len(sys.argv) > 1 else "r4"
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
           "--single-stream --no-reference-order --no-native-arith --no-evidence --no-f16x2 --no-counters` (%d passes incl. priming); "
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
out.append("| roofline (tuple MLP, %s) | %.0f TFLOP/s executed of %.0f = %.3f; launch_ms %.3f; HBM traffic %.2f GB per step |"
           % (r["frac_kind"], r["achieved"], r["peak"], r["frac"], r["launch_ms"], (r["traffic"] or 0) / 1e9))
out.append("| algorithmic float32 rate | %.0f TFLOP/s (f32-input MFMA peak %.0f) |" % (r["algorithmic_f32_tflops"], r["f32_input_mfma_peak_tflops"]))
cb = b.get("cpu_baseline")
if cb:
    out.append("| cpu_baseline | %.3f scenes/s on %d cores (%s) |" % (cb["value"], cb["cores"], cb["kind"]))
out.append("\n## Kernels of a single-stream step (rocprofv3 averages)\n")
out.append("| kernel | launches / step | avg us | us / step |\n|---|---|---|---|")
tot = 0.0
for key, per, us in rows:
    out.append("| `%s` | %.0f | %.1f | %.1f |" % (key, per, us, per * us))
    tot += per * us
out.append("| **sum** | %.0f | | **%.1f** |" % (sum(x[1] for x in rows), tot))
out.append("\nPyTorch / BLAS kernels launched once (or k times) per step: **%d** %s\n" % (len(foreign), foreign if foreign else ""))
out.append("## Stages: bound and fraction of that bound (from the bench line, counters of the same run)\n")
out.append("| stage | kernel | ms | bound | frac | alg MB | counters' MB | VALU / LDS / conflicts busy | work |\n|---|---|---|---|---|---|---|---|---|")
for s_, e in r["per_kernel"].items():
    a = e.get("activity") or {}
    w = e.get("work") or {}
    wtxt = ", ".join("%s %.3g" % (k_, v_) for k_, v_ in w.items() if isinstance(v_, (int, float)) and k_.endswith("_per_s"))
    out.append("| %s | `%s` | %.4f | %s | %s | %s | %s | %s / %s / %s | %s |" % (
        s_, e.get("kernel"), e["ms"], e.get("bound"), e.get("frac"), e.get("alg_MB"), e.get("pmc_MB"),
        a.get("valu_busy"), a.get("lds_busy"), a.get("lds_bank_conflict"), wtxt))
out.append("\nMatrix-pipe activity of the three tuple-MLP launches: " + "; ".join(
    "`%s` MFMA-busy %s at %s GHz" % (k_, v_.get("mfma_busy"), v_.get("shader_clock_ghz")) for k_, v_ in r["mfma_busy_per_launch"].items()) + "\n")
for tag, title in (("ensemble", "Ensemble workload (configs[2])"), ("dense64k", "Dense-pair workload (configs[4])")):
    try:
        e = load("%s_%s_bench_n1.json" % (RND, tag))
    except OSError:
        continue
    out.append("## %s\n" % title)
    out.append("`%s`\n" % e["config"]["workload"])
    out.append("value %.0f scenes/s (%.3f ms per step)%s; roofline frac %.3f; pose_5deg5cm_vs_gt %.3f\n" % (
        e["value"], e["ms_per_step"],
        (", single stream %.0f" % e["value_single_stream"]) if e.get("value_single_stream") else "", e["roofline"]["frac"], e["pose_5deg5cm_vs_gt"]))
    if "per_model_ms" in e:
        out.append("per model: %s\n" % json.dumps(e["per_model_ms"]))
    out.append("per stage (ms): %s\n" % json.dumps(e["roofline"]["per_stage_ms"]))
    if tag == "ensemble" and os.path.exists(os.path.join(P, RND + "_ensemble_kernel_stats.csv")):
        st, rws, fr = kernel_rows(RND + "_ensemble_kernel_stats.csv", "ensemble_select_kernel")
        out.append("kernels per step: %.0f launches, %.1f us; PyTorch / BLAS kernels per step: %d (`%s_ensemble_step_trace.txt`)\n"
                   % (sum(x[1] for x in rws), sum(x[1] * x[2] for x in rws), len(fr), RND))
open(os.path.join(P, RND + "_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
