"""Writes profiles/<round>_summary.md from the committed rocprofv3 kernel stats, PMC traffic passes and bench line.
usage: python scratch/profile_table.py [r2]"""
import csv, json, os, sys
RND = sys.argv[1] if len(sys.argv) > 1 else "r3"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles")
stats = list(csv.DictReader(open(os.path.join(P, RND + "_kernel_stats.csv"))))
pmc = json.load(open(os.path.join(P, RND + "_pmc_traffic.json")))
bench = json.load(open(os.path.join(P, RND + "_bench_n1.json")))
# passes of the step under rocprofv3 = launches of a kernel that runs exactly once per pass (priming + warm-up + timed)
steps = next(int(r["Calls"]) for r in stats if r["Name"].startswith("vote_worklist_kernel") or "vote_worklist_kernel" in r["Name"])
rows, gemm_ns, elt_ns, setup = [], 0.0, 0.0, []
for r in stats:
    n = r["Name"]
    calls = int(r["Calls"])
    torchy = n.startswith("Cijk") or "rocblas" in n or "at::native" in n or "elementwise" in n or "rocclr" in n
    if torchy and (calls % steps != 0 or calls < steps):
        # not launched once (or k times) per pass: one-time work of the priming pass (weight packing: bf16 casts, cat; the
        # teacher prior; uploads), not part of a steady-state step
        setup.append((n.split("(")[0].replace("void ", "")[:90], calls, float(r["TotalDurationNs"]) / 1e6))
        continue
    if n.startswith("Cijk") or "rocblas" in n:
        gemm_ns += float(r["TotalDurationNs"]); continue
    if torchy:
        elt_ns += float(r["TotalDurationNs"]); continue
    key = n.split("(")[0].replace("void ", "")
    t = pmc.get(key, {})
    hbm = (2 * t.get("FETCH_SIZE_KB_per_launch", 0) + t.get("WRITE_SIZE_KB_per_launch", 0)) * 1024 if t else None
    if not t and key + "#large" in pmc:
        # the PMC passes split this kernel's dispatches by duration (the gathering MLP kernel: tuple encoder + scale head):
        # launch-weighted mean over both classes, like the stats row's average
        parts = [pmc[key + sfx] for sfx in ("#large", "#small") if key + sfx in pmc]
        nl = sum(p_["launches"] for p_ in parts)
        hbm = sum((2 * p_["FETCH_SIZE_KB_per_launch"] + p_["WRITE_SIZE_KB_per_launch"]) * 1024 * p_["launches"] for p_ in parts) / nl
    rows.append((key, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, hbm))
rows.sort(key=lambda x: -x[1] * x[2])
with open(os.path.join(P, RND + "_summary.md"), "w") as f:
    f.write("# Round profile summary %s (one MI355X, `python bench.py --steps 5 --warmup 2 --cpu-scenes 0 --no-reference-order --no-native-arith --no-evidence --no-f16x2 --no-two-streams`, %d passes incl. priming)\n\n" % (RND, steps))
    f.write("Sources: `%s_kernel_stats.csv` = `rocprofv3 --kernel-trace --stats` of that command; `%s_pmc_traffic.json` = "
            "`rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes (KB per launch; HBM bytes = "
            "2 x FETCH + WRITE per MI355X_MICROARCH.md's gfx950 note); `%s_bench_n1.json` = the bench line of the same build. "
            "Regenerate with `scratch/refresh_profiles.sh` + `scratch/profile_table.py`.\n\n" % (RND, RND, RND))
    rf = bench["roofline"]
    f.write("Bench: %.0f scenes/s, %.2f ms/step (64 scenes); dominant kernel `%s` (%s-bound) %.3f ms per step, "
            "roofline %.0f %s of %d (frac %.3f), PMC traffic %.0f MB.\n"
            % (bench["value"], bench["ms_per_step"], rf["kernel_name"], rf["bound"], rf["launch_ms"],
               rf["achieved"], rf["unit"], rf["peak"], rf["frac"], (rf["traffic"] or 0) / 1e6))
    if "hbm" in rf:
        h = rf["hbm"]
        f.write("HBM-bound kernel with the longest launch: `%s` %.3f ms, %.0f GB/s of %d algorithmic (frac %.3f), PMC traffic %.0f MB per launch; "
                "same run with the MLP on the f32-input matrix cores: %s scenes/s.\n"
                % (h["kernel_name"], h["launch_ms"], h["achieved"], h["peak"], h["frac"], (h["traffic"] or 0) / 1e6,
                   "%.0f" % bench["value_f32_input_mfma"] if bench.get("value_f32_input_mfma") else "n/a"))
    f.write("\n")
    f.write("| HIP kernel | launches / step | avg us / launch | ms / step | HBM MB / launch (PMC) | HBM GB/s (PMC) |\n|---|---|---|---|---|---|\n")
    tot = 0.0
    for k, c, us, hbm in rows:
        tot += c * us / 1e3
        f.write("| `%s` | %.0f | %.1f | %.3f | %s | %s |\n" % (k, c, us, c * us / 1e3, "%.1f" % (hbm / 1e6) if hbm else "-",
                                                            "%.0f" % (hbm / 1e9 / (us / 1e6)) if hbm else "-"))
    f.write("| **all HIP kernels** | | | **%.3f** | | |\n" % tot)
    f.write("| hipBLASLt / rocBLAS GEMMs launched every pass (PyTorch) | | | %.3f | | |\n" % (gemm_ns / steps / 1e6))
    f.write("| PyTorch elementwise / copy kernels launched every pass | | | %.3f | | |\n" % (elt_ns / steps / 1e6))
    f.write("\nSetup only (PyTorch / runtime kernels whose call count is not a multiple of the %d passes: weight packing of the "
            "priming pass, the teacher prior, uploads; not part of a step): %d kernels, %.3f ms in total over the whole run.\n"
            % (steps, len(setup), sum(x[2] for x in setup)))
    for nme, calls, ms in sorted(setup, key=lambda x: -x[2])[:8]:
        f.write("  * `%s` x %d: %.3f ms\n" % (nme, calls, ms))
    f.write("\nPer-stage HIP-event times of the bench (ms per step): `%s`\n" % json.dumps(bench["roofline"]["per_stage_ms"]))
print(open(os.path.join(P, RND + "_summary.md")).read())
