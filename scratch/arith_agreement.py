"""Do the three MLP arithmetics draw the same bins?  Seeds x 64 bench scenes: bins (7.68 M draws per step), vote-grid arg-max,
rotation bins and kept-pair counts of split16 (f16x2) and native (library float32 GEMMs) against split (bf16x3)."""
import sys, os, types, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from cppf2_amd import models
seeds = [int(s) for s in (sys.argv[1:] or range(8))]
dev = torch.device("cuda")
tot = {m: dict(bins_diff=0, draws=0, argmax_diff=0, up_diff=0, right_diff=0, kept_diff=0, scenes=0, max_t_diff_m=0.0) for m in ("split16", "native")}
for seed in seeds:
    args = types.SimpleNamespace(scenes_per_gpu=64, points=4096, tuples=20000, rots=180, seed=seed, vote_mode=0, eager_scale_head=False)
    st = bench.Step(args, 0, 1, dev)
    out = {}
    for mode in ("split", "split16", "native"):
        models.MLP_ARITH = mode
        st.run(); torch.cuda.synchronize()
        out[mode] = (st.pipe.results_to_numpy().copy(), st.pipe.bins.clone())
    models.MLP_ARITH = "split"
    a = out["split"][0]
    for m in ("split16", "native"):
        r, t = out[m][0], tot[m]
        t["bins_diff"] += int((out[m][1] != out["split"][1]).sum()); t["draws"] += out[m][1].numel()
        t["argmax_diff"] += int((r["argmax"] != a["argmax"]).sum()); t["up_diff"] += int((r["up_idx"] != a["up_idx"]).sum())
        t["right_diff"] += int((r["right_idx"] != a["right_idx"]).sum()) if "right_idx" in r.dtype.names else 0
        t["kept_diff"] += int((r["kept"] != a["kept"]).sum()); t["scenes"] += 64
        t["max_t_diff_m"] = max(t["max_t_diff_m"], float(np.nanmax(np.abs(r["t"] - a["t"]))))
    print("seed", seed, {m: (tot[m]["bins_diff"], tot[m]["argmax_diff"]) for m in tot}, flush=True)
print(json.dumps(dict(seeds=seeds, against="split (bf16x3)", **tot)))
