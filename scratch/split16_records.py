import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from cppf2_amd import models
args = types.SimpleNamespace(scenes_per_gpu=64, points=4096, tuples=20000, rots=180, seed=0, vote_mode=0, eager_scale_head=False)
dev = torch.device("cuda")
st = bench.Step(args, 0, 1, dev)
out = {}
for mode in ("split", "split16", "native"):
    models.MLP_ARITH = mode
    st.run(); torch.cuda.synchronize()
    out[mode] = (st.pipe.results_to_numpy().copy(), st.pipe.bins.clone())
models.MLP_ARITH = "split"
a, b, c = out["split"][0], out["split16"][0], out["native"][0]
for name, r in (("split16", b), ("native", c)):
    ok = 0
    for s in range(64):
        sc = st.scenes[s]
        terr = np.linalg.norm(r["t"][s] - sc["t"]); cosang = abs(float(r["R"][s][:, 1] @ sc["R"][:, 1]))
        ok += int(terr < 0.05 and np.degrees(np.arccos(min(cosang, 1.0))) < 5.0)
    print(name, "pose ok", ok, "/64; argmax equal to split:", int((r["argmax"] == a["argmax"]).sum()), "up_idx equal:", int((r["up_idx"] == a["up_idx"]).sum()),
          "kept equal:", int((r["kept"] == a["kept"]).sum()), "flags", np.unique(r["flags"]), "nan t:", int(np.isnan(r["t"]).any(1).sum()))
print("bins differing split16 vs split:", int((out["split16"][1] != out["split"][1]).sum()), "of", out["split"][1].numel(), "| native vs split:", int((out["native"][1] != out["split"][1]).sum()))
bad = [s for s in range(64) if not np.array_equal(b["argmax"][s], a["argmax"][s])]
print("scenes with different argmax:", bad[:10])
for s in bad[:3]:
    print(s, "split t", a["t"][s], "split16 t", b["t"][s], "gt", st.scenes[s]["t"], "peak", a["peak"][s], b["peak"][s])
