"""Which rows of the HIP SHOT descriptor differ from the oracle's by more than 2e-5, and why (LRF eigen-gap, sign tally)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import shot_oracle as S
from cppf2_amd import shot, synth

def report(name, pc, rn, rs):
    hs, hn = shot.compute(pc, rn, rs)
    hs, hn = hs.reshape(-1, 352), hn.reshape(-1, 3)
    os_, on, orf, diag = S.compute_ex(pc, rn, rs)
    ok = ~np.isnan(os_).any(1)
    err = np.abs(hs - os_).max(1)
    bad = np.where(ok & (err >= 2e-5))[0]
    gap12 = (diag[:, 0] - diag[:, 1]) / diag[:, 0]
    gap23 = (diag[:, 1] - diag[:, 2]) / diag[:, 0]
    print(name, "rows", ok.sum(), "bad", len(bad), "nan-equal", np.array_equal(np.isnan(os_), np.isnan(hs)))
    nerr = np.abs(hn - on).max(1)
    print("  normals: max err %.3g, rows > 5e-6: %d" % (np.nanmax(nerr), (nerr > 5e-6).sum()))
    d2 = ((pc[:, None, :] - pc[None, bad[:200], :]) ** 2).sum(-1) if len(bad) else np.zeros((0, 0))
    for k, i in enumerate(bad[:25]):
        cnt = int((d2[:, k] < rs * rs).sum()) if k < 200 else -1
        print("   row %5d err %.3g gap12 %.3g gap23 %.3g tally x %d z %d nbrs %d normal err %.2g" % (i, err[i], gap12[i], gap23[i], diag[i, 3], diag[i, 4], cnt, nerr[i]))
    if len(bad):
        expl = (np.abs(diag[bad, 3]) <= 1) | (np.abs(diag[bad, 4]) <= 1) | (gap12[bad] < 1e-6) | (gap23[bad] < 1e-6)
        print("  explained by |tally|<=1 or gap<1e-6: %d of %d" % (expl.sum(), len(bad)))
        print("  err of bad rows: median %.3g max %.3g; frac of all rows with |tally|<=1: %.4f" % (np.median(err[bad]), err[bad].max(), ((np.abs(diag[ok, 3]) <= 1) | (np.abs(diag[ok, 4]) <= 1)).mean()))

for n, rn, rs in [(2600, 0.02, 0.02), (6000, 0.02, 0.02), (1500, 0.03, 0.012)]:
    rng = np.random.RandomState(n)
    v = rng.randn(n, 3)
    pc = (v / np.linalg.norm(v, axis=1, keepdims=True) * (rng.rand(n, 1) ** (1 / 3)) * 0.03 + 0.5).astype(np.float32)
    report("ball n=%d rn=%g rs=%g" % (n, rn, rs), pc, rn, rs)
sc = synth.make_scene(0, 0, 4096)
report("bench scene 4096", sc["pc"], 0.02, 0.02)
