"""Offline estimate of the vote_center slab kernel's work: candidates / quanta / windows per scene (NumPy, approximate)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cppf2_amd import synth
N, T, R, Q = 4096, 20000, 180, 4
res = 0.0025 if len(sys.argv) < 2 else float(sys.argv[1])
sc = synth.make_scene(0, 0, N)
pc = sc["pc"].astype(np.float64)
idx = synth.host_sample_tuples(0, 0, T, 5, N)
rng = np.random.RandomState(1)
can = sc["pc_canon"].astype(np.float64)
a, b = pc[idx[:, 0]], pc[idx[:, 1]]
ca, cb = can[idx[:, 0]], can[idx[:, 1]]
# true vote parameters (what a good model predicts): proj_len, dist to axis
ab = a - b; nab = np.linalg.norm(ab, axis=1, keepdims=True); u = ab / nab
t = sc["t"].astype(np.float64)
proj = ((a - t) * u).sum(1); c = a - u * proj[:, None]
od = np.linalg.norm(c - t, axis=1)
co = np.cross(u, np.array([0, 0, 1.0])); co /= np.linalg.norm(co, axis=1, keepdims=True)
x = co * od[:, None]; y = np.cross(x, u)
c0 = pc.min(0); g = ((pc.max(0) - c0) / res).astype(int) + 1
G = g.prod(); SL = 36864; ns = (G + SL - 1) // SL; gyz = g[1] * g[2]
print("grid", g, "cells", G, "slabs", ns)
A = np.hypot(x[:, 0], y[:, 0]); phi = np.arctan2(y[:, 0], x[:, 0]) * R / (2 * np.pi)
th = np.arange(R) * 2 * np.pi / R
vx = c[:, 0:1] + x[:, 0:1] * np.cos(th) + y[:, 0:1] * np.sin(th)
ix = np.floor((vx - c0[0]) / res + 0.5)
tot_c = tot_q = tot_w = 0
for s in range(ns):
    lo = s * SL; n = min(SL, G - lo); xl = lo // gyz; xh = (lo + n - 1) // gyz
    inr = (ix >= xl) & (ix <= xh)
    cnt = inr.sum(1) + 2 * (inr.sum(1) > 0)           # ~ arc margin
    # two arcs when both non-contiguous: estimate quanta as ceil per arc (assume 2 arcs if count<R and >0)
    q = np.where(cnt > 0, np.ceil(cnt / 2 / Q) * 2, 0)
    w = 0
    for wv in range(0, T, 64):
        w += int(np.ceil(q[wv:wv + 64].sum() / 64))
    tot_c += cnt.sum(); tot_q += q.sum(); tot_w += w
    print("slab", s, "layers", xl, xh, "cand", int(cnt.sum()), "quanta", int(q.sum()), "windows", w, "fill %.2f" % (cnt.sum() / max(1, w * 64 * Q)))
print("total cand", int(tot_c), "of", T * R, "quanta", int(tot_q), "windows", tot_w, "lane-slot fill %.2f" % (tot_c / (tot_w * 64 * Q)))

# fraction of the x-arc candidates that also land inside the grid in y and z (how much a second arc constraint could cut)
vy = c[:, 1:2] + x[:, 1:2] * np.cos(th) + y[:, 1:2] * np.sin(th)
vz = c[:, 2:3] + x[:, 2:3] * np.cos(th) + y[:, 2:3] * np.sin(th)
iy = np.floor((vy - c0[1]) / res + 0.5); iz = np.floor((vz - c0[2]) / res + 0.5)
inx = (ix >= 1) & (ix < g[0]); iny = (iy >= 1) & (iy < g[1]); inz = (iz >= 1) & (iz < g[2])
print("rotations with x in grid: %.3f of all; of those, y in grid %.3f, z in grid %.3f, both %.3f"
      % (inx.mean(), (inx & iny).sum() / inx.sum(), (inx & inz).sum() / inx.sum(), (inx & iny & inz).sum() / inx.sum()))
