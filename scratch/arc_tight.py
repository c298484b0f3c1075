"""Offline replay of vote_slab_item's arc set-up (slab_arcs) in NumPy float32: candidates / quanta / windows per scene
for the outward-rounded arcs (floor/ceil, margin 0.25) and the tight ones (ceil/floor, margin 0.02)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cppf2_amd import synth
F = np.float32
N, T, R = 4096, 20000, 180
res = 2e-3
sc = synth.make_scene(0, 0, N)
pc = sc["pc"].astype(np.float64)
idx = synth.host_sample_tuples(0, 0, T, 5, N)
rng = np.random.RandomState(1)
a, b = pc[idx[:, 0]], pc[idx[:, 1]]
ab = a - b; nab = np.linalg.norm(ab, axis=1, keepdims=True); u = ab / np.maximum(nab, 1e-9)
t = sc["t"].astype(np.float64) + rng.normal(0, 0.004, (T, 3))      # noisy predictions
proj = ((a - t) * u).sum(1); c = a - u * proj[:, None]
od = np.linalg.norm(c - t, axis=1)
co = np.stack([0 * u[:, 0], -u[:, 2], u[:, 1]], -1); co /= np.maximum(np.linalg.norm(co, axis=1, keepdims=True), 1e-9)
x = co * od[:, None]; y = np.cross(x, u)
c0 = pc.min(0); g = ((pc.max(0) - c0) / res).astype(int) + 2
G = int(g.prod()); SL = 36864; ns = (G + SL - 1) // SL; gyz = int(g[1] * g[2])
print("grid", g, "cells", G, "slabs", ns)
A = np.hypot(x[:, 0], y[:, 0]); invA = np.where(A > 1e-12, 1 / A, 0)
kappa = R / (2 * np.pi)
phi = np.arctan2(y[:, 0], x[:, 0]) * kappa
cx = c[:, 0]
th = np.arange(R) * 2 * np.pi / R
vx = cx[:, None] + x[:, 0:1] * np.cos(th) + y[:, 0:1] * np.sin(th)
ix = np.floor((vx - c0[0]) / res + 0.5)


def arcs(xl, xh, tight, m):
    slop = 0.01 * res + 4e-7 * (np.abs(cx) + abs(c0[0]))
    L = (xl - 0.5) * res + c0[0] - cx - slop
    U = (xh + 0.5) * res + c0[0] - cx + slop
    cl, cu = L * invA - 1e-6, U * invA + 1e-6
    none = (cl > 1) | (cu < -1)
    amin = np.arccos(np.minimum(cu, 1).clip(-1, 1)) * kappa
    amax = np.arccos(np.maximum(cl, -1).clip(-1, 1)) * kappa
    half = 0.5 * R
    if tight:
        near, far = amin <= 0.5, (half - amax) <= 0.5
        lo_f, hi_f = np.ceil, np.floor
    else:
        near, far = amin <= m + 1, (half - amax) <= m + 1
        lo_f, hi_f = np.floor, np.ceil
    n0 = np.zeros(T); n1 = np.zeros(T)
    both = near & far
    a0 = np.where(near, lo_f(phi - amax - m), lo_f(phi + amin - m))
    b0 = np.where(near, hi_f(phi + amax + m), np.where(far, hi_f(phi + R - amin + m), hi_f(phi + amax + m)))
    n0 = np.maximum(b0 - a0 + 1, 0)
    two = ~near & ~far
    a1 = lo_f(phi - amax - m); n1 = np.where(two, np.maximum(hi_f(phi - amin + m) - a1 + 1, 0), 0)
    n0 = np.where(both, R, n0); n1 = np.where(both, 0, n1)
    full = n0 >= R
    n0 = np.where(full, R, n0); n1 = np.where(full, 0, n1)
    n0 = np.where(none, 0, n0); n1 = np.where(none, 0, n1)
    z = invA == 0
    n0 = np.where(z, np.where((L <= 0) & (0 <= U), R, 0), n0); n1 = np.where(z, 0, n1)
    return n0, n1


for tight, m, Q in ((0, 0.25, 4), (1, 0.02, 4), (1, 0.02, 8), (0, 0.25, 8), (1, 0.02, 2)):
    tc = tq = tw = real = touch = 0
    for s in range(ns):
        lo = s * SL; n = min(SL, G - lo); xl = lo // gyz; xh = (lo + n - 1) // gyz
        n0, n1 = arcs(xl, xh, tight, m)
        inr = ((ix >= xl) & (ix <= xh)).sum(1)
        assert np.all(n0 + n1 >= inr), (s, (n0 + n1 - inr).min())
        q = np.ceil(n0 / Q) + np.ceil(n1 / Q)
        w = sum(int(np.ceil(q[i:i + 64].sum() / 64)) for i in range(0, T, 64))
        tc += (n0 + n1).sum(); tq += q.sum(); tw += w; real += inr.sum(); touch += (n0 + n1 > 0).sum()
    print("tight" if tight else "loose", "m", m, "Q", Q, "real votes in slabs", int(real), "candidates", int(tc), "(%.3f x)" % (tc / real),
          "quanta", int(tq), "slots", int(tq * Q), "(%.3f x)" % (tq * Q / real), "windows", tw, "window slots", tw * 64 * Q, "(%.3f x)" % (tw * 64 * Q / real),
          "pair-slab touches %.2f per pair" % (touch / T))
