"""Host-side profile (cProfile) of eval.run_ensemble at the reference's sizes.  usage: python scratch/eval_profile.py [instances] [pairs]"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import eval as ev
from cppf2_amd import synth, models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
P = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
dev = torch.device("cuda:0")
cfg, dino, shot = ev.load_category("mug", device=dev)
scenes = [synth.make_scene(0, s, 4096) for s in range(B)]
g = torch.Generator().manual_seed(1)
descs = [torch.nn.functional.normalize(torch.randn((4096, 1024), generator=g), dim=-1).numpy() for _ in scenes]
prior = ev._teacher_prior(np.concatenate([s["pc_canon"] for s in scenes]), dev)
def run():
    r = ev.run_ensemble(cfg, dino, shot, [s["pc"] for s in scenes], descs, 0, list(range(B)), P, 180, priors=prior,
                        scale_priors=np.stack([s["extent"] for s in scenes]))
    torch.cuda.synchronize()
    return r
for _ in range(3): run()
t0 = time.perf_counter(); run(); print("one call: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
