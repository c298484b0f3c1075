"""Config 3 throughput: eval.run_ensemble (DINO + SHOT models, both voted, alignment-loss selection) for one category at the
reference's sizes, instances per second.  usage: python scratch/eval_time.py [instances] [num_pairs] [num_rots]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import eval as ev
from cppf2_amd import synth, models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
P = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
R = int(sys.argv[3]) if len(sys.argv) > 3 else 180
dev = torch.device("cuda:0")
cfg, dino, shot = ev.load_category("mug", device=dev)
scenes = [synth.make_scene(0, s, 4096) for s in range(B)]
g = torch.Generator().manual_seed(1)
descs = [torch.nn.functional.normalize(torch.randn((4096, 1024), generator=g), dim=-1).numpy() for _ in scenes]
prior = ev._teacher_prior(np.concatenate([s["pc_canon"] for s in scenes]), dev)
for mode in ("split", "native"):
    models.MLP_ARITH = mode
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = ev.run_ensemble(cfg, dino, shot, [s["pc"] for s in scenes], descs, 0, list(range(B)), P, R, priors=prior,
                            scale_priors=np.stack([s["extent"] for s in scenes]))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: %d instances x %d pairs x %d rots, both models: %.1f ms = %.1f instances/s; picks %s" % (mode, B, P, R, dt * 1e3, B / dt, r["pick"].tolist()), flush=True)
