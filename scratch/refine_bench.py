import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
exec(open(os.path.join(os.path.dirname(__file__), "vc_bench.py")).read().split("fn = {")[0])
pipe.assemble()
torch.cuda.synchronize()
saved = pipe.results.clone()
for y in (True, False):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pipe.results.copy_(saved); pipe.refine(pts, idx, y); torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        pipe.results.copy_(saved); pipe.refine(pts, idx, y)
    e1.record(); torch.cuda.synchronize()
    print("refine y_only=%s B=%d: %.3f ms" % (y, B, e0.elapsed_time(e1) / 5))
