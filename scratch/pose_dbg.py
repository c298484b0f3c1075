import sys, os, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
sys.argv = ["bench.py", "--cpu-scenes", "0"]
args = bench.parse()
dev = torch.device("cuda"); step = bench.Step(args, 0, 1, dev); step.run()
torch.cuda.synchronize()
res = step.pipe.results_to_numpy()
for b in range(step.B):
    sc = step.scenes[b]
    terr = np.linalg.norm(res["t"][b] - sc["t"])
    cosang = abs(float(res["R"][b][:, 1] @ sc["R"][:, 1]))
    ang = np.degrees(np.arccos(min(cosang, 1)))
    if terr > 0.05 or ang > 5:
        print(b, "terr %.4f ang %.2f peak %d kept %d upcnt %.1f" % (terr, ang, res["peak"][b], res["kept"][b], res["up_count"][b]), "tilt", np.degrees(np.arccos(sc["R"][1,1])))
