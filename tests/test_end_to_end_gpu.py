"""End to end at BASELINE's full size, inside the suite: the bench's Step (sampler -> normals + SHOT352 -> point MLP -> tuple MLP with
the pair features built and the bins drawn inside its launches -> vote parameters -> centre vote -> back-vote filter -> rotation
votes -> scale head -> pose; every launch a kernel of libcppf_hip.so) against the CPU oracle's whole-scene restatement
(oracle/pipeline_oracle.py: C SHOT in PCL's arithmetic, NumPy float32 MLP, eval.py:225-313 in NumPy / C), 4096 points x 20 000
tuples x 180 rotations, on the headline's clouds and on a cloud at voxel-grid density.  The bar (north_star): vote-grid arg-max
bit-exact, rotation bins equal, centre within 1e-3 m, rotation within 0.1 deg -- here: equal arg-max, equal bins, bit-equal
translation, rotation within 1e-4 deg; and at most a handful of the 120 000 bin draws per scene apart (the MLP's arithmetic at a
CDF edge).  bench.py does the same for all 64 scenes of its timed batch (oracle_agreement); this is the suite's own copy."""
import os
import sys
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pipeline_oracle as PO      # noqa: E402  (the checker)


def _args(scenes):
    return types.SimpleNamespace(scenes_per_gpu=scenes, points=4096, tuples=20000, rots=180, seed=0, vote_mode=0, eager_scale_head=False)


@pytest.mark.parametrize("cloud,scenes", [("synthetic", 2), ("voxel2mm", 1)])
def test_full_size_step_equals_the_whole_scene_oracle(cloud, scenes):
    from cppf2_amd.benchlib import workloads as W
    from cppf2_amd.metrics import rt_degree_cm
    dev = torch.device("cuda")
    st = W.Step(_args(scenes), 0, 1, dev, cloud=cloud)
    st.run()
    torch.cuda.synchronize()
    rec = st.pipe.results_to_numpy()
    bins = st.pipe.bins.reshape(scenes, st.T, 6).cpu().numpy()
    prior = st.prior_dense.reshape(scenes, st.T, 6, 32).cpu().numpy()          # the teacher prior's values, as the kernels add them
    weights = {k: v.detach().cpu().numpy() for k, v in st.model.state_dict().items()}
    trig = (st.pipe.cs.cpu().numpy(), st.pipe.sn.cpu().numpy())
    for b in range(scenes):
        o = PO.run_scene_full(weights, st.scenes[b]["pc"], 0, st.scene0 + b, st.T, res=W.Cfg.res, num_rots=180, trig=trig,
                              prior_fn=lambda idx, b=b: prior[b], topk_impl="c")
        assert int(rec["argmax"][b]) == o["argmax"]
        assert int(rec["up_idx"][b]) == o["up_idx"] and int(rec["right_idx"][b]) == o["right_idx"]
        assert int(rec["kept"][b]) == int(o["pairs_mask"].sum())
        assert np.array_equal(rec["t"][b], np.asarray(o["T_est"], dtype=rec["t"].dtype))
        m1, m2 = np.eye(4), np.eye(4)
        m1[:3, :3], m1[:3, 3] = rec["R"][b], rec["t"][b]
        m2[:3, :3], m2[:3, 3] = o["R_est"], o["T_est"]
        deg, cm = rt_degree_cm(m1, m2, "bottle", clip=True)
        assert deg < 1e-4 and cm == 0.0
        assert np.abs(rec["R"][b] - o["R_est"]).max() < 1e-6
        assert int((bins[b] != o["bins"]).sum()) <= 6, "bin draws apart from the oracle's: more than the MLP's arithmetic explains"
        # the scene really was solved (teacher prior): the centre is the synthetic ground truth's
        assert np.linalg.norm(rec["t"][b] - st.scenes[b]["t"]) < 5e-3
