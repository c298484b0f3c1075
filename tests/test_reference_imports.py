"""The reference's own import lines resolve, unchanged, against this repository's root (the drop-in boundary):
eval.py:5,7,15-19,22, train_shot.py:16, train_dino.py:9,15,18,165,168, dataset.py:4,9,12,20 -- written out below BY NAME -- and
the host helpers behind those names return what the reference's return (tests/golden/util_helpers.npz, generated from the real
reference by tests/golden/make_golden_util.py).  No GPU: importing loads libcppf_hip.so but calls nothing on a device."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def helpers():
    return dict(np.load(os.path.join(GOLDEN, "util_helpers.npz")))


def test_reference_import_lines_resolve_by_name():
    assert sys.path[0] == ROOT or ROOT in sys.path
    # eval.py:5 (demo.py:5 is the same line)
    from utils.util import (downsample, backproject, dilate_mask, fibonacci_sphere, real2prob, prob2real,  # noqa: F401
                            calculate_2d_projections, draw, get_3d_bbox, process_data, transform_coordinates_3d,
                            compute_degree_cm_mAP)
    # eval.py:7, 19 (eval.py:22's DINOV2 is checked below: it is a class whose constructor needs the hub weights)
    from dataset import id2category
    from dataset import resize_crop  # noqa: F401
    # train_dino.py:9 and train_shot.py:9, the literal lines
    from dataset import ShapeNetExportDataset, id2category  # noqa: F401,F811
    from dataset import ShapeNetExportDataset  # noqa: F401,F811
    # eval.py:15-18
    from src_shot.build import shot
    from train_dino import vote_center, vote_rotation, generate_target_pairs
    from train_dino import BeyondCPPF as BeyondCPPFDINO
    from train_shot import BeyondCPPF as BeyondCPPFSHOT
    # train_dino.py:165, 168; dataset.py:4, 9, 20
    from dataset import generate_target_pairs as gtp2, rotx, roty, rotz  # noqa: F401
    from dataset import DINOV2, interpolate_features  # noqa: F401
    import cppf2_amd
    from cppf2_amd import models, ops
    here = os.path.dirname(os.path.abspath(cppf2_amd.__file__))
    for mod in ("utils.util", "dataset", "src_shot.build.shot", "train_dino", "train_shot"):
        assert os.path.abspath(sys.modules[mod].__file__).startswith(ROOT), (mod, sys.modules[mod].__file__)
    assert here.startswith(ROOT)
    assert id2category == {1: "bottle", 2: "bowl", 3: "camera", 4: "can", 5: "laptop", 6: "mug"}
    # the names are the library-backed callables, not copies with other behaviour
    assert vote_center is ops.vote_center and vote_rotation is ops.vote_rotation and generate_target_pairs is ops.generate_target_pairs
    assert gtp2 is ops.generate_target_pairs
    assert BeyondCPPFDINO is models.BeyondCPPFDino and BeyondCPPFSHOT is models.BeyondCPPFShot
    from cppf2_amd import shot as lib_shot
    assert shot.compute is lib_shot.compute and shot.estimate_normal is lib_shot.estimate_normal
    assert shot.compute_color is lib_shot.compute_color
    assert fibonacci_sphere is ops.fibonacci_sphere and callable(compute_degree_cm_mAP) and callable(process_data)
    assert issubclass(DINOV2, torch.nn.Module)


def test_shapenet_export_dataset_reads_the_reference_item_layout(tmp_path, monkeypatch):
    """dataset.py:338-364: ShapeNetExportDataset(cfg) = 200 items per epoch out of <cwd>/data/category_training_data/<category>/
    {:06d}.pkl, each a dict of float32 arrays; an empty directory fails loudly with its path."""
    import pickle
    from types import SimpleNamespace
    from dataset import ShapeNetExportDataset
    monkeypatch.chdir(tmp_path)
    cfg = SimpleNamespace(category=1, seed=3)
    with pytest.raises(FileNotFoundError, match="category_training_data"):
        ShapeNetExportDataset(cfg)
    d = tmp_path / "data" / "category_training_data" / "1"
    d.mkdir(parents=True)
    rng = np.random.RandomState(0)
    for i in range(7):
        item = dict(pc=rng.rand(100, 3), pc_canon=rng.rand(100, 3) - 0.5, desc=rng.rand(100, 1024), bound=rng.rand(3),
                    shot=rng.rand(100, 352), normal=rng.rand(100, 3))
        with open(d / ("%06d.pkl" % i), "wb") as f:
            pickle.dump({k: v.astype(np.float32) for k, v in item.items()}, f)
    ds = ShapeNetExportDataset(cfg)
    assert len(ds) == 200 and isinstance(ds, torch.utils.data.Dataset)
    it = ds[5]
    assert sorted(it) == ["bound", "desc", "normal", "pc", "pc_canon", "shot"] and it["shot"].shape == (100, 352) and it["pc"].dtype == torch.float32
    assert torch.equal(ds[5]["pc"], it["pc"])                      # seeded draw
    with pytest.raises(IndexError):
        ds[200]
    full = tmp_path / "data" / "category_training_data_full_rot" / "1"
    full.mkdir(parents=True)
    with pytest.raises(FileNotFoundError, match="_full_rot"):
        ShapeNetExportDataset(cfg, full_rot=True)


def test_real2prob_prob2real_match_the_reference(helpers):
    from utils.util import real2prob, prob2real
    h = helpers
    assert np.array_equal(real2prob(h["r2p_in"].copy(), 1.0, 32), h["r2p_np"])
    assert np.array_equal(real2prob(torch.from_numpy(h["r2p_in"].copy()), 1.0, 32).numpy(), h["r2p_torch"])
    assert np.array_equal(real2prob(h["r2p_circ_in"].copy(), 2 * np.pi, 36, True), h["r2p_circ_np"])
    assert np.array_equal(real2prob(torch.from_numpy(h["r2p_circ_in"].copy()), 2 * np.pi, 36, True).numpy(), h["r2p_circ_torch"])
    # expectations: same sums, possibly associated differently (sum of products vs product then sum)
    assert np.allclose(prob2real(h["p2r_in"], 1.0, 32), h["p2r_np"], rtol=0, atol=1e-6)
    assert np.allclose(prob2real(torch.from_numpy(h["p2r_in"]), 1.0, 32).numpy(), h["p2r_torch"], rtol=0, atol=1e-6)
    assert np.allclose(prob2real(h["p2r_circ_in"], 2 * np.pi, 36, True), h["p2r_circ_np"], rtol=0, atol=1e-12)
    assert np.allclose(prob2real(torch.from_numpy(h["p2r_circ_in"]), 2 * np.pi, 36, True).numpy(), h["p2r_circ_torch"], rtol=0, atol=1e-12)
    # round trip: the expectation of a soft one-hot is the value
    v = np.linspace(0, 1, 23)
    assert np.allclose(prob2real(real2prob(v, 1.0, 32), 1.0, 32), v, atol=1e-12)


def test_box_and_projection_helpers_match_the_reference(helpers):
    from utils.util import get_3d_bbox, transform_coordinates_3d, calculate_2d_projections, draw
    from dataset import rotx, roty, rotz
    h = helpers
    assert np.array_equal(get_3d_bbox(h["bbox_scale"], 0), h["bbox_vec"])
    assert np.array_equal(get_3d_bbox(0.3, 0.05), h["bbox_scalar"])
    cam = transform_coordinates_3d(h["bbox_vec"], h["RT"])
    assert np.array_equal(cam, h["bbox_cam"])
    px = calculate_2d_projections(cam, h["K"])
    assert px.dtype == np.int32 and np.array_equal(px, h["bbox_px"])
    for name, fn in (("rotx", rotx), ("roty", roty), ("rotz", rotz)):
        assert np.array_equal(fn(0.37), h[name])
    # draw: the box edges and the three axes land on the image (eval.py:386-392's call shapes)
    img = np.zeros((480, 640, 3), np.uint8)
    axes = calculate_2d_projections(transform_coordinates_3d(np.array([[0, 0, 0], [0, 0, .1], [0, .1, 0], [.1, 0, 0]]).T, h["RT"]), h["K"])
    out = draw(img, px, axes, (255, 0, 0))
    assert out.shape == (480, 640, 3) and (out.reshape(-1, 3).max(0) == 255).all()
    for p in px:
        if 0 <= p[0] < 640 and 0 <= p[1] < 480:
            assert out[p[1], p[0]].any()


def test_dilate_mask_and_resize_crop():
    from utils.util import dilate_mask
    from dataset import resize_crop
    m = np.zeros((40, 50), bool)
    m[10:20, 10] = True
    m[10, 10:25] = True           # an L: its convex hull is the triangle
    m[35, 45] = True              # a second, smaller component
    out = dilate_mask(m, size=5, largest_comp=True)
    assert out.dtype == np.uint8 and out[12, 12] == 1 and out[8, 8] == 1 and out[35, 45] == 0
    assert out[19, 24] == 0 and out.sum() > m.sum()
    both = dilate_mask(m, size=3)
    assert both[35, 45] == 1 or both[30, 40] == 1     # hull of both components reaches towards the far pixel
    img = np.zeros((60, 80, 3), np.uint8)
    img[20:40, 30:50] = 200
    crop, tf = resize_crop(img, bbox=(30, 20, 50, 40), padding=0, out_size=32)
    assert crop.shape == (32, 32, 3) and crop.min() > 150
    # the transform maps crop pixel coordinates to image coordinates: crop centre -> box centre, crop corner -> box corner
    assert np.allclose(tf @ [16, 16, 1], [40, 30, 1]) and np.allclose(tf @ [0, 0, 1], [30, 20, 1])


def test_process_data_parses_a_nocs_frame(tmp_path):
    from utils.util import process_data
    root = tmp_path / "obj_models"
    (root / "real_test").mkdir(parents=True)
    np.savetxt(root / "real_test" / "bottle_x_norm.txt", [0.1, 0.3, 0.1])
    np.savetxt(root / "real_test" / "mug_y_norm.txt", [0.2, 0.2, 0.3])
    meta = tmp_path / "0000_meta.txt"
    meta.write_text("1 1 bottle_x_norm\n2 6 mug_y_norm\n")
    mask = np.full((8, 10), 255, np.uint8)
    mask[1:3, 1:4] = 1
    mask[5:7, 5:9] = 2
    coord = np.full((8, 10, 3), 51, np.uint8)       # 0.2 in every channel
    masks, coords, cls, scales, words = process_data(mask, coord, {1: 1, 2: 6, 3: 0}, str(meta), model_root=str(root))
    assert masks.shape == (8, 10, 2) and masks[..., 0].sum() == 6 and masks[..., 1].sum() == 8
    assert cls.tolist() == [1, 6] and words == [["1", "1", "bottle_x_norm"], ["2", "6", "mug_y_norm"]]
    assert np.allclose(np.linalg.norm(scales, axis=1), 1.0) and np.allclose(scales[0] * np.sqrt(0.11), [0.1, 0.3, 0.1], atol=1e-6)
    assert np.allclose(coords[1, 1, 0], [0.2, 0.2, 0.8]) and not coords[0, 0].any()
