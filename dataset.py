"""`dataset` of the reference for the names its entry points import from it (eval.py:7,19,22: id2category, resize_crop, DINOV2;
train_dino.py:9,165 and train_shot.py:9: ShapeNetExportDataset, id2category, generate_target_pairs, rotx/roty/rotz), so those
lines resolve unchanged with this repository first on sys.path.  The ShapeNet / BlenderProc rendering datasets
(dataset.py:140-336, 367-412) are outside the voting path; ShapeNetExportDataset -- the reader of the reference's
EXPORTED training items, the only dataset class its trainers construct (train_shot.py:148, train_dino.py:159) -- is a thin adapter
over cppf2_amd.training.ExportedItems.
"""
import numpy as np
import torch

from cppf2_amd.ops import generate_target_pairs  # noqa: F401  dataset.py:118-135 -> cppf_generate_target_pairs (float64 on the GPU)
from cppf2_amd.ops import interpolate_features   # noqa: F401  dataset.py:40-59   -> cppf_interpolate_features
from utils.util import downsample                # noqa: F401  dataset.py:107-115 is utils/util.py:39-46 again

category2id = {"bottle": 1, "bowl": 2, "camera": 3, "can": 4, "laptop": 5, "mug": 6}          # dataset.py:29-37
id2category = {v: k for k, v in category2id.items()}


class ShapeNetExportDataset(torch.utils.data.Dataset):
    """dataset.py:338-364: `ShapeNetExportDataset(cfg, full_rot=False)` -- 200 items per epoch, each a pickled dict (pc, pc_canon,
    desc, bound, shot, normal) drawn from `data/category_training_data[_full_rot]/<cfg.category>/*.pkl` (resolved against the working
    directory, as hydra.utils.to_absolute_path does).  The reference draws with the unseeded global NumPy generator from the models
    that survive its blacklist files (data/shapenet_*.txt, data/blacklists.txt: not shipped with the reference); here every file of
    the directory is a candidate and the draw is seeded by (cfg.seed, item) -- cppf2_amd.training.ExportedItems, which raises
    FileNotFoundError naming the directory when there are no items.  `cfg.data_dir`, if set, overrides the directory."""

    def __init__(self, cfg, full_rot=False):
        super().__init__()
        import os
        from cppf2_amd.training import ExportedItems
        self.cfg = cfg
        self.category = cfg.category
        get = cfg.get if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
        self.root = str(get("data_dir") or os.path.abspath(os.path.join(
            "data", "category_training_data%s" % ("_full_rot" if full_rot else ""), str(cfg.category))))
        self.items = ExportedItems(self.root, length=200, seed=int(get("seed", 0) or 0))          # dataset.py:362-363: 200

    def __getitem__(self, idx):
        if idx >= len(self):
            raise IndexError("Index out of bounds")                                           # dataset.py:355-356
        return self.items[idx]

    def __len__(self):
        return len(self.items)


def _rot4(a, i, j):
    m = np.eye(4)
    c, s = np.cos(a), np.sin(a)
    m[i, i], m[i, j], m[j, i], m[j, j] = c, s, -s, c
    return m


def rotz(a):
    """dataset.py:84-88 (4x4, the reference's sign convention: [[c, s], [-s, c]] in the x-y block)."""
    return _rot4(a, 0, 1)


def roty(a):
    """dataset.py:91-95 ([[c, -s], [s, c]] in the x-z block)."""
    return _rot4(a, 2, 0)


def rotx(a):
    """dataset.py:97-101 ([[c, -s], [s, c]] in the y-z block)."""
    return _rot4(a, 2, 1)


def resize_crop(img, padding=0.2, out_size=224, bbox=None):
    """utils/util.py:3076-3091 (imported by eval.py:19 from dataset): square crop around `bbox` (default: the image's non-zero
    bounding box) enlarged by `padding`, resized to out_size; returns (crop uint8 [out,out,3], 3x3 map from crop pixels to image
    pixels)."""
    from PIL import Image
    im = Image.fromarray(img)
    if bbox is None:
        bbox = im.getbbox()
    size = max(bbox[3] - bbox[1], bbox[2] - bbox[0]) * (1 + padding)
    cx, cy = (bbox[2] + bbox[0]) / 2, (bbox[3] + bbox[1]) / 2
    # torchvision's functional.crop on a PIL image is Image.crop of the float box (PIL rounds each edge, zero-pads outside the
    # image); functional.resize's default is bilinear
    left, top = cx - size / 2, cy - size / 2
    im = im.crop((left, top, left + size, top + size)).resize((out_size, out_size), Image.BILINEAR)
    s = size / out_size
    transform = np.array([[s, 0, cx - s * out_size / 2], [0, s, cy - s * out_size / 2], [0, 0, 1.0]])
    return np.array(im), transform


class DINOV2(torch.nn.Module):
    """dataset.py:62-81: DINOv2 ViT-L/14 patch tokens sampled at the cloud's pixels.  The backbone is an INPUT of the voting
    path (its weights come from torch.hub and are not part of this build); the token interpolation is cppf_interpolate_features."""

    def __init__(self, stride=4):
        super().__init__()
        self.dinov2_vit = torch.hub.load("facebookresearch/dinov2", "dinov2_vitl14").eval()
        self.stride = stride
        self.mean = torch.tensor((0.485, 0.456, 0.406)).view(1, 3, 1, 1)
        self.std = torch.tensor((0.229, 0.224, 0.225)).view(1, 3, 1, 1)

    def forward(self, rgb, pts):
        ph, pw = rgb.shape[-2] // self.stride, rgb.shape[-1] // self.stride
        x = torch.nn.functional.interpolate(rgb[None], size=(ph * 14, pw * 14), mode="bilinear", antialias=True, align_corners=False)
        x = (x - self.mean.to(x)) / self.std.to(x)
        tokens = self.dinov2_vit.forward_features(x)["x_norm_patchtokens"].reshape(1, ph, pw, -1).permute(0, 3, 1, 2)
        return interpolate_features(tokens, pts[None], strides=self.stride, normalize=True)[0].T
