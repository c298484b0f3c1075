"""Training-side pieces shared by train_shot.py / train_dino.py: soft bin targets, losses, the synthetic
dataset that stands in for ShapeNetExportDataset (ShapeNet renders are not available), checkpoint I/O."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn.functional as F

from . import synth


def real2prob(val, max_val, num_bins):
    """Non-circular soft one-hot over `num_bins` knots (utils/util.py:215-239): weight 1-frac on the lower knot,
    frac on the upper one."""
    interval = max_val / (num_bins - 1)
    x = val / interval
    low = torch.clamp(torch.floor(x).long(), max=num_bins - 2)
    res = torch.zeros((*val.shape, num_bins), dtype=val.dtype, device=val.device)
    wl = 1.0 - (x - low)
    res.scatter_(-1, low[..., None], wl[..., None])
    res.scatter_(-1, (low + 1)[..., None], (1.0 - wl)[..., None])
    return res


def cppf_losses(preds_cls, preds_scale, pc_canon, point_idxs_all, target_scale):
    """train_shot.py:95-104: KL(log_softmax(logits) || soft targets of the pair's canonical coords) + MSE(scale)."""
    T = preds_cls.shape[0]
    with torch.no_grad():
        tgt = real2prob(torch.clamp(pc_canon[point_idxs_all[:, :2].long()], -0.5, 0.5) + 0.5, 1.0,
                        preds_cls.shape[-1]).reshape(T, 6, -1)
    loss_cls = F.kl_div(F.log_softmax(preds_cls, dim=-1), tgt, reduction="batchmean")
    loss_scale = F.mse_loss(preds_scale, target_scale[None].expand_as(preds_scale))
    return loss_cls, loss_scale


class SyntheticObjects:
    """Stand-in for ShapeNetExportDataset (dataset.py:341-364): items carry the same keys
    (pc, pc_canon, bound, shot, normal, desc); clouds have 100 points like the reference's pickles."""

    def __init__(self, cfg, length=200, n_points=100, seed=0, with_desc=False):
        self.cfg, self.length, self.n, self.seed, self.with_desc = cfg, length, n_points, seed, with_desc

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        sc = synth.make_scene(self.seed, i, self.n, max_tilt_deg=180.0)
        item = dict(pc=torch.from_numpy(sc["pc"]), pc_canon=torch.from_numpy(sc["pc_canon"]),
                    bound=torch.tensor([synth.RADIUS, synth.HEIGHT / 2, synth.RADIUS], dtype=torch.float32) / synth.DIAG)
        if self.with_desc:
            g = torch.Generator().manual_seed(self.seed * 100003 + i)
            item["desc"] = F.normalize(torch.randn((self.n, 1024), generator=g), dim=-1)
        return item


class ExportedItems:
    """The reference's dumped training items (dataset.py:341-364 reads them, :380-412 writes them): `<data_dir>/{:06d}.pkl`,
    each a dict with pc [n,3], pc_canon [n,3], desc [n,1024], bound [3], shot [n,352], normal [n,3] (float32; n = 100).
    The reference draws a file at random per item (dataset.py:357); here the draw is seeded (seed, item index) so that runs
    are reproducible.  `data_dir=<dir>` on the trainers' command line selects this instead of SyntheticObjects."""

    KEYS = ("pc", "pc_canon", "desc", "bound", "shot", "normal")

    def __init__(self, data_dir, length=200, seed=0):
        import glob
        self.files = sorted(glob.glob(os.path.join(data_dir, "*.pkl")))
        if not self.files:
            raise FileNotFoundError("no *.pkl training items under %r (dataset.py:344 layout: data/category_training_data/<category>/)" % data_dir)
        self.length, self.seed = int(length), int(seed)

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        import pickle
        pick = int(np.random.RandomState((self.seed * 1000003 + int(i)) % (2 ** 32)).randint(len(self.files)))
        with open(self.files[pick], "rb") as f:
            d = pickle.load(f)
        return {k: torch.from_numpy(np.ascontiguousarray(np.asarray(d[k], dtype=np.float32))) for k in self.KEYS if k in d}


def make_dataset(cfg, with_desc=False):
    """SyntheticObjects, or the reference's exported items when the command line gives data_dir=<dir>."""
    length = int(cfg.get("iters_per_epoch", 200))                   # dataset.py:363: 200 items per epoch
    if cfg.get("data_dir"):
        return ExportedItems(str(cfg.data_dir), length=length, seed=int(cfg.get("seed", 0)))
    return SyntheticObjects(cfg, length=length, with_desc=with_desc)


def checkpoint_dir(run_directory):
    """Where Lightning's ModelCheckpoint under TensorBoardLogger(save_dir=<run dir>) puts its files (train_shot.py:136-142),
    and where eval.py:93,98 reads last.ckpt from."""
    return os.path.join(run_directory, "lightning_logs", "version_0", "checkpoints")


def save_checkpoint(model, path, epoch):
    """Lightning-style container: weights under 'state_dict' with the reference's key names."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save({"state_dict": model.state_dict(), "epoch": epoch}, path)
