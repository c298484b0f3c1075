#!/usr/bin/env python
"""Entry point mirroring the reference's train_dino.py (`train(cfg)`, train_dino.py:142-161) and re-exporting the
voting callables the reference defines in that file (vote_center, vote_rotation: train_dino.py:171-239), so that
`from train_dino import vote_center, vote_rotation, generate_target_pairs` (eval.py:16) keeps working.

    python train_dino.py category=bottle [max_epochs=101 iters_per_epoch=200]

DINOv2 features are inputs to the path; without the hub weights the synthetic dataset supplies seeded unit vectors.
"""
import sys

import torch

from cppf2_amd import ops
from cppf2_amd.config import load_config, run_dir, save_run_config
from cppf2_amd.models import BeyondCPPFDino as BeyondCPPF  # noqa: F401  (name used by the reference's eval.py:17)
from cppf2_amd.ops import generate_target_pairs, vote_center, vote_rotation  # noqa: F401
from cppf2_amd.training import checkpoint_dir, cppf_losses, make_dataset, save_checkpoint


def train(cfg, hydra_node=None):
    dev = ops._dev()
    model = BeyondCPPF(cfg).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=cfg.opt.lr, weight_decay=cfg.opt.weight_decay)
    sched = torch.optim.lr_scheduler.StepLR(opt, 25, 0.5)
    ds = make_dataset(cfg, with_desc=True)
    k = cfg.num_more + 2
    # the reference's run directory (config/config.yaml:16-22: hydra.run.dir = checkpoints/${cat_name}): the resolved cfg under
    # .hydra/config.yaml, the weights under lightning_logs/version_0/checkpoints/{epoch=N,last}.ckpt (train_shot.py:136-142) --
    # the layout eval.py:91-99 loads a category from
    run = run_dir(cfg, hydra_node)
    save_run_config(cfg, run)
    out_dir = checkpoint_dir(run)
    step = 0
    for epoch in range(int(cfg.get("max_epochs", 101))):
        for i in range(len(ds)):
            item = ds[(epoch * len(ds) + i) % 100000]
            points, pc_canon, desc = item["pc"].to(dev), item["pc_canon"].to(dev), item["desc"].to(dev)
            idx = ops.sample_tuples(points.shape[0], 10000, k, seed=step, scene_ids=(0,), device=dev)  # train_dino.py:103
            preds_cls, preds_scale = model(points, desc, idx)
            loss_cls, loss_scale = cppf_losses(preds_cls, preds_scale, pc_canon, idx, item["bound"].to(dev))
            loss = loss_cls + loss_scale
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            step += 1
        sched.step()
        print("epoch %d cls %.4f scale %.5f lr %.2e" % (epoch, float(loss_cls.detach()), float(loss_scale.detach()), sched.get_last_lr()[0]),
              flush=True)
        if epoch % 10 == 0:
            save_checkpoint(model, "%s/epoch=%d.ckpt" % (out_dir, epoch), epoch)
        save_checkpoint(model, "%s/last.ckpt" % out_dir, epoch)
    return model


if __name__ == "__main__":
    train(*load_config("config", "config", sys.argv[1:], with_hydra=True))
