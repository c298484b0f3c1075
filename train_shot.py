#!/usr/bin/env python
"""Entry point mirroring the reference's train_shot.py (hydra app `train(cfg)`, train_shot.py:133-150):

    python train_shot.py category=bottle [opt.lr=1e-3 max_epochs=101 iters_per_epoch=200]

Same recipe: 10 000 tuples sampled per step (train_shot.py:88), SHOT encoder + tuple MLP, KL + MSE losses,
Adam + StepLR(25, 0.5) (train_shot.py:124-130), a checkpoint every 10 epochs + last.ckpt.  The tuple sampler, the
SHOT352 descriptor and the tuple encode (with its feature-table backward) run through the HIP library.
Data: ShapeNet renders are not available here -> cppf2_amd.training.SyntheticObjects; `data_dir=<dir>` reads the reference's
dumped items instead (dataset.py:341-364: {:06d}.pkl with pc, pc_canon, desc, bound, shot, normal).
Output: the reference's run directory, hydra.run.dir = checkpoints/${cat_name} (override: hydra.run.dir=...):
.hydra/config.yaml + lightning_logs/version_0/checkpoints/{epoch=N,last}.ckpt -- e.g.
    python train_shot.py category=bottle 'hydra.run.dir=ckpts/shot/${cat_name}-num_more-3'
is what eval.py --ckpt_dir ckpts then loads (eval.py:91-99).
"""
import sys

import torch

from cppf2_amd import ops, shot
from cppf2_amd.config import load_config, run_dir, save_run_config
from cppf2_amd.models import BeyondCPPFShot
from cppf2_amd.models import BeyondCPPFShot as BeyondCPPF  # noqa: F401  (the class name in the reference's file: eval.py:18 imports it)
from cppf2_amd.training import checkpoint_dir, cppf_losses, make_dataset, save_checkpoint


def train(cfg, hydra_node=None):
    dev = ops._dev()
    model = BeyondCPPFShot(cfg).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=cfg.opt.lr, weight_decay=cfg.opt.weight_decay)
    sched = torch.optim.lr_scheduler.StepLR(opt, 25, 0.5)
    ds = make_dataset(cfg)
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
            points, pc_canon = item["pc"].to(dev), item["pc_canon"].to(dev)
            n = points.shape[0]
            idx = ops.sample_tuples(n, 10000, k, seed=step, scene_ids=(0,), device=dev)          # train_shot.py:88
            pt_off = ops._offsets([n], dev)
            if "shot" in item and "normal" in item:        # exported items carry the descriptor of the full cloud (dataset.py:278,310)
                sfeat, normal = torch.nan_to_num(item["shot"].to(dev)), torch.nan_to_num(item["normal"].to(dev))
            else:
                with torch.no_grad():
                    sfeat, normal = shot.compute_device(points, pt_off, cfg.res * 10, cfg.res * 10)  # dataset.py:278
                    sfeat, normal = torch.nan_to_num(sfeat), torch.nan_to_num(normal)
            preds_cls, preds_scale = model(points, idx, sfeat, normal)
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
