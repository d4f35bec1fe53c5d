"""A Lightning-free training loop for ``Px2Px_PL`` ("train.py drops in without Lightning").

What the reference delegates to ``pytorch_lightning.Trainer.fit`` (train.py:118-136) and what of it touches the hot
path: per batch the two optimizer passes (here ONE fused call, ``model.train_batch``), per epoch the validation scalars
(``validation_step``: val/L1, val/L2, val/PSNR, val/SSIM) and ``ReduceLROnPlateau`` on ``config.Schedulers.metric``
for both optimizers (model/pix2pix.py:485-492), plus a checkpoint in Lightning's layout: ``state_dict`` with the
reference's keys (train.py:61-65 / create_synthetic_dataset.py:24-26 load it with ``strict=False``), ``optimizer_states``
(both Adams: moments and step counts, torch.optim.Adam's format) and ``lr_schedulers`` (both ReduceLROnPlateau), so that
``resume_from`` continues a run the way ``Trainer(resume_from_checkpoint=...)`` does (train.py:66-70,126).  Loggers, wandb, image plots and
callbacks are out of scope.  Data parallel: pass a ``parallel.GradReducer`` (one process per GPU, RCCL); validation
metrics are averaged over ranks when a process group is initialised.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, Optional

import torch


def _to_device(batch: dict, device) -> dict:
    return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}


def _rank_mean(value: float, device) -> float:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t)
        return float(t.item()) / dist.get_world_size()
    return value


def fit(model, train_loader: Iterable[dict], val_loader: Optional[Iterable[dict]] = None, *, max_epochs: int = 1,
        device=None, reducer=None, log_every: int = 10, on_log: Optional[Callable[[Dict[str, float]], None]] = None,
        ckpt_path: Optional[str] = None, resume_from: Optional[str] = None) -> Dict[str, list]:
    """Train ``model`` (model.pix2pix.Px2Px_PL).  Returns the history {'train': [...], 'val': [...], 'lr': [...]}."""
    device = device or next(model.parameters()).device
    trainer = model.fused_trainer(reducer=reducer)
    (optim_d, optim_g), scheds = model.configure_optimizers()
    sched_d, sched_g = scheds[0]["scheduler"], scheds[1]["scheduler"]
    monitor = scheds[0]["monitor"]
    history = {"train": [], "val": [], "lr": []}
    step, first_epoch = 0, 0
    if resume_from is not None:
        ck = torch.load(resume_from, map_location=device, weights_only=False)
        model.load_state_dict(ck["state_dict"], strict=False)
        trainer.flatG.touch()
        trainer.flatD.touch()
        if "optimizer_states" in ck:                     # order of configure_optimizers: [D, G] (pix2pix.py:490)
            optim_d.load_state_dict(ck["optimizer_states"][0])
            optim_g.load_state_dict(ck["optimizer_states"][1])
            trainer.lr_d, trainer.lr_g = optim_d.param_groups[0]["lr"], optim_g.param_groups[0]["lr"]
        for sch, sd in zip((sched_d, sched_g), ck.get("lr_schedulers", [])):
            sch.load_state_dict(sd)
        step, first_epoch = int(ck.get("global_step", 0)), int(ck.get("epoch", -1)) + 1
    for epoch in range(first_epoch, max_epochs):
        model.train()
        for batch in train_loader:
            view = model.train_batch(_to_device(batch, device))
            if log_every and step % log_every == 0:          # reading the losses synchronises: not every step
                rec = {"epoch": epoch, "step": step, **view.as_dict()}
                history["train"].append(rec)
                if on_log:
                    on_log(rec)
            step += 1
        if val_loader is not None:
            model.eval()
            sums, n = {}, 0
            for i, batch in enumerate(val_loader):
                model.logged.clear() if hasattr(model, "logged") else None
                model.validation_step(_to_device(batch, device), i)
                for k, v in getattr(model, "logged", {}).items():
                    if k.startswith("val/"):
                        sums[k] = sums.get(k, 0.0) + float(v)
                n += 1
            val = {k: _rank_mean(v / max(n, 1), device) for k, v in sums.items()}
            val["epoch"] = epoch
            history["val"].append(val)
            if on_log:
                on_log(val)
            if monitor in val:                               # ReduceLROnPlateau, interval 'epoch' (pix2pix.py:488-492)
                sched_d.step(val[monitor])
                sched_g.step(val[monitor])
                trainer.lr_d, trainer.lr_g = optim_d.param_groups[0]["lr"], optim_g.param_groups[0]["lr"]
        history["lr"].append({"epoch": epoch, "lr_d": trainer.lr if trainer.lr_d is None else trainer.lr_d,
                              "lr_g": trainer.lr if trainer.lr_g is None else trainer.lr_g})
        if ckpt_path is not None and (reducer is None or getattr(reducer, "rank", 0) == 0):
            torch.save({"epoch": epoch, "global_step": step, "state_dict": model.state_dict(),
                        "optimizer_states": [optim_d.state_dict(), optim_g.state_dict()],
                        "lr_schedulers": [sched_d.state_dict(), sched_g.state_dict()]}, ckpt_path)
    return history
