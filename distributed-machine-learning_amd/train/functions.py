"""Function trainables: the single-trial training loops of the two reference workflows.

``train_dummy_model``       <-> /root/reference/ray-tune-hpo-regression-sample.py:88-135
``train_transformer_model`` <-> /root/reference/ray-tune-hpo-regression.py:260-373

Both keep the reference's signature ``fn(config, data_loader, val_loader)`` (bind the loaders with
``mitune.with_parameters``) and its metric names (``validation_loss``, ``validation_mape``).  What
changed, on purpose:

* a result is reported after EVERY epoch -- the reference reports once, after training, which gives
  ASHA nothing to prune on;
* ``input_size`` comes from the data (the reference hard-codes 10 against 81 real features);
* the warm-up/decay schedule is stepped per optimiser step (stepped per epoch against step-scale
  constants it pins the learning rate near zero);
* each optimiser receives only the arguments it accepts;
* losses are accumulated on the device and read once per epoch instead of ``.item()`` per batch;
* an in-memory checkpoint (model + optimiser ``state_dict``) rides along with each report so PBT and
  failure retries can resume, and ``mitune.get_checkpoint()`` is honoured at start.
"""
from __future__ import annotations

import logging
from typing import Any, Dict, Iterable, Optional

import torch

from ..models.transformer import SimpleTransformerModel, TransformerModel
from ..tune import session
from .losses import MAPE_EPS, get_criterion
from .optim import build_optimizer, warmup_decay_scheduler

log = logging.getLogger("mitune")


def pick_device(config: Optional[Dict] = None) -> torch.device:
    want = (config or {}).get("device")
    if want:
        return torch.device(want)
    return torch.device("cuda:0" if torch.cuda.is_available() else "cpu")


def _infer_input_size(loader: Iterable) -> int:
    ds = getattr(loader, "dataset", None)
    if ds is not None and hasattr(ds, "tensors"):
        return int(ds.tensors[0].shape[-1])
    x, _ = next(iter(loader))
    return int(x.shape[-1])


@torch.no_grad()
def evaluate(model: torch.nn.Module, loader: Iterable, criterion, device: torch.device) -> Dict[str, float]:
    """Sample-weighted validation loss and MAPE (the reference averages per-batch means, which over-weights a
    ragged last batch)."""
    model.eval()
    n = 0
    loss_sum = torch.zeros((), device=device, dtype=torch.float64)
    ape_sum = torch.zeros((), device=device, dtype=torch.float64)
    for x, y in loader:
        x, y = x.to(device, non_blocking=True), y.to(device, non_blocking=True)
        out = model(x).reshape(y.shape)
        loss_sum += criterion(out, y).double() * y.numel()
        ape_sum += (torch.abs(y - out) / (torch.abs(y) + MAPE_EPS)).sum().double()
        n += y.numel()
    return {"validation_loss": float(loss_sum) / max(n, 1), "validation_mape": 100.0 * float(ape_sum) / max(n, 1)}


def _fit(model: torch.nn.Module, config: Dict, data_loader, val_loader, epochs: int, optimizer, scheduler,
         criterion, device: torch.device, clip: float = 0.0, start_epoch: int = 0, report_checkpoints: bool = True):
    for epoch in range(start_epoch, epochs):
        model.train()
        running = torch.zeros((), device=device)
        batches = 0
        for x, y in data_loader:
            x, y = x.to(device, non_blocking=True), y.to(device, non_blocking=True)
            optimizer.zero_grad(set_to_none=True)
            out = model(x).reshape(y.shape)
            loss = criterion(out, y)
            loss.backward()
            if clip and clip > 0:
                torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
            optimizer.step()
            if scheduler is not None:
                scheduler.step()
            running += loss.detach()
            batches += 1
        metrics = evaluate(model, val_loader, criterion, device)
        metrics["train_loss"] = float(running) / max(batches, 1)
        log.info("epoch %d  train_loss %.5f  validation_loss %.5f", epoch + 1, metrics["train_loss"],
                 metrics["validation_loss"])
        ckpt = None
        if report_checkpoints:
            ckpt = {"model": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                    "optimizer": optimizer.state_dict(), "epoch": epoch + 1}
        if session.is_session_enabled():
            session.report(metrics, checkpoint=ckpt)
    return model


def _device_tensors(loader, device: torch.device):
    """The loader's whole dataset as device tensors, or None when it is not a plain TensorDataset."""
    ds = getattr(loader, "dataset", None)
    tensors = getattr(ds, "tensors", None)
    if tensors is None or len(tensors) != 2:
        return None
    return tensors[0].to(device, non_blocking=True), tensors[1].to(device, non_blocking=True)


def _fit_graphed(model, config: Dict, data_loader, val_loader, epochs: int, criterion, device: torch.device,
                 opt_name: str, lr: float, weight_decay: float, momentum: float = 0.9, clip: float = 0.0,
                 warmup: int = 0, total: int = 0, report_checkpoints: bool = True) -> bool:
    """GPU path: device-resident data, flat parameters, fused optimiser, hipGraph-replayed steps.

    Returns False (having done nothing) when the loaders do not wrap plain tensors; the caller then runs the
    eager loop.
    """
    from .graphed import FlatOptimizer, FlatParams, GraphedTrainer
    train_t, val_t = _device_tensors(data_loader, device), _device_tensors(val_loader, device)
    if train_t is None or val_t is None:
        return False
    flat = FlatParams(model, grad_slab=False)      # autograd's own gradient tensors, read in place by the optimiser
    opt = FlatOptimizer(flat, opt_name, lr, weight_decay, momentum, clip=clip, warmup=warmup, total=total)
    start = _resume(model, opt)
    opt.set_hyper(lr, weight_decay)           # PBT may have perturbed these since the checkpoint was taken
    bs = int(getattr(data_loader, "batch_size", None) or 32)
    signature = (type(model).__name__, getattr(criterion, "__name__", type(criterion).__name__), opt_name, bs, clip > 0,
                 tuple(train_t[0].shape[1:]), tuple(tuple(p.shape) for p in flat.params),
                 tuple(sorted((k, v if isinstance(v, (str, bool)) else v > 0) for k, v in config.items()
                              if isinstance(v, (str, bool, int, float)))))
    trainer = GraphedTrainer(model, criterion, train_t[0], train_t[1], bs, opt, shuffle=True,
                             seed=int(config.get("seed", 0)), use_graph=bool(config.get("hip_graph", True)),
                             signature=signature)
    for epoch in range(start, epochs):
        metrics = trainer.epoch_with_validation(val_t[0], val_t[1], mape_eps=MAPE_EPS)
        train_loss = metrics["train_loss"]
        metrics["graphed"] = trainer.graphed
        log.info("epoch %d  train_loss %.5f  validation_loss %.5f", epoch + 1, train_loss, metrics["validation_loss"])
        ckpt = None
        if report_checkpoints:
            ckpt = {"model": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                    "optimizer": opt.state_dict(), "epoch": epoch + 1}
        if session.is_session_enabled():
            session.report(metrics, checkpoint=ckpt)
    return True


def _resume(model, optimizer) -> int:
    ck = session.get_checkpoint()
    if not ck or "model" not in ck:
        return 0
    model.load_state_dict(ck["model"])
    try:
        if ck.get("optimizer"):
            optimizer.load_state_dict(ck["optimizer"])
    except (ValueError, KeyError, TypeError, AttributeError):
        pass       # e.g. PBT changed the optimiser family: keep fresh moments
    return int(ck.get("epoch", 0))


def _rebatch(loader, batch_size: Optional[int], shuffle: bool):
    """Honour a sampled ``batch_size`` (the reference samples one but trains every trial at 32)."""
    ds = getattr(loader, "dataset", None)
    if not batch_size or ds is None or getattr(loader, "batch_size", None) == batch_size:
        return loader
    return torch.utils.data.DataLoader(ds, batch_size=int(batch_size), shuffle=shuffle)


def train_dummy_model(config: Dict, data_loader, val_loader, epochs: int = 5) -> None:
    """Sample workflow: small stock Transformer, AdamW + MSE, 5 epochs."""
    device = pick_device(config)
    model = SimpleTransformerModel.from_config(config, _infer_input_size(data_loader)).to(device)
    n_epochs = int(config.get("epochs", epochs))
    if device.type == "cuda" and _fit_graphed(model, config, data_loader, val_loader, n_epochs, torch.nn.MSELoss(), device,
                                              "adamw", config["learning_rate"], config.get("weight_decay", 0.0)):
        return
    opt = torch.optim.AdamW(model.parameters(), lr=config["learning_rate"], weight_decay=config.get("weight_decay", 0.0))
    start = _resume(model, opt)
    for g in opt.param_groups:            # PBT may have perturbed these since the checkpoint was taken
        g["lr"], g["weight_decay"] = config["learning_rate"], config.get("weight_decay", 0.0)
    _fit(model, config, data_loader, val_loader, n_epochs, opt, None, torch.nn.MSELoss(), device, start_epoch=start)


def train_transformer_model(config: Dict, data_loader, val_loader, epochs: int = 20) -> None:
    """Full workflow: configurable Transformer, 4 optimisers x 4 losses, warm-up/decay LR, gradient clipping."""
    device = pick_device(config)
    data_loader = _rebatch(data_loader, config.get("batch_size"), shuffle=True)
    model = TransformerModel.from_config(config, _infer_input_size(data_loader)).to(device)
    n_epochs = int(config.get("epochs", epochs))
    steps_per_epoch = max(1, len(data_loader))
    warm = int(config.get("warmup_steps", 0))
    total = int(config.get("total_steps", 0))
    # the search space's step counts are sized for long runs; never let the decay end before training does
    total = max(total, n_epochs * steps_per_epoch) if total else 0
    warm = min(warm, max(1, (n_epochs * steps_per_epoch) // 4)) if warm else 0
    criterion = get_criterion(config.get("loss_function", "mse"))
    if device.type == "cuda" and _fit_graphed(
            model, config, data_loader, val_loader, n_epochs, criterion, device, config.get("optimizer", "adamw"),
            float(config["learning_rate"]), float(config.get("weight_decay", 0.0)), float(config.get("momentum", 0.9)),
            clip=float(config.get("gradient_clipping", 0.0)), warmup=warm, total=total):
        return
    opt = build_optimizer(config.get("optimizer", "adamw"), model.parameters(), float(config["learning_rate"]),
                          float(config.get("weight_decay", 0.0)), float(config.get("momentum", 0.9)))
    start = _resume(model, opt)
    sched = warmup_decay_scheduler(opt, warm, total) if (warm or total) else None
    if sched is not None:
        for _ in range(start * steps_per_epoch):
            sched.step()
    _fit(model, config, data_loader, val_loader, n_epochs, opt, sched, criterion, device,
         clip=float(config.get("gradient_clipping", 0.0)), start_epoch=start)
