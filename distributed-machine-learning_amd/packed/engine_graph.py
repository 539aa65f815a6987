"""GraphPackEngine: non-MLP trial families (FT-Transformer, custom factories) on the GPU, one captured hipGraph
per slot.

The grouped-GEMM engine packs MLP trials into shared launches; Transformer-style trials differ in depth, width and
head count, so they keep one module per slot -- but nothing else of the eager loop survives: parameters of a slot
live in one flat slab, the optimiser is the fused device-state kernel, and a slot's whole training step
(gather -> forward in bf16 autocast with the fused attention / LayerNorm kernels -> loss -> backward -> update) is
captured once and replayed for every batch of every epoch (``train/graphed.py``).  Row order stays the keyed
Feistel permutation of ``(seed, trial_index, epoch)``, so a trial's data order does not depend on its slot or GPU.

PBT exploits inside a pack are slab-to-slab copies (weights + both moment slabs + step counter); the captured graph
of the receiving slot stays valid because it only holds pointers.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .engine import TorchPackEngine, _Slot
from .permutation import permute, round_keys
from ..models.mlp import MLPTrialSpec, init_mlp_state
from ..ops import rng
from ..train.graphed import FlatOptimizer, FlatParams, GraphedTrainer
from ..train.losses import get_criterion


class GraphPackEngine(TorchPackEngine):
    name = "graph"

    def __init__(self, trainable, device="cuda", capacity: Optional[int] = None, seed: int = 0):
        super().__init__(trainable, device, capacity, seed)
        if self.device.type != "cuda":
            raise ValueError("GraphPackEngine needs a GPU; use engine='torch' on the CPU")
        self._trainers: Dict[int, GraphedTrainer] = {}

    # ------------------------------------------------------------------ slot management
    def load(self, slot, trial_index, config, state=None):
        if slot in self.slots:
            raise RuntimeError(f"slot {slot} is occupied")
        spec = MLPTrialSpec.from_config(config, self.t.in_features)
        s = _Slot()
        s.trial_index, s.spec, s.config = int(trial_index), spec, dict(config)
        model, _ = self._build(spec, config, trial_index)
        if state is None and self.t.model == "mlp":
            model.load_state_dict(init_mlp_state(spec, self.seed, trial_index))
        if state is not None and "model" in state:
            model.load_state_dict(state["model"])
        flat = FlatParams(model, grad_slab=False)
        s.model = model
        s.opt = FlatOptimizer(flat, spec.optimizer, spec.learning_rate, spec.weight_decay, spec.momentum,
                              (spec.beta1, spec.beta2), spec.eps, spec.gradient_clipping, spec.warmup_steps,
                              spec.total_steps)
        s.epoch, s.step, s.train_loss = 0, 0, float("nan")
        if state is not None:
            if state.get("optimizer"):
                try:
                    s.opt.load_state_dict(state["optimizer"])
                except (ValueError, KeyError, TypeError):
                    pass       # another optimiser family or layout: fresh moments
            s.epoch, s.step = int(state.get("epoch", 0)), int(state.get("step", 0))
        self.slots[slot] = s
        self._trainers[slot] = GraphedTrainer(model, get_criterion(spec.loss_function), self.Xtr, self.ytr,
                                              spec.batch_size, s.opt, shuffle=False, drop_last=True,
                                              forward_ctx=self._autocast)

    def evict(self, slot):
        self.slots.pop(slot, None)
        self._trainers.pop(slot, None)

    def export(self, slot):
        s = self.slots[slot]
        return {"model": {k: v.detach().cpu().clone() for k, v in s.model.state_dict().items()},
                "optimizer": s.opt.state_dict(), "epoch": s.epoch, "step": s.step, "trial_index": s.trial_index}

    def clone(self, src, dst, config):
        a, b = self.slots[src], self.slots[dst]
        if a.opt.flat.n != b.opt.flat.n or a.opt.name != MLPTrialSpec.from_config(config, self.t.in_features).optimizer:
            return super().clone(src, dst, config)       # shapes differ: go through a state_dict
        with torch.no_grad():
            b.opt.flat.P.copy_(a.opt.flat.P)
            b.opt.M.copy_(a.opt.M)
            b.opt.V.copy_(a.opt.V)
            b.opt._i32[b.opt._STEP:b.opt._STEP + 1].copy_(a.opt._i32[a.opt._STEP:a.opt._STEP + 1])
            for (_, bb), (_, ab) in zip(b.model.named_buffers(), a.model.named_buffers()):
                bb.copy_(ab)
        b.spec, b.config = MLPTrialSpec.from_config(config, self.t.in_features), dict(config)
        b.epoch, b.step = a.epoch, a.step
        self._apply_hparams(b)

    @staticmethod
    def _apply_hparams(s: _Slot) -> None:
        s.opt.set_hyper(s.spec.learning_rate, s.spec.weight_decay)

    # ------------------------------------------------------------------ training
    def _train_slot_epoch(self, s: _Slot, slot: Optional[int] = None) -> None:
        tr = self._trainers[slot]
        n = self.Xtr.shape[0]
        steps = n // tr.B
        keys = round_keys(self.seed, s.trial_index, s.epoch)
        perm = permute(np.arange(steps * tr.B, dtype=np.uint32), n, keys).astype(np.int64)
        tr.perm[:steps * tr.B].copy_(torch.from_numpy(perm))      # blocking: the host array dies with this frame
        # dropout draws of this epoch depend on (seed, trial, epoch) only, never on the slot or on neighbours
        rng.seed_all(((self.seed * 1_000_003 + s.trial_index) * 4099 + s.epoch) * 0x2545F491 + 77, self.device)
        s.train_loss = tr.train_epoch()
        s.step += steps
        s.epoch += 1

    def train_epoch(self):
        for slot in self.occupied():
            self._train_slot_epoch(self.slots[slot], slot)
