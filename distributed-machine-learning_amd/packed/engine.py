"""Pack engines: the device-side trainer behind the packed executor.

``PackEngine`` is the small interface the executor drives (load / evict / train_epoch / evaluate /
export / clone).  ``TorchPackEngine`` implements it with one ordinary ``MLPRegressor`` +
``torch.optim`` optimiser per slot: it is the CPU path (the "plumbing" configuration: a few
trials on a 20k x 8 table) and the oracle the HIP engine is tested against -- "a pack of T trials
equals T independent single-trial trainings" is checked against *this* code.

What both engines guarantee, so that a trial's outcome never depends on where it ran:
initial weights come from ``init_mlp_state(seed, trial_index)``; the row order of epoch ``e`` is the
keyed Feistel permutation ``(seed, trial_index, e)``; an epoch is ``N_train // batch_size`` steps
(the ragged tail batch is dropped); validation metrics are sample-weighted means over the whole
validation set.  The training loop has the same steps as the reference's
(/root/reference/ray-tune-hpo-regression.py:322-370): zero_grad, forward, loss, backward,
optional global-norm clip, optimiser step, LR schedule; but the loss is accumulated on the device
and read once per epoch instead of ``.item()`` per batch.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from .permutation import permute, round_keys
from .trainable import PackedTrainable
from ..models.mlp import MLPRegressor, MLPTrialSpec, init_mlp_state, lr_factor
from ..train.losses import MAPE_EPS, get_criterion
from ..train.optim import build_optimizer


MODEL_FAMILIES = ("mlp", "ft_transformer", "transformer", "simple_transformer")


class PackEngine:
    """Interface; slots are integers in ``[0, capacity)``."""

    capacity: int
    name = "base"
    move_cost_s = 0.0        # what bringing a trial up from its exported state costs (seconds); weighed against the round
                             # time a move to an emptier GPU saves

    def load(self, slot: int, trial_index: int, config: Dict, state: Optional[Dict] = None) -> None:
        raise NotImplementedError

    def trial_cost(self, config: Dict) -> float:
        """What one round of a trial with this configuration weighs on its GPU, in any unit that adds up over the trials
        of a rank (the executor spreads trials over the ranks by it: a round lasts as long as its most loaded GPU).  A
        function of the configuration alone, so every rank computes the same number.  1.0 = every trial weighs the same."""
        return 1.0

    def evict(self, slot: int) -> None:
        raise NotImplementedError

    def occupied(self) -> List[int]:
        raise NotImplementedError

    def train_epoch(self) -> None:
        raise NotImplementedError

    def evaluate(self) -> Dict[int, Dict[str, float]]:
        raise NotImplementedError

    def export(self, slot: int) -> Dict[str, Any]:
        raise NotImplementedError

    def clone(self, src: int, dst: int, config: Dict) -> None:
        """PBT exploit + explore inside the pack: ``dst`` takes ``src``'s weights and optimiser state."""
        raise NotImplementedError

    def import_state(self, slot: int, state: Dict, config: Dict) -> None:
        """Overwrite a loaded slot's weights/optimiser state (cross-GPU PBT exploit)."""
        raise NotImplementedError

    def close(self) -> None:
        pass


class _Slot:
    __slots__ = ("trial_index", "spec", "model", "opt", "epoch", "step", "train_loss", "config")


class TorchPackEngine(PackEngine):
    name = "torch"

    def __init__(self, trainable: PackedTrainable, device: str = "cpu", capacity: Optional[int] = None,
                 seed: int = 0):
        if not (trainable.model in MODEL_FAMILIES or callable(trainable.model)):
            raise ValueError(f"model must be one of {MODEL_FAMILIES} or a factory fn(config, in_features) -> nn.Module")
        self.t = trainable
        self.device = torch.device(device)
        self.capacity = int(capacity or trainable.capacity or 64)
        self.seed = int(seed or 0)
        self.Xtr = trainable.train[0].to(self.device, torch.float32)
        self.ytr = trainable.train[1].to(self.device, torch.float32)
        self.Xva = trainable.val[0].to(self.device, torch.float32)
        self.yva = trainable.val[1].to(self.device, torch.float32)
        self.slots: Dict[int, _Slot] = {}
        # bf16 autocast for the non-MLP families on a GPU: projections run as bf16 library GEMMs, attention
        # and LayerNorm as the fused HIP kernels
        self.amp = self.device.type == "cuda" and trainable.dtype == "bf16" and trainable.model != "mlp"

    def _autocast(self):
        return torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.amp)

    # ------------------------------------------------------------------ slot management
    def occupied(self) -> List[int]:
        return sorted(self.slots)

    def _spec(self, config: Dict) -> MLPTrialSpec:
        """Optimiser/batch/loss settings of a trial; ``model_kwargs`` supplies what the search space leaves fixed."""
        spec = MLPTrialSpec.from_config({**self.t.model_kwargs, **config}, self.t.in_features)
        if self.t.model != "mlp" and (spec.warmup_steps or spec.total_steps):
            # The full search space samples step counts sized for long runs; as in the function trainable
            # (train/functions.py::train_transformer_model) the decay never ends before training does and the
            # warm-up takes at most a quarter of it.
            n = int(self.Xtr.shape[0])
            planned = self.t.max_epochs * max(1, n // max(1, min(spec.batch_size, n)))
            spec = dataclasses.replace(
                spec, total_steps=max(spec.total_steps, planned) if spec.total_steps else 0,
                warmup_steps=min(spec.warmup_steps, max(1, planned // 4)) if spec.warmup_steps else 0)
        return spec

    def _build(self, spec: MLPTrialSpec, config: Dict, trial_index: int = 0, with_optimizer: bool = True):
        if self.t.model == "mlp":
            model = MLPRegressor(spec.in_features, spec.hidden_sizes)
        else:
            # weights drawn from the trial's own stream, so a trial's start does not depend on its slot
            with torch.random.fork_rng(devices=[]):
                torch.manual_seed((self.seed * 1_000_003 + int(trial_index)) * 16 + 7)
                merged = {**self.t.model_kwargs, **config}
                if self.t.model == "ft_transformer":
                    from ..models.ft_transformer import FTTransformer
                    model = FTTransformer.from_config(merged, spec.in_features)
                elif self.t.model == "transformer":         # the reference's full sequence regressor
                    from ..models.transformer import TransformerModel
                    model = TransformerModel.from_config(merged, spec.in_features)
                elif self.t.model == "simple_transformer":  # the reference's sample model
                    from ..models.transformer import SimpleTransformerModel
                    model = SimpleTransformerModel.from_config(merged, spec.in_features)
                else:
                    model = self.t.model(config, spec.in_features)
        if not with_optimizer:
            return model, None              # still on the CPU: the flat parameter slab uploads it in one copy
        model = model.to(self.device)
        opt = build_optimizer(spec.optimizer, model.parameters(), spec.learning_rate, spec.weight_decay,
                              spec.momentum, spec.beta1, spec.beta2, spec.eps)
        return model, opt

    def load(self, slot, trial_index, config, state=None):
        if slot in self.slots:
            raise RuntimeError(f"slot {slot} is occupied")
        spec = self._spec(config)
        s = _Slot()
        s.trial_index, s.spec, s.config = int(trial_index), spec, dict(config)
        s.model, s.opt = self._build(spec, config, trial_index)
        s.epoch, s.step, s.train_loss = 0, 0, float("nan")
        if state is not None and "model" in state:
            s.model.load_state_dict(state["model"])
            if "optimizer" in state and state["optimizer"] is not None:
                try:
                    s.opt.load_state_dict(state["optimizer"])
                except (ValueError, KeyError):
                    pass      # optimiser family changed (PBT explore): keep fresh moments
            s.epoch = int(state.get("epoch", 0))
            s.step = int(state.get("step", 0))
        elif self.t.model == "mlp":
            s.model.load_state_dict(init_mlp_state(spec, self.seed, trial_index))
        self.slots[slot] = s

    def evict(self, slot):
        self.slots.pop(slot, None)

    def export(self, slot):
        s = self.slots[slot]
        return {"model": {k: v.detach().cpu().clone() for k, v in s.model.state_dict().items()},
                "optimizer": s.opt.state_dict(), "epoch": s.epoch, "step": s.step,
                "trial_index": s.trial_index}

    def clone(self, src, dst, config):
        state = self.export(src)
        keep = self.slots[dst].trial_index
        self.evict(dst)
        self.load(dst, keep, config, state=state)
        self._apply_hparams(self.slots[dst])

    def import_state(self, slot, state, config):
        keep = self.slots[slot].trial_index
        self.evict(slot)
        self.load(slot, keep, config, state=state)
        self._apply_hparams(self.slots[slot])

    @staticmethod
    def _apply_hparams(s: _Slot) -> None:
        for g in s.opt.param_groups:
            g["lr"] = s.spec.learning_rate
            g["weight_decay"] = s.spec.weight_decay
            if "betas" in g:
                g["betas"] = (s.spec.beta1, s.spec.beta2)
            if "momentum" in g:
                g["momentum"] = s.spec.momentum

    # ------------------------------------------------------------------ training
    def _train_slot_epoch(self, s: _Slot) -> None:
        n = self.Xtr.shape[0]
        b = min(s.spec.batch_size, n)
        steps = n // b
        keys = round_keys(self.seed, s.trial_index, s.epoch)
        perm = torch.from_numpy(permute(np.arange(steps * b, dtype=np.uint32), n, keys)).to(self.device)
        crit = get_criterion(s.spec.loss_function)
        loss_sum = torch.zeros((), device=self.device)
        s.model.train()
        for i in range(steps):
            idx = perm[i * b:(i + 1) * b]
            for g in s.opt.param_groups:
                g["lr"] = s.spec.learning_rate * lr_factor(s.step, s.spec.warmup_steps, s.spec.total_steps)
            s.opt.zero_grad(set_to_none=True)
            with self._autocast():
                out = s.model(self.Xtr[idx])
            loss = crit(out.float().reshape(idx.shape[0]), self.ytr[idx])
            loss.backward()
            if s.spec.gradient_clipping > 0:
                torch.nn.utils.clip_grad_norm_(s.model.parameters(), s.spec.gradient_clipping)
            s.opt.step()
            s.step += 1
            loss_sum += loss.detach()
        s.train_loss = float(loss_sum) / max(steps, 1)
        s.epoch += 1

    def _cpu_threads(self, s: _Slot):
        """On the CPU, steps of a small model run on one intra-op thread: below a few MFLOP per step the thread pool's
        wake-ups cost more than the arithmetic (measured on an 8-vCPU box: 16 small MLP trials 123 s with eight threads,
        10.5 s with one).  Returns a context manager."""
        import contextlib
        if self.device.type != "cpu":
            return contextlib.nullcontext()
        n_params = sum(p.numel() for p in s.model.parameters())
        if 6.0 * s.spec.batch_size * n_params > 2e8:          # forward + backward FLOP of one step
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def one_thread():
            before = torch.get_num_threads()
            torch.set_num_threads(1)
            try:
                yield
            finally:
                torch.set_num_threads(before)
        return one_thread()

    def train_epoch(self):
        for slot in self.occupied():
            with self._cpu_threads(self.slots[slot]):
                self._train_slot_epoch(self.slots[slot])

    @torch.no_grad()
    def evaluate(self):
        out: Dict[int, Dict[str, float]] = {}
        n = self.Xva.shape[0]
        for slot in self.occupied():
            s = self.slots[slot]
            s.model.eval()
            crit = s.spec.loss_function
            se = torch.zeros((), device=self.device, dtype=torch.float64)
            ape = torch.zeros((), device=self.device, dtype=torch.float64)
            ls = torch.zeros((), device=self.device, dtype=torch.float64)
            from ..train.losses import elementwise_loss
            for c0 in range(0, n, self.t.val_chunk):
                x, y = self.Xva[c0:c0 + self.t.val_chunk], self.yva[c0:c0 + self.t.val_chunk]
                with self._autocast():
                    o = s.model(x)
                o = o.float().reshape(y.shape)
                se += ((o - y) ** 2).sum().double()
                ape += ((o - y).abs() / (y.abs() + MAPE_EPS)).sum().double()
                ls += elementwise_loss(crit, o, y).sum().double()
            out[slot] = {"train_loss": s.train_loss,
                         "validation_loss": float(ls) / n,
                         "validation_mse": float(se) / n,
                         "validation_mape": 100.0 * float(ape) / n}
        return out
