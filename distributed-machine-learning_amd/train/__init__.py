"""Single-trial training utilities: losses, optimisers, LR schedule, reference-style train functions."""
from .losses import mape_loss, get_criterion, elementwise_loss, LOSS_CODES   # noqa: F401
from .optim import build_optimizer, optimizer_mapping, warmup_decay_scheduler, OPT_CODES   # noqa: F401
