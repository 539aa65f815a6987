"""Packed multi-trial training: many trials per GPU as slots of one batched model."""
from .trainable import PackedTrainable            # noqa: F401
from .engine import PackEngine, TorchPackEngine   # noqa: F401
from .executor import PackedExecutor, RankLostError   # noqa: F401
from .permutation import epoch_permutation, permute, round_keys   # noqa: F401
