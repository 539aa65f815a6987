"""Model families: MLP regressor, Transformer regressors, FT-Transformer."""
from .mlp import MLPRegressor, MLPTrialSpec, init_mlp_state, lr_factor   # noqa: F401
