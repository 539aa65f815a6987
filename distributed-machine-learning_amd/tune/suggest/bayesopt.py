"""``from mitune.tune.suggest.bayesopt import BayesOptSearch`` (/root/reference/ray-tune-hpo-regression.py:9)."""
from ...search.bayesopt import BayesOptSearch, GaussianProcess, SpaceEncoder   # noqa: F401
