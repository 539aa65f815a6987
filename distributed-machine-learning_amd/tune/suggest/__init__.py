"""``mitune.tune.suggest`` -- counterpart of ``ray.tune.suggest`` (/root/reference/ray-tune-hpo-regression.py:9)."""
from ...search import BasicVariantGenerator, BayesOptSearch, ConcurrencyLimiter, Searcher   # noqa: F401
