"""Search algorithms: random/grid variant generation, GP-UCB Bayesian optimisation."""
from .searcher import Searcher, BasicVariantGenerator, ConcurrencyLimiter   # noqa: F401
from .bayesopt import BayesOptSearch, GaussianProcess, SpaceEncoder          # noqa: F401
