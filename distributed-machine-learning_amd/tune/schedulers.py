"""``mitune.tune.schedulers`` -- same import path shape as the reference's ``ray.tune.schedulers``
(/root/reference/ray-tune-hpo-regression.py:8): ``from mitune.tune.schedulers import ASHAScheduler``."""
from ..schedulers import (ASHAScheduler, AsyncHyperBandScheduler, FIFOScheduler, MedianStoppingRule,   # noqa: F401
                          PopulationBasedTraining, TrialScheduler)
