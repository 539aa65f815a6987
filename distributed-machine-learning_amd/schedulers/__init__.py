"""Trial schedulers: FIFO, ASHA (async + collective/lock-step form), median stopping, PBT."""
from .base import TrialScheduler, FIFOScheduler, CONTINUE, PAUSE, STOP, RESTART   # noqa: F401
from .asha import AsyncHyperBandScheduler, ASHAScheduler, MedianStoppingRule      # noqa: F401
from .pbt import PopulationBasedTraining, explore                                  # noqa: F401
