"""mitune -- MI355X-native hyper-parameter optimisation.

The sources live in ``distributed-machine-learning_amd/`` (a directory name Python cannot import
directly); this shim puts that directory on the package path so that ``mitune.tune``,
``mitune.schedulers``, ``mitune.ops`` ... resolve to it, and re-exports the user-facing API:

    from mitune import tune
    analysis = tune.run(tune.with_parameters(train_fn, data=...), config={...},
                        scheduler=tune.schedulers.ASHAScheduler(metric="validation_loss", mode="min"))
"""
from pathlib import Path as _Path

_SRC = _Path(__file__).resolve().parent.parent / "distributed-machine-learning_amd"
__path__.append(str(_SRC))          # noqa: F821  (module attribute provided by the import system)

__version__ = "0.1.0"

from .tune.space import (choice, uniform, loguniform, quniform, qloguniform, randn, randint,     # noqa: E402,F401
                         lograndint, sample_from, grid_search)
from .tune.session import report, get_checkpoint, get_trial_id, get_trial_dir          # noqa: E402,F401
from .tune.api import run, with_parameters, TuneError                                   # noqa: E402,F401
from .tune.analysis import ExperimentAnalysis                                            # noqa: E402,F401
from . import tune, schedulers, search                                                   # noqa: E402,F401
