"""``mitune.tune`` -- a ``ray.tune``-shaped namespace (run / report / with_parameters / domains)."""
from .space import (choice, uniform, loguniform, quniform, qloguniform, randn, randint, lograndint,  # noqa: F401
                    sample_from, grid_search, Domain, Categorical, Float, Integer, Function)
from .session import report, get_checkpoint, get_trial_id, get_trial_dir, is_session_enabled  # noqa: F401
from .trial import Trial, PENDING, RUNNING, PAUSED, TERMINATED, ERROR   # noqa: F401
from .analysis import ExperimentAnalysis   # noqa: F401
from .logger import JsonLogger, CSVLogger, MemoryLogger, Logger   # noqa: F401
from .checkpoint import save_checkpoint, load_checkpoint, latest_checkpoint, list_checkpoints  # noqa: F401


def __getattr__(name):
    # run()/with_parameters() import the scheduler and search packages, which import this
    # package for Trial/space: resolve them lazily to keep the import graph acyclic.
    if name in ("run", "with_parameters", "TuneError", "ProgressReporter"):
        from . import api as _run
        return getattr(_run, name)
    if name in ("schedulers", "search", "suggest"):
        import importlib
        return importlib.import_module("mitune." + ("search" if name == "suggest" else name))
    raise AttributeError(name)
