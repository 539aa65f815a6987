"""Utilities: logging, timers, seeding, HBM budgeting, profiling helpers."""
from .logging import setup_logging, get_logger            # noqa: F401
from .timers import PhaseTimer, DeviceTimer               # noqa: F401
from .hbm import default_capacity, mlp_pack_footprint     # noqa: F401
from .seeding import seed_everything, derive_seed         # noqa: F401
