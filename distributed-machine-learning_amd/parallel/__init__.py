"""Multi-GPU layer: collectives (RCCL / gloo / in-process), state exchange, data-parallel trials."""
from .comm import Comm, LocalComm, TorchDistComm, ThreadComm, default_comm   # noqa: F401
