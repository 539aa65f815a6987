"""Workflow logging: one timestamped log file per run, same line format as the reference.

The reference configures ``logging.basicConfig`` at import time with a file under a hard-coded home
directory (/root/reference/ray-tune-hpo-regression-sample.py:16-23), which makes the script unimportable on any other
machine.  ``setup_logging`` is explicit, takes the directory as an argument and never raises because a
directory is missing.
"""
from __future__ import annotations

import logging
import os
import time
from typing import Optional

FORMAT = "%(asctime)s - %(levelname)s - %(message)s"


def setup_logging(log_dir: Optional[str] = None, prefix: str = "mitune_workflow", level: int = logging.INFO,
                  also_stderr: bool = False) -> logging.Logger:
    logger = logging.getLogger("mitune")
    logger.setLevel(level)
    for h in list(logger.handlers):
        logger.removeHandler(h)
    if log_dir:
        os.makedirs(log_dir, exist_ok=True)
        path = os.path.join(log_dir, f"{prefix}_{time.strftime('%Y_%m_%d_%H_%M_%S')}.log")
        fh = logging.FileHandler(path)
        fh.setFormatter(logging.Formatter(FORMAT))
        logger.addHandler(fh)
        logger.log_path = path          # type: ignore[attr-defined]
    if also_stderr or not log_dir:
        sh = logging.StreamHandler()
        sh.setFormatter(logging.Formatter(FORMAT))
        logger.addHandler(sh)
    logger.propagate = False
    return logger


def get_logger() -> logging.Logger:
    return logging.getLogger("mitune")
