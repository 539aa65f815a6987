"""Search-algorithm interface, random/grid variant generation and a concurrency limiter.

With no ``search_alg`` the reference's ``tune.run`` draws ``num_samples`` independent random
configs from the search space (/root/reference/ray-tune-hpo-regression-sample.py:159-167);
``BasicVariantGenerator`` is that behaviour.
"""
from __future__ import annotations

from typing import Any, Dict, Iterator, List, Optional

import numpy as np

from ..tune import space as sp


class Searcher:
    FINISHED = "FINISHED"
    lookahead_safe = False       # True when suggestions do not depend on reported results

    def __init__(self, metric: Optional[str] = None, mode: Optional[str] = None):
        self.metric, self.mode = metric, mode
        self._space: Optional[Dict] = None

    def set_search_properties(self, metric: Optional[str], mode: Optional[str], config: Dict) -> bool:
        self.metric = self.metric or metric
        self.mode = self.mode or mode
        self._space = config
        return True

    def suggest(self, trial_id: str) -> Any:
        """A config dict, ``None`` (nothing to suggest right now) or ``Searcher.FINISHED``."""
        raise NotImplementedError

    def on_trial_result(self, trial_id: str, result: Dict) -> None:
        pass

    def on_trial_complete(self, trial_id: str, result: Optional[Dict] = None, error: bool = False) -> None:
        pass

    def state_dict(self) -> Dict:
        return {}

    def load_state_dict(self, state: Dict) -> None:
        pass


class BasicVariantGenerator(Searcher):
    """Random search with optional grid axes; deterministic for a given seed."""

    lookahead_safe = True        # suggestions never depend on results: trials may be drawn ahead of their start

    def __init__(self, num_samples: int = 1, seed: Optional[int] = 0, points_to_evaluate: Optional[List[Dict]] = None):
        super().__init__()
        self.num_samples, self.seed = num_samples, seed
        self._points = list(points_to_evaluate or [])
        self._iter: Optional[Iterator[Dict]] = None
        self._served = 0

    def set_search_properties(self, metric, mode, config) -> bool:
        super().set_search_properties(metric, mode, config)
        self._iter = sp.generate_variants(config, self.num_samples, self.seed)
        return True

    @property
    def total(self) -> int:
        return len(self._points) + sp.count_variants(self._space or {}, self.num_samples)

    def suggest(self, trial_id: str):
        if self._points:
            self._served += 1
            cfg = dict(self._space or {})
            cfg.update(self._points.pop(0))
            return sp.resolve(cfg, np.random.default_rng(self.seed))
        assert self._iter is not None, "set_search_properties() must be called first"
        try:
            cfg = next(self._iter)
        except StopIteration:
            return Searcher.FINISHED
        self._served += 1
        return cfg

    def state_dict(self) -> Dict:
        return {"served": self._served}

    def load_state_dict(self, state: Dict) -> None:
        # replay the deterministic stream up to where the previous run stopped
        n = int(state.get("served", 0))
        for _ in range(n - self._served):
            if self.suggest("_replay") is Searcher.FINISHED:
                break


class ConcurrencyLimiter(Searcher):
    """Cap the number of suggestions whose outcome the wrapped searcher has not seen yet."""

    def __init__(self, searcher: Searcher, max_concurrent: int):
        super().__init__(searcher.metric, searcher.mode)
        self.searcher, self.max_concurrent = searcher, max_concurrent
        self._live: set = set()

    def set_search_properties(self, metric, mode, config) -> bool:
        return self.searcher.set_search_properties(metric, mode, config)

    def suggest(self, trial_id: str):
        if len(self._live) >= self.max_concurrent:
            return None
        cfg = self.searcher.suggest(trial_id)
        if isinstance(cfg, dict):
            self._live.add(trial_id)
        return cfg

    def on_trial_result(self, trial_id, result):
        self.searcher.on_trial_result(trial_id, result)

    def on_trial_complete(self, trial_id, result=None, error=False):
        self._live.discard(trial_id)
        self.searcher.on_trial_complete(trial_id, result, error)

    def state_dict(self):
        return self.searcher.state_dict()

    def load_state_dict(self, state):
        self.searcher.load_state_dict(state)
