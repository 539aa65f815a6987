"""Wearable-sensor feature-column names for the glucose-forecasting workload.

The reference keeps these as nine hand-written lists in ``config.py``
(/root/reference/config.py:1-78) and assembles 81 model inputs from them
(/root/reference/ray-tune-hpo-regression.py:13-19).  Here the same names are *generated* from the
naming rule (signal x rolling window x statistic), which also documents the rule's one irregularity:
heart-rate rolling columns carry no ``min`` unit suffix while the other three signals do.
"""
from __future__ import annotations

from typing import Dict, List

ROLLING_WINDOWS_MIN = (15, 30, 60, 90, 180, 240, 360, 720, 1440)
STATISTICS = ("Mean", "Std")
SIGNALS = ("HeartRate", "Sleep", "Intensity", "Steps")
_NO_UNIT_SUFFIX = {"HeartRate"}
LABEL_COLUMN = "Historic Glucose mg/dL"


def raw_feature(signal: str) -> List[str]:
    return [signal]


def rolling_features(signal: str) -> List[str]:
    unit = "" if signal in _NO_UNIT_SUFFIX else "min"
    return [f"{signal}_{w}{unit}_{stat}" for w in ROLLING_WINDOWS_MIN for stat in STATISTICS]


heart_rate_features_1, heart_rate_features_2 = raw_feature("HeartRate"), rolling_features("HeartRate")
sleep_features_1, sleep_features_2 = raw_feature("Sleep"), rolling_features("Sleep")
intensity_features_1, intensity_features_2 = raw_feature("Intensity"), rolling_features("Intensity")
steps_features_1, steps_features_2 = raw_feature("Steps"), rolling_features("Steps")
temporal_features = [f"{name}_{fn}" for name in ("MinuteOfDay", "DayOfWeek") for fn in ("Sin", "Cos")] + ["Is_Weekend"]

# 4 raw signals + 5 temporal columns, then the 4 x 18 rolling statistics: 81 unique columns
features_1 = heart_rate_features_1 + sleep_features_1 + intensity_features_1 + steps_features_1 + temporal_features
features = features_1 + heart_rate_features_2 + sleep_features_2 + intensity_features_2 + steps_features_2


def feature_groups() -> Dict[str, List[str]]:
    return {"heart_rate_features_1": heart_rate_features_1, "heart_rate_features_2": heart_rate_features_2,
            "sleep_features_1": sleep_features_1, "sleep_features_2": sleep_features_2,
            "intensity_features_1": intensity_features_1, "intensity_features_2": intensity_features_2,
            "steps_features_1": steps_features_1, "steps_features_2": steps_features_2,
            "temporal_features": temporal_features}
