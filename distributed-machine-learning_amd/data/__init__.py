"""Datasets: synthetic regression tables, windowing, feature lists, loaders."""
from .windows import split_into_intervals, window_labels                     # noqa: F401
from .synthetic import create_dummy_data, tabular_regression, sequence_regression_arrays   # noqa: F401
from .loaders import get_data_loaders, load_dataframe_from_npy, frames_to_loaders, synthetic_glucose_frame  # noqa: F401
from . import features                                                       # noqa: F401
