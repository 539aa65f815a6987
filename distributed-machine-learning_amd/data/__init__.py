"""Datasets: synthetic regression tables, windowing, feature lists, loaders."""
