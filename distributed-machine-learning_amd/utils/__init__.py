"""Utilities: logging, timers, seeding, HBM budgeting, profiling helpers."""
