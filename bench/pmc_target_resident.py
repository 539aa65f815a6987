#!/usr/bin/env python
"""Counter-collection target: two resident epochs of the 16-survivor pack (bench/resident_micro.py, resident path only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0], "--modes", "resident", "--rows", "60000", "--epochs", "1"]
import resident_micro  # noqa: E402

resident_micro.main()
