#!/usr/bin/env python
"""Counter-collection target: one short epoch of the 16-survivor exact-f32 pack on the row-block step
(bench/small_pack_micro.py, 20 000 rows = 62 steps per epoch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0], "--forms", os.environ.get("PMC_FORM", "1"), "--rows", "20000", "--epochs", "1"]
import small_pack_micro  # noqa: E402

small_pack_micro.main()
