#!/usr/bin/env python
"""Summarise rocprofv3 counter-collection CSVs: mean counter value per dispatch, per kernel and grid size.

    python bench/pmc_summary.py <dir with *counter_collection.csv> [> profiles/pmc_<pass>.txt]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for path in files:
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
                if "mt_" not in name:
                    continue
                key = (short(name), r.get("Grid_Size", "?"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for v in acc.values() for c in v})
    print("kernel | grid | lds | dispatches | " + " | ".join(counters))
    for key in sorted(acc):
        v = acc[key]
        n = max(len(x) for x in v.values())
        print(" | ".join([key[0], str(key[1]), str(key[2]), str(n)] +
                         [f"{sum(v[c]) / len(v[c]):.4g}" if c in v else "-" for c in counters]))


if __name__ == "__main__":
    main()
