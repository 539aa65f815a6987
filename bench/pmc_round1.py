#!/usr/bin/env python
"""MFMA utilisation of the launches of a search's first round, from a rocprofv3 counter run of ``bench.py --steps 1``:

    cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv \
        -d /tmp/pmc_r1 -- python3 <repo>/bench.py --steps 1 --warmup 0 --no-bf16
    python bench/pmc_round1.py /tmp/pmc_r1 > profiles/...

Utilisation of a kernel = sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), over all its
dispatches (a dispatch that runs longer weighs more: this is the share of the kernel's SIMD-cycles with the matrix pipe
busy).
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r.get("Kernel_Name") or ""
                if "mt_" not in name:
                    continue
                key = re.sub(r"^void ", "", re.sub(r"\(.*$", "", name))[:60]
                acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[key].add(r.get("Dispatch_Id"))
    rows = []
    for k, v in acc.items():
        cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        rows.append((cyc, k, len(calls[k]), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print(f"{'kernel':62s} {'calls':>7s} {'time share':>10s} {'ms @2.4GHz':>10s} {'MFMA util':>9s}")
    gc = gm = 0.0
    for cyc, k, n, mf in rows:
        util = mf / (cyc * 1024.0) if cyc else 0.0
        print(f"{k:62s} {n:7d} {100 * cyc / total:9.1f}% {cyc / 2.4e6:10.1f} {100 * util:8.1f}%")
        if "gemm" in k:
            gc += cyc
            gm += mf
    print(f"all GEMM launches: {100 * gc / total:.1f} % of kernel cycles, MFMA utilisation {100 * gm / (gc * 1024):.1f} %")
    allm = sum(r[3] for r in rows)
    print(f"all kernels: MFMA utilisation {100 * allm / (total * 1024):.1f} %")


if __name__ == "__main__":
    main()
