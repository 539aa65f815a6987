#!/usr/bin/env python
"""Summarise a rocprofv3 kernel-trace CSV per training epoch: per-kernel mean duration, busy time and wall span.

    python bench/trace_summary.py <kernel_trace.csv> [epochs_to_print ...]

Epochs are separated by the host synchronisation at the end of each epoch (a gap > 1 ms without any
mt_* kernel).  Output is a small text table meant to be committed under profiles/.
"""
import csv
import os
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    want = [int(a) for a in sys.argv[2:]]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if "mt_" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            if os.environ.get("TRACE_BY_GRID"):       # one line per launch shape: the layers of a pack differ in grid size
                grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
                wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or "256"
                try:
                    short += f" [{int(grid) // max(int(wg), 1)} wg]"
                except ValueError:
                    short += f" [{grid}]"
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
    rows.sort()
    epochs, cur, last_end = [], [], None
    for s, e, n in rows:
        if last_end is not None and s - last_end > 1_000_000 and any("gather" in x[2] for x in cur):
            epochs.append(cur)
            cur = []
        cur.append((s, e, n))
        last_end = max(last_end or 0, e)
    if cur:
        epochs.append(cur)
    train = [ep for ep in epochs if sum("gather" in x[2] for x in ep) > 10]
    print(f"{len(train)} training epochs found")
    for i, ep in enumerate(train):
        if want and i not in want:
            continue
        span = (max(x[1] for x in ep) - min(x[0] for x in ep)) / 1e6
        busy = sum(e - s for s, e, _ in ep) / 1e6
        steps = sum("gather" in x[2] for x in ep)
        print(f"--- epoch {i}: {steps} steps, span {span:.1f} ms, kernel-busy {busy:.1f} ms, "
              f"{span / steps * 1e3:.1f} us/step")
        agg = defaultdict(lambda: [0, 0])
        for s, e, n in ep:
            agg[n][0] += 1
            agg[n][1] += e - s
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"    {n:<46} calls {c:>6}  mean {t / c / 1e3:>8.2f} us  total {t / 1e6:>8.1f} ms")


if __name__ == "__main__":
    main()
