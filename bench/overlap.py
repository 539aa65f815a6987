#!/usr/bin/env python
"""Kernel-overlap report from a rocprofv3 kernel trace: busy time (union of kernel intervals) vs summed kernel time,
per stream and overall.  usage: python bench/overlap.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for path in files:
        with open(path) as f:
            for r in csv.DictReader(f):
                if "mt_" in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?")),
                                 r["Kernel_Name"].split("(")[0][:40]))
    rows.sort()
    tail = rows[len(rows) // 2:]                      # second half: past warm-up and capture
    total = sum(e - s for s, e, _, _ in tail)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _, _ in tail:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = tail[-1][1] - tail[0][0]
    per = defaultdict(int)
    for s, e, q, _ in tail:
        per[q] += e - s
    print(f"kernels {len(tail)}  span {span / 1e6:.2f} ms  busy(union) {busy / 1e6:.2f} ms  summed {total / 1e6:.2f} ms  "
          f"overlap factor {total / busy:.2f}  idle {100 * (1 - busy / span):.1f}%")
    print("per stream/queue summed ms:", {k: round(v / 1e6, 2) for k, v in per.items()})
    byname = defaultdict(lambda: [0, 0])
    for s, e, _, n in tail:
        byname[n][0] += 1
        byname[n][1] += e - s
    for n, (c, t) in sorted(byname.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:<42} {c:6d} calls  avg {t / c / 1e3:8.2f} us")


if __name__ == "__main__":
    main()
