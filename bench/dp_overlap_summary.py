#!/usr/bin/env python
"""Reduce a rocprofv3 trace of ``bench/dp_overlap_target.py`` to a table of what ran while the buckets moved.

    python bench/dp_overlap_summary.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv>

For every transfer on the communication stream (device copies of a one-rank run; RCCL kernels of a real one) of the last
traced epoch: its size class, duration, and how much of it overlapped compute kernels, by kernel family.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(path, name_key):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get(name_key, "") or "", r))
    return sorted(out)


def family(name):
    for key, fam in (("wgrad", "weight gradient"), ("mt_gemm", "forward / input gradient GEMM"), ("slab", "bucket sum / cast"),
                     ("head", "head + loss"), ("optim", "optimiser"), ("gather", "gather")):
        if key in name:
            return fam
    return "other"


def main():
    d = sys.argv[1]
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    mt = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
    kernels = load(kt[0], "Kernel_Name")
    comm = [(s, e, "copy " + r.get("Direction", ""), r) for s, e, _, r in load(mt[0], "Direction")] if mt else []
    comm = [c for c in comm if "DEVICE_TO_DEVICE" in c[2]]
    # RCCL kernels: the ring kernels of a multi-rank run, the one-rank reduce kernel of a forced one-rank run
    comm += [(s, e, "rccl kernel", r) for s, e, n, r in kernels if any(k in n.lower() for k in ("nccl", "rccl", "onerank"))]
    compute = [(s, e, n) for s, e, n, _ in kernels if "mt_" in n]
    if not comm or not compute:
        print(f"nothing to summarise: {len(comm)} transfers, {len(compute)} compute kernels")
        return
    # the last epoch: transfers after the largest gap between compute kernels
    gaps = [(compute[i + 1][0] - compute[i][1], i) for i in range(len(compute) - 1)]
    cut = compute[max(gaps)[1] + 1][0] if gaps else 0
    compute = [c for c in compute if c[0] >= cut]
    comm = [c for c in comm if c[0] >= cut and (c[1] - c[0]) > 2000]         # transfers of more than 2 us: the buckets
    t0, t1 = compute[0][0], compute[-1][1]
    steps = sum("gather" in n for _, _, n in compute)
    print(f"last epoch: {steps} steps, span {(t1 - t0) / 1e6:.2f} ms = {(t1 - t0) / 1e3 / max(steps, 1):.0f} us per step, "
          f"{len(compute)} compute kernels, {len(comm)} bucket transfers on the communication stream")
    busy = sum(e - s for s, e, _ in compute)
    print(f"compute-stream busy {busy / 1e6:.2f} ms ({100.0 * busy / (t1 - t0):.1f} % of the span)")
    tot, hidden = 0, 0
    by_fam = defaultdict(int)
    rows = []
    for s, e, what, _ in comm:
        ov = defaultdict(int)
        for cs, ce, n in compute:
            if ce <= s or cs >= e:
                continue
            ov[family(n)] += min(e, ce) - max(s, cs)
        covered = min(sum(ov.values()), e - s)
        tot += e - s
        hidden += covered
        for k, v in ov.items():
            by_fam[k] += v
        rows.append((s - t0, e - s, covered, max(ov, key=ov.get) if ov else "-"))
    print(f"transfers: {tot / 1e6:.2f} ms in total, {hidden / 1e6:.2f} ms ({100.0 * hidden / max(tot, 1):.1f} %) while a compute "
          "kernel was running")
    print("compute time under the transfers, by kernel family:")
    for k, v in sorted(by_fam.items(), key=lambda kv: -kv[1]):
        print(f"    {k:<32} {v / 1e6:8.2f} ms")
    print("first step's transfers (start after epoch start, duration, overlapped, mostly with):")
    for at, dur, cov, fam in rows[:max(1, len(rows) // max(steps, 1))]:
        print(f"    +{at / 1e3:9.1f} us  {dur / 1e3:8.1f} us  {cov / 1e3:8.1f} us  {fam}")


if __name__ == "__main__":
    main()
