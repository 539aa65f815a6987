"""rocprofv3 command lines that are safe on a shared MI355X pool, plus a kernel-stats reader.

Counter collection and tracing are kept in separate runs (``--pmc`` never together with system/runtime
tracing), counters are grouped per pass within the hardware limits (8 SQ, 4 TCC -- FETCH_SIZE uses 3 of
them, WRITE_SIZE 2 --, 4 TCP, 2 TA, 2 TD, 2 GRBM), and the profiled program is always the interpreter
itself, never a shell or ``env`` wrapper.
"""
from __future__ import annotations

import csv
import sys
from typing import Dict, List, Sequence

PMC_PASSES: Dict[str, List[str]] = {
    "mfma": ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU",
             "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"],
    "lds": ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
            "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_INSTS_VMEM"],
    "hbm_read": ["FETCH_SIZE", "TCC_HIT_sum"],
    "hbm_write": ["WRITE_SIZE", "TCC_MISS_sum", "TCC_REQ_sum"],
}


def kernel_trace_cmd(program: Sequence[str], out_dir: str) -> List[str]:
    return ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out_dir, "--", sys.executable,
            *program]


def pmc_cmd(program: Sequence[str], out_dir: str, pass_name: str) -> List[str]:
    return ["rocprofv3", "--pmc", *PMC_PASSES[pass_name], "--kernel-trace", "--output-format", "csv", "-d", out_dir, "--",
            sys.executable, *program]


def read_kernel_stats(path: str, prefix: str = "mt_") -> List[Dict[str, float]]:
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if prefix in r["Name"]:
                rows.append({"name": r["Name"].split("(")[0], "calls": int(r["Calls"]),
                             "total_ms": float(r["TotalDurationNs"]) / 1e6, "mean_us": float(r["AverageNs"]) / 1e3,
                             "pct": float(r["Percentage"])})
    return rows
