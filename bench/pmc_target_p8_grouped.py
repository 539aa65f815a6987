#!/usr/bin/env python
"""Counter-collection target: the grouped bf16 GEMM on a full pack (256 slots x [512 x 512] . [512 x 512]) with
256-tiles on eight waves (launch form 258) next to the 128-tile throughput form (2), forward / input gradient /
weight gradient."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

from gemm_micro import Problem        # noqa: E402
from mitune.ops import lib as L       # noqa: E402


def main():
    so = L.load()
    stream = torch.cuda.current_stream().cuda_stream
    for nstage, edge in ((258, 256), (2, 128)):
        for mode in (0, 1, 2):
            prob = Problem(mode, 256, 512, 512, edge=edge)
            for _ in range(3):
                rc = so.mt_gemm_launch(mode, prob.gt.data_ptr(), prob.tt.data_ptr(), prob.ntiles, 0, 1, 0, 0, 1,
                                       prob.slab if mode == 2 else 0, nstage, stream)
                assert rc == 0, rc
            torch.cuda.synchronize()


if __name__ == "__main__":
    main()
