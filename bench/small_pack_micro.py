#!/usr/bin/env python
"""Microseconds per optimiser step of a small exact-f32 MLP pack: the 16 survivors of the headline search (their widths
and batch sizes as the seed-0 search of bench.py produces them) trained by tile launches and by the row-block step.

    python bench/small_pack_micro.py [--rows 250000] [--forms 0,1s1,1s2,1s4] [--epochs 2]

A form is the ``rowblock`` option ("0": tile launches, "1": row blocks), optionally followed by ``s<n>``: the most
workgroups that may share one row block (``rowblock_split``), ``c<cost>``: the exchange cost the split rule assumes,
``b<n>``: the most workgroups a split work list may hold.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

SURVIVORS = [([512, 256, 128], 256), ([512, 128, 512], 256), ([512, 512, 128], 512), ([512, 512, 256], 256),
             ([512, 128, 64], 256), ([256, 512, 512], 256), ([256, 256, 512], 256), ([512, 64, 512], 256),
             ([256, 512, 64], 512), ([512, 128, 256], 256), ([256, 64, 256], 256), ([512, 512, 64], 256),
             ([512, 128, 64], 256), ([512, 64, 256], 256), ([512, 256, 512], 512), ([512, 64, 128], 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=250_000)
    ap.add_argument("--forms", default="0,1")
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--slots", type=int, default=16)
    ap.add_argument("--trials", type=int, default=0, help="64 or 256: the trials of rounds 2-4 / round 1 of the headline "
                    "search instead of the 16 survivors (configurations of the seed-0 search, survivor lists from "
                    "profiles/r2_parity_full.jsonl)")
    ap.add_argument("--only", type=int, default=-1, help="train just this one of the survivors")
    ap.add_argument("--check", action="store_true", help="compare the trained weights of the forms bit for bit")
    ap.add_argument("--stamps", action="store_true", help="print the row-block kernel's phase stamps of its last launch")
    args = ap.parse_args()
    import bench as headline
    from mitune.options import OPTIONS
    from mitune.packed import PackedTrainable, engine_hip
    dev = torch.device("cuda", 0)
    train, val = headline.make_data(args.rows, 64, dev)
    out, states = {}, {}
    survivors = SURVIVORS[:args.slots] if args.only < 0 else [SURVIVORS[args.only]]
    if args.trials:
        from mitune.tune import space as sp
        cfgs = list(sp.generate_variants(headline.search_space(), 256, 0))
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        with open(os.path.join(root, "profiles", "r2_parity_full.jsonl")) as f:
            rec = json.loads(f.readline())
        ids = {16: rec["past_rung4"], 64: rec["past_rung1"], 256: list(range(256))}[args.trials]
        survivors = [([cfgs[i]["width_0"], cfgs[i]["width_1"], cfgs[i]["width_2"]], cfgs[i]["batch_size"]) for i in ids]
        args.slots = len(survivors)
    prof = None
    if args.stamps:
        import ctypes as C
        from mitune.ops import lib as L
        prof = torch.zeros(4096 * 16, device=dev, dtype=torch.int64)
        L.load().mt_rowblock_set_prof(C.c_void_p(prof.data_ptr()))
    import re
    for form in args.forms.split(","):
        m = re.fullmatch(r"([01a-z]+?)(?:s(\d+))?(?:c(\d+))?(?:b(\d+))?", form)
        OPTIONS.rowblock = m.group(1)
        if m.group(2):
            OPTIONS.rowblock_split = int(m.group(2))
        if m.group(3):
            OPTIONS.rowblock_split_cost = int(m.group(3))
        if m.group(4):
            OPTIONS.rowblock_wg_budget = int(m.group(4))
        tr = PackedTrainable(train=train, val=val, max_epochs=args.epochs, engine="hip", dtype="f32", capacity=args.slots,
                             val_chunk=4096, model_kwargs={"max_width": 512, "max_layers": 3, "max_batch": 1024})
        eng = engine_hip.HipPackEngine(tr, device=dev, capacity=args.slots, seed=0)
        for slot, (widths, bs) in enumerate(survivors):
            eng.load(slot, slot, {"hidden_sizes": widths, "batch_size": bs, "learning_rate": 2e-3, "weight_decay": 1e-4,
                                  "optimizer": "adamw", "loss_function": "mse"})
        eng.train_epoch()                       # warm-up: tables, first touch
        torch.cuda.synchronize()
        steps = max(eng._round_steps.values())
        t0 = time.perf_counter()
        for _ in range(args.epochs):
            eng.train_epoch()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ev = eng.evaluate()
        out[form] = {"s_per_epoch": dt / args.epochs, "us_per_step": dt / (args.epochs * steps) * 1e6, "steps_per_epoch": steps,
                     "chains": [len(ch.ids) for ch in eng._chains],
                     "rb_blocks": [int(ch.plan.rb_nblocks) for ch in eng._chains],
                     "rb_split": [eng._rowblock_split(ch.ids, int(ch.plan.rb_mt)) for ch in eng._chains if ch.plan.rb_nblocks],
                     "val_mse_slot0": ev[0]["validation_mse"], "val_mse_slot3": ev[3]["validation_mse"] if 3 in ev else None}
        if prof is not None and OPTIONS.rowblock != "0":
            import numpy as np
            blocks = eng._tables[eng._chains[0].tag + "rb_blocks"].cpu().numpy().view(np.int32).reshape(-1, 2)
            st = prof.cpu().numpy().reshape(-1, 16)[:len(blocks)]
            rows = []
            for b, (slot, row0) in enumerate(blocks):
                if slot < 0 or st[b, 0] == 0:
                    continue
                row0 &= 0xFFFFF
                v = st[b][st[b] > 0]
                rows.append((int(slot), int(row0), [int(x - v[0]) for x in v]))
            by_slot = {}
            for slot, row0, v in rows:
                by_slot.setdefault(slot, []).append(v)
            t0 = min(int(st[b, 0]) for b in range(len(blocks)) if blocks[b][0] >= 0 and st[b, 0] > 0)
            t1 = max(int(st[b][st[b] > 0][-1]) for b in range(len(blocks)) if blocks[b][0] >= 0 and st[b, 0] > 0)
            out[form]["kernel_ticks_first_start_to_last_end"] = t1 - t0
            out[form]["stamps_ticks_median_by_slot"] = {
                str(slot): [int(np.median([v[k] for v in vs if len(v) > k])) for k in range(max(len(v) for v in vs))]
                for slot, vs in sorted(by_slot.items())}
            out[form]["widths_by_slot"] = {str(i): w for i, (w, _) in enumerate(survivors)}
        if args.check:
            states[form] = {slot: eng.export(slot)["model"] for slot in range(len(survivors))}
        eng.close()
        del eng
    if args.check and len(states) > 1:
        forms = list(states)
        bad = []
        for slot in states[forms[0]]:
            for k, v in states[forms[0]][slot].items():
                for f in forms[1:]:
                    if not torch.equal(v, states[f][slot][k]):
                        bad.append((slot, k, f, float((v - states[f][slot][k]).abs().max())))
        out["mismatches"] = bad[:20]
        out["n_mismatches"] = len(bad)
    print(json.dumps({"slots": args.slots, "rows": args.rows, "forms": out}))


if __name__ == "__main__":
    main()
