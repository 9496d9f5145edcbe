#!/usr/bin/env python3
"""Times the DC operating point of tests/dbmixer.sp Monte-Carlo batches with the general kernel, the faithful
generated kernel (the engine's default) and the fast generated kernel (option dc_fast).  Run under rocprofv3 --kernel-trace --stats
for profiles/r01_dc_kernel_stats.csv.

    python tools/dc_bench.py [--batch 4096 131072]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[4096, 131072])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--netlist", default="dbmixer.sp", help="file under tests/golden/")
    a = ap.parse_args()
    import torch
    from circuitsimulator_amd import Engine, Netlist
    nl = Netlist.from_file(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", a.netlist))
    eng = Engine(nl, 0)
    for B in a.batch:
        params = eng.mc_params(12345, 0.05, 0, B)
        rec = {"batch": B}
        for kern in ("general", "auto", "fast"):
            eng.set_kernel("general" if kern == "general" else "auto")
            eng.set_option("dc_fast", 1 if kern == "fast" else 0)
            x, it, st = eng.dc(params)          # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                x, it, st = eng.dc(params)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.reps
            total = int(it.sum().item())
            rec[{"general": "general", "auto": "faithful_generated", "fast": "fast_generated"}[kern]] = {
                "ms": 1e3 * dt, "nr_iters": total, "nr_iter_inst_per_s": total / dt,
                "replayed_by_general": int((st & 0x80).ne(0).sum().item()),
                "flagged": int((st & 0x1F).ne(0).sum().item())}
        print(json.dumps(rec))


if __name__ == "__main__":
    main()
