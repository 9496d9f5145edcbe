#!/usr/bin/env python3
"""BASELINE configs[4], one GPU's share: tests/dbmixer.sp, 131 072 Monte-Carlo instances, the FULL
50 000-step transient (no shortening of tstop), outputs = final V(102), V(103) per instance.

    python tools/run_mc_share.py [--batch 131072] [--steps 50000]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=131072)
    ap.add_argument("--steps", type=int, default=50000)
    ap.add_argument("--chunk", type=int, default=1000)
    a = ap.parse_args()
    import torch
    from circuitsimulator_amd import Engine, Netlist
    nl = Netlist.from_file(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "dbmixer.sp"))
    eng = Engine(nl, 0)
    B = a.batch
    t0 = time.perf_counter()
    params = eng.mc_params(12345, 0.05, 0, B)
    x, dc_it, st = eng.dc(params)
    torch.cuda.synchronize()
    t_dc = time.perf_counter() - t0
    iters = torch.zeros(B, dtype=torch.int64, device="cuda:0")
    t0 = time.perf_counter()
    for s0 in range(0, a.steps, a.chunk):
        eng.tran(params, x, nl.tstep, s0, min(a.chunk, a.steps - s0), iters, st)
        if (s0 // a.chunk) % 10 == 0:
            torch.cuda.synchronize()
            print("step %d / %d  %.1f s" % (s0, a.steps, time.perf_counter() - t0), flush=True)
    torch.cuda.synchronize()
    t_tr = time.perf_counter() - t0
    v = x[nl.probes, :].cpu().numpy()
    total = int(iters.sum().item())
    rec = {"batch": B, "steps": a.steps, "kernel": eng.tran_kernel, "dc_s": t_dc, "tran_s": t_tr,
           "nr_iters": total, "nr_iters_instance0": int(iters[0]), "rate": total / t_tr,
           "flagged": int((st & 0xA7).ne(0).sum().item()),
           "V102_mean_std": [float(v[0].mean()), float(v[0].std())],
           "V103_mean_std": [float(v[1].mean()), float(v[1].std())],
           "V102_V103_nominal": [float(v[0][0]), float(v[1][0])]}
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
