#!/usr/bin/env python3
"""One-off fuzz beyond tests/test_gpu_parity.py::test_random_netlists_*: many more seeded random circuits through
the general kernels against the oracle (GPU box).  Prints every mismatch; exit code 1 if any.

    python tools/fuzz_general.py [--first 2000] [--count 400]
"""
import argparse
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=2000)
    ap.add_argument("--count", type=int, default=400)
    a = ap.parse_args()
    import numpy as np
    import torch
    spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
    t = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(t)
    from circuitsimulator_amd import Engine, Netlist
    from oracle import binding as orc
    bad = 0
    flagged = 0
    worst = 0.0
    for seed in range(a.first, a.first + a.count):
        rs = np.random.RandomState(seed)
        nl = Netlist.from_text(t._random_netlist(rs, rs.randint(3, 25), rs.randint(0, 8)))
        eng = Engine(nl, 0)
        B, steps = 2, 30
        params = eng.mc_params(seed, 0.05, 0, B)
        r = t._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True)
        ph = params.cpu().numpy()
        for b in range(B):
            xo, ito, sto = orc.dc(nl.ir_ptr, nl.n_unknowns, ph, b)
            o = orc.tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstep * steps, want_step_iters=True)
            problems = []
            if r["dc_iters"][b] != ito:
                problems.append("dc iters %d vs %d" % (r["dc_iters"][b], ito))
            if (r["status"][b] & t.NOFB) != (o["status"] | (sto & 0x1C)):
                problems.append("status %#x vs %#x|%#x" % (r["status"][b], o["status"], sto))
            flagged += int(r["status"][b] != 0)
            if not (o["status"] & 1):
                if not np.array_equal(r["step_iters"][:, b], o["step_iters"]):
                    problems.append("tran iters differ")
                e = t.rel_err(r["x"][:, b], o["x_final"]).max()
                worst = max(worst, e)
                if e >= t.TOL:
                    problems.append("x deviates %.2e" % e)
            if problems:
                bad += 1
                print("seed %d instance %d N=%d: %s" % (seed, b, nl.n_unknowns, "; ".join(problems)), flush=True)
        eng.close()
    print("fuzz: %d circuits, %d mismatching instance runs, %d flagged, worst deviation %.2e" % (a.count, bad, flagged, worst))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
