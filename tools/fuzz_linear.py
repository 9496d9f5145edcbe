#!/usr/bin/env python3
"""One-off fuzz of the LINEAR-circuit kernel generator (K1l16 / K1l, csrc/engine/codegen_linear.cpp): seeded random
R/L/C/source networks (no MOSFETs; SIN sources, a DC current source now and then, inductors, the odd floating node) are
planned, generated, compiled (hipcc) and run; the generated kernels perform the reference's operations, so states must
equal the general kernels' BIT FOR BIT and per-step NR counts must be equal.

    python tools/fuzz_linear.py [--first 5000] [--count 60]
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
    ap.add_argument("--first", type=int, default=5000)
    ap.add_argument("--count", type=int, default=60)
    a = ap.parse_args()
    import numpy as np
    import torch
    os.environ.setdefault("CSIM_JIT_DIR", "/tmp/csim_jit_fuzz_lin")
    spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
    t = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(t)
    from circuitsimulator_amd import Engine, Netlist
    bad = refused = n16 = 0
    for seed in range(a.first, a.first + a.count):
        rs = np.random.RandomState(seed)
        nl = Netlist.from_text(t._random_netlist(rs, rs.randint(3, 40), 0))
        eng = Engine(nl, 0)
        B, steps = 37, 60
        params = eng.mc_params(seed, 0.05, 0, B)
        eng.set_kernel("general")
        slow = t._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True, chunks=[25, 35])
        eng.set_kernel("auto")
        try:
            eng.jit_scheduled(params, plan_steps=5)
        except Exception as e:
            refused += 1
            print("seed %d N=%d: JIT refused: %s" % (seed, nl.n_unknowns, str(e)[:120]), flush=True)
            continue
        n16 += int(eng.lanes_for_batch(B) == 16)
        fast = t._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True, chunks=[25, 35])
        clean = (slow["status"] & 0x7) == 0
        problems = []
        if not np.array_equal(fast["step_iters"][:, clean], slow["step_iters"][:, clean]):
            problems.append("tran iters")
        if not np.array_equal((fast["status"] & t.NOFB)[clean], slow["status"][clean]):
            problems.append("status")
        if not np.array_equal(fast["x"][:, clean], slow["x"][:, clean]):
            problems.append("x differs by up to %.2e" % np.abs(fast["x"][:, clean] - slow["x"][:, clean]).max())
        print("seed %d N=%d lanes=%d %s" % (seed, nl.n_unknowns, eng.lanes_for_batch(B), "ok" if not problems else "MISMATCH"), flush=True)
        if problems:
            bad += 1
            print("seed %d N=%d (%d clean of %d): %s" % (seed, nl.n_unknowns, int(clean.sum()), B, "; ".join(problems)), flush=True)
        eng.close()
    print("fuzz (linear): %d circuits, %d refused by the JIT, %d on the sixteen-lane kernel, %d with mismatches"
          % (a.count, refused, n16, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
