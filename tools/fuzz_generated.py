#!/usr/bin/env python3
"""One-off fuzz of the kernel generator: seeded random circuits are planned, generated, compiled (hipcc) and
run; generated kernels (transient and DC) must agree with the general kernels on a Monte-Carlo batch.

    python tools/fuzz_generated.py [--first 3000] [--count 40]
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
    ap.add_argument("--first", type=int, default=3000)
    ap.add_argument("--count", type=int, default=40)
    a = ap.parse_args()
    import numpy as np
    import torch
    os.environ.setdefault("CSIM_JIT_DIR", "/tmp/csim_jit_fuzz")
    spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
    t = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(t)
    from circuitsimulator_amd import Engine, Netlist
    bad = 0
    n_fb = 0
    n_ver = 0
    n_quad = 0
    for seed in range(a.first, a.first + a.count):
        rs = np.random.RandomState(seed)
        nl = Netlist.from_text(t._random_netlist(rs, rs.randint(3, 25), rs.randint(0, 8)))
        eng = Engine(nl, 0)
        B, steps = 70, 50
        params = eng.mc_params(seed, 0.05, 0, B)
        slow = t._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True)
        try:
            eng.jit_scheduled(params, plan_steps=steps)
        except Exception as e:
            print("seed %d N=%d: JIT refused: %s" % (seed, nl.n_unknowns, e), flush=True)
            continue
        clean = (slow["status"] == 0)                    # converged instances: the strict bar applies
        problems = []
        for dc_fast in (0, 1):                           # the faithful generated DC kernel (default) and the fast, guarded one
            eng.set_option("dc_fast", dc_fast)
            fast = t._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True)
            tag = " (dc_fast)" if dc_fast else ""
            if not np.array_equal(fast["dc_iters"][clean], slow["dc_iters"][clean]):
                problems.append("dc iters" + tag)
            if not np.array_equal(fast["step_iters"][:, clean], slow["step_iters"][:, clean]):
                problems.append("tran iters" + tag)
            if not np.array_equal((fast["status"] & t.NOFB)[clean], slow["status"][clean]):
                problems.append("status" + tag)
            if clean.any():
                e = t.rel_err(fast["x"].T[clean], slow["x"].T[clean]).max()
                if e >= t.TOL:
                    problems.append("x deviates %.2e%s" % (e, tag))
            n_fb += int(((fast["status"] & 0xA0) != 0).sum())
            n_fb += int(((fast["status"] & 0xA0) != 0).sum())
        eng.set_option("dc_fast", 0)
        if "group4" in eng.sched_info["text"]:           # the four-lanes-per-instance transient kernel
            eng.set_option("lanes_per_instance", 4)
            quad = t._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True)
            if not np.array_equal(quad["step_iters"][:, clean], slow["step_iters"][:, clean]):
                problems.append("tran iters (4 lanes)")
            if not np.array_equal((quad["status"] & t.NOFB)[clean], slow["status"][clean]):
                problems.append("status (4 lanes)")
            if clean.any():
                e = t.rel_err(quad["x"].T[clean], slow["x"].T[clean]).max()
                if e >= t.TOL:
                    problems.append("x deviates %.2e (4 lanes)" % e)
            n_quad += 1
        n_ver += eng.stat("near_verified")
        if problems:
            bad += 1
            print("seed %d N=%d (%d clean of %d): %s" % (seed, nl.n_unknowns, int(clean.sum()), B, "; ".join(problems)), flush=True)
        eng.close()
        if (seed - a.first) % 10 == 9:
            print("  ... %d of %d circuits, %d with mismatches so far" % (seed - a.first + 1, a.count, bad), flush=True)
    print("fuzz: %d circuits (each with the faithful and with the fast DC kernel), %d with mismatches, %d instance runs "
          "replayed by the general kernels, %d near-threshold decisions verified; %d circuits also on four lanes per instance"
          % (a.count, bad, n_fb, n_ver, n_quad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
