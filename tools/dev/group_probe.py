"""Dev probe: sixteen-lanes-per-instance kernel against the lane-per-instance and general kernels."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from circuitsimulator_amd import Engine, Netlist
import test_gpu_parity as T
from conftest import rel_err

import sys as _s
cases = [c.split(":") for c in _s.argv[1:]] or [("dbmixer", 64, 60), ("buffer", 64, 100), ("dbmixer", 4096, 400)]
for name, B, steps in [(a, int(b), int(c)) for a, b, c in cases]:
    print("case", name, B, steps, flush=True)
    nl = Netlist.from_file(os.path.join(R, "tests", "golden", name + ".sp"))
    eng = Engine(nl, 0)
    params = eng.mc_params(12345, 0.05, 0, B)
    eng.set_option("lanes_per_instance", 1)
    print("  ref run", flush=True)
    t0 = time.time(); ref = T._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True); t1 = time.time() - t0
    eng.set_option("lanes_per_instance", 16)
    print(name, "lanes for batch:", eng.lanes_for_batch(B), eng.sched_info["text"][-120:])
    print("  group run", flush=True)
    t0 = time.time(); got = T._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True); t2 = time.time() - t0
    same_it = np.array_equal(got["step_iters"], ref["step_iters"])
    e = rel_err(got["x"].T, ref["x"].T).max()
    nfb = int(((got["status"] & 0x20) != 0).sum())
    print(name, "B", B, "step_iters equal:", same_it, "max rel err", e, "fallback flagged:", nfb, "status eq:",
          np.array_equal(got["status"] & T.NOFB, ref["status"] & T.NOFB), "time lane1 %.3f lane16 %.3f" % (t1, t2))
    if not same_it:
        bad = np.argwhere(got["step_iters"] != ref["step_iters"])
        print("  first mismatches", bad[:5], got["step_iters"][bad[0][0], bad[0][1]], ref["step_iters"][bad[0][0], bad[0][1]])
