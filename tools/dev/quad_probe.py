#!/usr/bin/env python3
"""scratch: the four-lanes-per-instance group kernel against the general kernel, then timed at mid-size batches

    python tools/dev/quad_probe.py [netlist] [extra gen opts]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from circuitsimulator_amd import Engine, Netlist
os.environ.setdefault("CSIM_JIT_DIR", "/tmp/csim_jit_quad")
import importlib.util
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
t = importlib.util.module_from_spec(spec)
spec.loader.exec_module(t)

name = sys.argv[1] if len(sys.argv) > 1 else "dbmixer.sp"
extra = sys.argv[2] if len(sys.argv) > 2 else ""
nl = Netlist.from_file(os.path.join(ROOT, "tests", "golden", name))
eng = Engine(nl, 0)
sched, dc = eng.loaded_schedules()
eng.set_option("jit_gen_opts", "group4=1" + ("," + extra if extra else ""))
eng.jit_with_schedules(sched, dc)
print(eng.sched_info["text"], flush=True)

B, steps = 203, 60
params = eng.mc_params(4242, 0.05, 0, B)
eng.set_kernel("general")
ref = t._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True, chunks=[25, 35], probes=nl.probes)
eng.set_kernel("auto")
for lanes in (16, 4):
    eng.set_option("lanes_per_instance", lanes)
    assert eng.lanes_for_batch(B) == lanes, eng.lanes_for_batch(B)
    got = t._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True, chunks=[25, 35], probes=nl.probes)
    print("lanes", lanes, "step_iters equal", np.array_equal(got["step_iters"], ref["step_iters"]),
          "status equal", np.array_equal(got["status"] & t.NOFB, ref["status"]),
          "handed over", int(((got["status"] & 0xA0) != 0).sum()),
          "x rel", t.rel_err(got["x"].T, ref["x"].T).max(),
          "wave rel", t.rel_err(np.moveaxis(got["wave"], 2, 0), np.moveaxis(ref["wave"], 2, 0)).max(), flush=True)

steps = 1000
quick = os.environ.get("QUAD_QUICK") == "1"
for B in ((16384,) if quick else (4096, 8192, 16384, 32768, 65536)):
    params = eng.mc_params(1, 0.05, 0, B)
    x0, _, st0 = eng.dc(params)
    for lanes in ((4,) if quick else (16, 4, 1)):
        eng.set_option("lanes_per_instance", lanes)
        best = None
        for rep in range(3):
            x = x0.clone(); st = st0.clone()
            iters = torch.zeros(B, dtype=torch.int64, device="cuda:0")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.tran(params, x, nl.tstep, 0, steps, iters, st)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print("B %6d lanes %2d: %.1f ms, %.3e NR-iter*inst/s, handed over %d" % (B, lanes, best * 1e3, float(iters.sum()) / best,
              int(((st & 0xA0) != 0).sum())), flush=True)
