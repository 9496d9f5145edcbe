#!/usr/bin/env python3
"""scratch: rates on mid-size nonlinear circuits (amplifier lines of 30 / 60 / 120 stages: N = 63 / 123 / 243)"""
import os, sys, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
os.environ.setdefault("CSIM_JIT_DIR", "/tmp/csim_jit_mid")
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
from circuitsimulator_amd import Engine, Netlist
for stages in (int(a) for a in (sys.argv[1:] or ["30", "60", "120"])):
    nl = Netlist.from_text(t._amplifier_line(stages))
    eng = Engine(nl, 0)
    for B in (256, 4096):
        steps = 100
        params = eng.mc_params(3, 0.03, 0, B)
        def run(label):
            x, dc_it, st = eng.dc(params)
            iters = torch.zeros(B, dtype=torch.int64, device="cuda:0")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.tran(params, x, nl.tstep, 0, steps, iters, st)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print("N=%3d B=%5d %-10s kernel=%-9s lanes=%2d  %8.1f ms  %.3e NR-iter*inst/s  flagged %d" % (
                nl.n_unknowns, B, label, eng.tran_kernel, eng.lanes_for_batch(B), dt * 1e3, float(iters.sum()) / dt, int((st & 0xA7).ne(0).sum())), flush=True)
            return x.clone(), iters.clone()
        xg, ig = run("general")
        if B == 256:
            t0 = time.perf_counter()
            try:
                eng.jit_scheduled(params, plan_steps=steps)
                print("   JIT %.1f s: %s" % (time.perf_counter() - t0, eng.sched_info["text"][:160]), flush=True)
            except Exception as e:
                print("   JIT refused:", e, flush=True)
        if eng.tran_kernel == "scheduled":
            xs, is_ = run("generated")
            print("   iters equal", bool((is_ == ig).all()), "x rel", float(((xs - xg).abs() / xg.abs().clamp_min(1e-6)).max()), flush=True)
            eng.set_kernel("general")
            if B == 4096: eng.set_kernel("auto")
        if B == 256 and eng.tran_kernel == "general": pass
    eng.close()
