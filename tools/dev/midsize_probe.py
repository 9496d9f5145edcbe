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
for stages in (int(a) for a in (sys.argv[1:] or ["30", "37", "45"])):
    nl = Netlist.from_text(t._amplifier_line(stages))
    eng = Engine(nl, 0)
    if os.environ.get("CSIM_PROBE_GENOPTS"): eng.set_option("jit_gen_opts", os.environ["CSIM_PROBE_GENOPTS"])
    steps = 100
    def run(label, B, params):
        x, dc_it, st = eng.dc(params)
        iters = torch.zeros(B, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.tran(params, x, nl.tstep, 0, steps, iters, st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("N=%3d B=%6d %-10s kernel=%-9s lanes=%2d  %8.1f ms  %.3e NR-iter*inst/s  flagged %d" % (
            nl.n_unknowns, B, label, eng.tran_kernel, eng.lanes_for_batch(B), dt * 1e3, float(iters.sum()) / dt, int((st & 0xA7).ne(0).sum())), flush=True)
        return x.clone(), iters.clone()
    p256 = eng.mc_params(3, 0.03, 0, 256)
    ref = {}
    for B in (256, 4096):
        ref[B] = run("general", B, eng.mc_params(3, 0.03, 0, B))
    t0 = time.perf_counter()
    try:
        eng.jit_scheduled(p256, plan_steps=steps)
        print("   JIT %.1f s: %s" % (time.perf_counter() - t0, eng.sched_info["text"][:200]), flush=True)
    except Exception as e:
        print("   JIT refused:", e, flush=True)
        continue
    for B in (256, 4096, 65536):
        xs, is_ = run("generated", B, eng.mc_params(3, 0.03, 0, B))
        if B in ref:
            xg, ig = ref[B]
            print("   iters equal", bool((is_ == ig).all()), "x rel", float(((xs - xg).abs() / xg.abs().clamp_min(1e-6)).max()), flush=True)
    eng.close()
