#!/usr/bin/env python3
"""Fuzz of the BIT-FAITHFUL generated kernels: random circuits are planned, generated, compiled and run on the faithful
family (csim_engine_set_kernel 3: K2f for the operating point, K1f for the transient); every instance the generated
kernels finished themselves (no hand-over to the general kernel) must equal the general kernel BIT FOR BIT -- operating
point, final state, per-step NR counts, status.

    python tools/fuzz_faithful.py [--first 20000] [--count 100]
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
    ap.add_argument("--first", type=int, default=20000)
    ap.add_argument("--count", type=int, default=100)
    ap.add_argument("--tstep-scale", type=float, default=1.0,
                    help="multiply the circuits' time step (large steps make the MOSFETs switch hard: several pivot sequences "
                         "per run, hand-overs to the general kernel and back)")
    ap.add_argument("--stress", action="store_true",
                    help="add degenerate parts to every circuit: a node that is only a MOSFET gate, a node that hangs on "
                         "capacitors only (their matrix rows are gmin alone in DC: the last bit of gmin shows)")
    a = ap.parse_args()
    import numpy as np
    import torch
    os.environ.setdefault("CSIM_JIT_DIR", "/tmp/csim_jit_fuzz_faithful")
    spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
    t = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(t)
    from circuitsimulator_amd import Engine, Netlist
    bad = n_run = n_kept_dc = n_kept_tr = 0
    for seed in range(a.first, a.first + a.count):
        rs = np.random.RandomState(seed)
        text = t._random_netlist(rs, rs.randint(3, 25), rs.randint(0, 8))
        if a.stress:
            nodes = sorted({w for ln in text.splitlines() if ln[:1] in "RCL" for w in ln.split()[1:3] if w.startswith("n")})
            extra = []
            if len(nodes) >= 2:
                d_, s_ = rs.choice(nodes, 2, replace=False)
                extra.append("MG1 %s gx1 %s n %.3ge-6 0.35e-6 2" % (d_, s_, rs.uniform(5, 40)))          # gx1: a gate and nothing else
                extra.append("MG2 %s gx2 vdd p %.3ge-6 0.35e-6 1" % (rs.choice(nodes), rs.uniform(5, 40)))
                extra.append("CG2 gx2 %s %.4g" % (rs.choice(nodes), 10 ** rs.uniform(-14, -12)))           # gx2: a gate and a capacitor
                extra.append("CX1 cx1 0 %.4g" % 10 ** rs.uniform(-14, -12))                                 # cx1: capacitors only
                extra.append("CX2 cx1 %s %.4g" % (rs.choice(nodes), 10 ** rs.uniform(-14, -12)))
            text = text.replace(".MODEL 1", "\n".join(extra) + "\n.MODEL 1", 1)
        nl = Netlist.from_text(text)
        eng = Engine(nl, 0)
        B, steps = 70, 50
        params = eng.mc_params(seed, 0.05, 0, B)
        eng.set_kernel("general")
        tstep = nl.tstep * a.tstep_scale
        slow = t._run_tran(torch, eng, params, steps, tstep, want_step_iters=True)
        eng.set_kernel("auto")
        try:
            eng.jit_scheduled(params, tstep=tstep, plan_steps=steps)
            eng.set_kernel("faithful")
        except Exception as e:
            print("seed %d N=%d: no faithful kernel: %s" % (seed, nl.n_unknowns, e), flush=True)
            eng.close()
            continue
        n_run += 1
        fast = t._run_tran(torch, eng, params, steps, tstep, want_step_iters=True)
        problems = []
        st = fast["status"]
        kept_dc = (st & 0x80) == 0                       # operating point finished by the generated DC kernel
        kept_tr = kept_dc & ((st & 0x20) == 0)           # ... and the transient never went to the general kernel
        n_kept_dc += int(kept_dc.sum())
        n_kept_tr += int(kept_tr.sum())
        if not np.array_equal(fast["dc_iters"], slow["dc_iters"]):
            problems.append("dc iters")
        if not np.array_equal(fast["x_dc"], slow["x_dc"]):
            d = np.abs(fast["x_dc"] - slow["x_dc"])
            problems.append("x_dc differs in %d entries of %d instances (worst %.2e; %d of them finished on K2f)" % (
                int((d != 0).sum()), int((d != 0).any(axis=0).sum()), d.max(), int(((d != 0).any(axis=0) & kept_dc).sum())))
        if not np.array_equal(fast["step_iters"], slow["step_iters"]):
            problems.append("tran iters")
        if not np.array_equal(st & t.NOFB, slow["status"]):
            problems.append("status")
        if not np.array_equal(fast["x"], slow["x"]):
            d = np.abs(fast["x"] - slow["x"])
            problems.append("x differs in %d entries of %d instances (worst %.2e)" % (int((d != 0).sum()), int((d != 0).any(axis=0).sum()), d.max()))
        if problems:
            bad += 1
            print("seed %d N=%d: %s" % (seed, nl.n_unknowns, "; ".join(problems)), flush=True)
        eng.close()
        if (seed - a.first) % 10 == 9:
            print("  ... %d of %d circuits, %d not bitwise so far" % (seed - a.first + 1, a.count, bad), flush=True)
    print("fuzz (faithful family, bitwise): %d circuits run, %d with differences; %d operating points and %d transients "
          "finished on the generated kernels" % (n_run, bad, n_kept_dc, n_kept_tr))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
