#!/usr/bin/env python3
"""scratch: rate of the HOST-pointer API (csim_tran_batch: parameters in, DC + transient, results out over PCIe)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from circuitsimulator_amd import Engine, Netlist
nl = Netlist.from_file(os.path.join(ROOT, "tests", "golden", "dbmixer.sp"))
eng = Engine(nl, 0)
for B in (4096, 16384, 65536):
    table = np.ascontiguousarray(nl.mc_params_host(12345, 0.05, 0, B).T)     # [B][P], instance-major, host memory
    steps = 6000
    for probes, stride, label in ((None, 1, "final state only"), (nl.probes[:2] if len(nl.probes) >= 2 else [0, 1], 10, "two probes, every 10th step")):
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            wave, xf, it, st = eng.tran_host(params=table, tstep=nl.tstep, tstop=nl.tstep * steps, probes=probes, out_stride=stride)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print("B %6d %-28s %.3f s  %.3e NR-iter*inst/s (DC + %d steps, host tables in and out)%s" % (
            B, label, best, float(it.sum()) / best, steps, "" if wave is None else "  waveform %.1f MB" % (wave.nbytes / 1e6)), flush=True)
