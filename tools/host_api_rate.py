#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point csim_tran_batch (params in from host memory,
DC + transient on the GPU, final state / counters back), for DESIGN.md section 6.  Never bench.py's `value`.

    python tools/host_api_rate.py [--batch 4096] [--steps 1600]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=1600)
    ap.add_argument("--heavy", type=int, default=0, help="also time a run of this many steps that returns every unknown at every step")
    a = ap.parse_args()
    import numpy as np
    from circuitsimulator_amd import Engine, Netlist
    nl = Netlist.from_file(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "dbmixer.sp"))
    eng = Engine(nl, 0)
    params = np.ascontiguousarray(nl.mc_params_host(12345, 0.05, 0, a.batch).T)      # [B][P] instance-major, host
    tstop = nl.tstep * a.steps
    eng.tran_host(params, tstep=nl.tstep, tstop=nl.tstep * 8)                        # warm-up (allocations, module load)
    out = {}
    cases = [("final state only", None, 1, a.steps), ("2 probes, every 10th step", nl.probes[:2], 10, a.steps)]
    if a.heavy:
        cases.append(("every unknown, every step, %d steps" % a.heavy, list(range(nl.n_unknowns)), 1, a.heavy))
    for label, probes, stride, steps in cases:
        t0 = time.perf_counter()
        wave, xf, it, st = eng.tran_host(params, tstep=nl.tstep, tstop=nl.tstep * steps, probes=probes, out_stride=stride)
        dt = time.perf_counter() - t0
        out[label] = {"seconds": dt, "nr_iters": int(it.sum()), "nr_iter_inst_per_s": float(it.sum() / dt),
                      "flagged": int((st & 0x1F).astype(bool).sum()),
                      "waveform_GB": (wave.nbytes / 1e9) if wave is not None else 0.0}
    print(json.dumps({"batch": a.batch, "steps": a.steps, "host_api": out}))


if __name__ == "__main__":
    main()
