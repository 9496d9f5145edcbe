"""Dev probe: where do scheduled and general kernels disagree on the inverter chain?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from circuitsimulator_amd import Engine, Netlist
import test_gpu_parity as T

os.environ["CSIM_JIT_DIR"] = "/tmp/jit_probe"
nl = Netlist.from_text(T.INVERTER_CHAIN)
eng = Engine(nl, 0)
B, steps = 96, 600
params = eng.mc_params(4242, 0.05, 0, B)
slow = T._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True)
nonconv = np.where((slow["status"] & 0x02) != 0)[0]
print("nonconv", nonconv)
si = slow["step_iters"]
print("hist of per-step iters (all):", np.bincount(si.ravel(), minlength=51))
seqs, other = eng.record_pivot_schedules(params, int(nonconv[0]), nl.tstep, steps)
print("seqs", seqs, other)
eng.jit_with_schedules([s for s, _ in seqs[:4]])
fast = T._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True, chunks=[250, 350])
fi = fast["step_iters"]
bad = np.argwhere(fi != si)
print("mismatches:", len(bad))
seen = set()
for s, b in bad:
    if b in seen: continue
    seen.add(b)
    print("inst", b, "first mismatch at step", s, "fast", fi[s, b], "slow", si[s, b], "prev steps slow", si[max(0, s - 3):s + 1, b], "fast", fi[max(0, s - 3):s + 1, b],
          "status fast %x slow %x" % (fast["status"][b], slow["status"][b]))
