"""Dev probe (GPU box): which pivot sequences beyond the shipped ones Monte-Carlo batches of buffer.sp need."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, numpy as np
os.environ.setdefault("CSIM_JIT_DIR", "/tmp/jit_ref")
from circuitsimulator_amd import Engine, Netlist
import test_gpu_parity as T
nl = Netlist.from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "buffer.sp")); eng = Engine(nl, 0)
for seed, B in ((12345, 4096), (777, 4096), (99, 8192)):
    params = eng.mc_params(seed, 0.05, 0, B)
    r = T._run_tran(torch, eng, params, 600, nl.tstep)
    n0 = int(((r["status"] & 0x20) != 0).sum())
    added = eng.refine_schedules(params, r["status"], nl.tstep, n_steps=600, max_instances=16) if n0 else 0
    r = T._run_tran(torch, eng, params, 600, nl.tstep)
    print("seed", seed, "B", B, "flagged before", n0, "added", added, "after", int(((r["status"] & 0x20) != 0).sum()))
print("\n".join(eng.loaded_schedules()[0]))
