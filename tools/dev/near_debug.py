#!/usr/bin/env python3
"""scratch: near-threshold events per kernel at a wide guard band"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from circuitsimulator_amd import Engine, Netlist
os.environ.setdefault("CSIM_JIT_DIR", "/tmp/csim_jit_dbg")
nl = Netlist.from_file("tests/golden/dbmixer.sp")
eng = Engine(nl, 0)
B = 200
params = eng.mc_params(4242, 0.05, 0, B)
sched, dc = eng.loaded_schedules()
eng.set_option("jit_gen_opts", "near_band=" + (sys.argv[1] if len(sys.argv) > 1 else "0.02"))
eng.jit_with_schedules(sched, dc)
print(eng.sched_info["text"][:200])
for lanes in (16, 1):
    eng.set_option("lanes_per_instance", lanes)
    x, _, st = eng.dc(params)
    iters = torch.zeros(B, dtype=torch.int64, device="cuda:0")
    for s0, n in ((0, 25), (25, 35)):
        v0 = eng.stat("near_verified")
        eng.tran(params, x, nl.tstep, s0, n, iters, st)
        torch.cuda.synchronize()
        print("lanes", lanes, "chunk", s0, n, "verified", eng.stat("near_verified") - v0, "rolled", eng.stat("near_rolled_back"),
              "status bits", sorted(set(int(v) for v in st.cpu().tolist())), "iters0", int(iters[0]))
