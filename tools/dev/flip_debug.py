#!/usr/bin/env python3
"""scratch: where does instance b of the dbmixer Monte-Carlo batch take another NR count on a fast kernel?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from circuitsimulator_amd import Engine, Netlist
os.environ.setdefault("CSIM_JIT_DIR", "/tmp/csim_jit_dbg")
b0 = int(sys.argv[1]) if len(sys.argv) > 1 else 125528
steps = 50000
nl = Netlist.from_file("tests/golden/dbmixer.sp")
eng = Engine(nl, 0)
B = 64
params = eng.mc_params(12345, 0.05, b0, B)          # column 0 = instance b0
eng.set_kernel("general")
x0, _, st0 = eng.dc(params)
def run(tag):
    x = x0.clone(); st = st0.clone(); it = torch.zeros(B, dtype=torch.int64, device="cuda:0")
    si = torch.zeros((steps, B), dtype=torch.int32, device="cuda:0")
    v0, r0 = eng.stat("near_verified"), eng.stat("near_rolled_back")
    for s0 in range(0, steps, 1000):
        eng.tran(params, x, nl.tstep, s0, 1000, it, st, step_iters=si[s0:s0 + 1000])
    torch.cuda.synchronize()
    print(tag, "iters[0]", int(it[0]), "verified", eng.stat("near_verified") - v0, "rolled back", eng.stat("near_rolled_back") - r0, flush=True)
    return si.cpu().numpy(), x.cpu().numpy()
eng.set_kernel("faithful"); sf, xf = run("faithful")
eng.set_kernel("scheduled"); eng.set_option("lanes_per_instance", 1); s1, x1 = run("fast lanes=1 (guard)")
eng.set_option("lanes_per_instance", 16); s16, x16 = run("fast lanes=16 (guard)")
for name, s in (("lanes1", s1), ("lanes16", s16)):
    d = np.nonzero((s != sf).any(axis=1))[0]
    print(name, "steps with another count:", d[:10], [(int(k), sf[k, :3].tolist(), s[k, :3].tolist()) for k in d[:3]])
sched, dc = eng.loaded_schedules()
eng.set_option("jit_gen_opts", "near_band=0"); eng.jit_with_schedules(sched, dc)
eng.set_option("lanes_per_instance", 1); s0_, x0_ = run("fast lanes=1 (no guard, sqrt)")
d = np.nonzero((s0_ != sf).any(axis=1))[0]
print("no guard: steps with another count:", d[:10])
