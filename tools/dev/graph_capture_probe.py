#!/usr/bin/env python3
"""scratch: can an asynchronous csim_tran_batch_dev call be captured in a HIP graph and replayed?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from circuitsimulator_amd import Engine, Netlist
nl = Netlist.from_file("tests/golden/dbmixer.sp")
eng = Engine(nl, 0)
B, n = 512, 200
params = eng.mc_params(1, 0.05, 0, B)
x0, _, st0 = eng.dc(params)
# direct
x = x0.clone(); st = st0.clone(); it = torch.zeros(B, dtype=torch.int64, device="cuda:0")
eng.tran(params, x, nl.tstep, 0, n, it, st)
torch.cuda.synchronize()
ref_x, ref_it = x.clone(), it.clone()
eng.set_option("hybrid_sync", 0)
xs = x0.clone(); sts = st0.clone(); its = torch.zeros(B, dtype=torch.int64, device="cuda:0")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    eng.tran(params, xs, nl.tstep, 0, n, its, sts)          # warm-up on the side stream: buffers get allocated
torch.cuda.synchronize()
xs.copy_(x0); sts.copy_(st0); its.zero_()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        eng.tran(params, xs, nl.tstep, 0, n, its, sts)
    print("captured")
    for rep in range(2):
        xs.copy_(x0); sts.copy_(st0); its.zero_()
        g.replay()
        torch.cuda.synchronize()
        print("replay", rep, "state equal:", bool(torch.equal(xs, ref_x)), "iters equal:", bool(torch.equal(its, ref_it)))
except Exception as e:
    print("capture failed:", type(e).__name__, e)
