"""Dev probe: which instances of an MC batch leave the generated kernels, and what their steps look like."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from circuitsimulator_amd import Engine, Netlist
import test_gpu_parity as T

name, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
nl = Netlist.from_file(os.path.join(R, "tests", "golden", name + ".sp"))
eng = Engine(nl, 0)
params = eng.mc_params(12345, 0.05, 0, B)
eng.set_kernel("general")
ref = T._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True)
eng.set_kernel("auto")
got = T._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True)
si = ref["step_iters"]            # [steps, B]
flag = (got["status"] & 0x20) != 0
print("flagged", int(flag.sum()), "of", B, "status bits of flagged:", sorted(set(int(s) for s in got["status"][flag])))
print("all: mean iters/step %.2f, steps >= 35: %d, steps at cap: %d" % (si.mean(), int((si >= 35).sum()), int((si >= si.max()).sum())), "cap seen", int(si.max()))
f = si[:, flag]
print("flagged: mean iters/step %.2f; per instance steps>=35:" % f.mean(), (f >= 35).sum(axis=0)[:40])
print("flagged share of all NR work: %.3f" % (f.sum() / si.sum()))
nf = si[:, ~flag]
print("unflagged: steps >= 35:", int((nf >= 35).sum()))
first = [(int(np.argmax(si[:, b] >= 35)) if (si[:, b] >= 35).any() else -1) for b in np.where(flag)[0][:40]]
print("first slow step per flagged instance:", first)
