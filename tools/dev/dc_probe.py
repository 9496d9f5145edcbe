"""Dev probe (GPU box): DC NR counts of one random circuit -- general kernels vs oracle vs generated DC kernel."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
os.environ.setdefault("CSIM_JIT_DIR", "/tmp/csim_jit_probe")
from circuitsimulator_amd import Engine, Netlist
from oracle import binding as orc
import test_gpu_parity as T
seed = int(sys.argv[1])
rs = np.random.RandomState(seed)
nl = Netlist.from_text(T._random_netlist(rs, rs.randint(3, 25), rs.randint(0, 8)))
eng = Engine(nl, 0)
B = 70
params = eng.mc_params(seed, 0.05, 0, B)
x, it, st = eng.dc(params)
ph = params.cpu().numpy()
ito = np.array([orc.dc(nl.ir_ptr, nl.n_unknowns, ph, b)[1] for b in range(B)])
sto = np.array([orc.dc(nl.ir_ptr, nl.n_unknowns, ph, b)[2] for b in range(B)])
itg = it.cpu().numpy()
print("N", nl.n_unknowns, "general vs oracle: iters equal", np.array_equal(itg, ito), "status equal", np.array_equal(st.cpu().numpy() & 0x1C, sto & 0x1C))
eng.jit_scheduled(params, plan_steps=50)
x2, it2, st2 = eng.dc(params)
its = it2.cpu().numpy()
bad = np.nonzero(its != ito)[0]
print("scheduled vs oracle: mismatching instances", bad, "sched", its[bad], "oracle", ito[bad], "status", st2.cpu().numpy()[bad], "oracle status", sto[bad])
print("sched info", eng.sched_info["text"][:300])
