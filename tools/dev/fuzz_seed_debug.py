#!/usr/bin/env python3
"""scratch: one seed of tools/fuzz_generated.py in detail"""
import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
os.environ.setdefault("CSIM_JIT_DIR", "/tmp/csim_jit_dbg")
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
from circuitsimulator_amd import Engine, Netlist
seed = int(sys.argv[1])
rs = np.random.RandomState(seed)
text = t._random_netlist(rs, rs.randint(3, 25), rs.randint(0, 8))
print(text)
nl = Netlist.from_text(text)
eng = Engine(nl, 0)
B, steps = 70, 50
params = eng.mc_params(seed, 0.05, 0, B)
slow = t._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True)
eng.jit_scheduled(params, plan_steps=steps)
print(eng.sched_info["text"])
runs = {}
for name, setup in (("lanes16", lambda: eng.set_option("lanes_per_instance", 16)), ("lanes4", lambda: eng.set_option("lanes_per_instance", 4)),
                    ("lanes1", lambda: eng.set_option("lanes_per_instance", 1)), ("faithful", lambda: eng.set_kernel("faithful"))):
    setup()
    runs[name] = t._run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True)
    r = runs[name]
    e = t.rel_err(r["x"].T, slow["x"].T)
    b, n = np.unravel_index(np.argmax(e), e.shape)
    print(name, "iters equal", np.array_equal(r["step_iters"], slow["step_iters"]), "max rel", e.max(), "instance", b, "unknown", n, nl.eq_names[n] if n < len(nl.eq_names) else "?",
          "values", r["x"][n, b], slow["x"][n, b], "status", hex(int(r["status"][b])), "dc rel", t.rel_err(r["x_dc"].T, slow["x_dc"].T).max())
ph = params.cpu().numpy()
e = t.rel_err(runs["lanes16"]["x"].T, slow["x"].T)
b = int(np.unravel_index(np.argmax(e), e.shape)[0])
o = t._orc().tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstep * steps, want_step_iters=True)
print("oracle vs general, instance", b, ":", t.rel_err(slow["x"][:, b], o["x_final"]).max(), " oracle vs lanes16:", t.rel_err(runs["lanes16"]["x"][:, b], o["x_final"]).max())
print("x general", slow["x"][:, b])
print("x lanes16", runs["lanes16"]["x"][:, b])
print("step iters", slow["step_iters"][:, b])
