#!/usr/bin/env python3
"""scratch: DC operating points of one fuzz seed, general kernel against the generated DC kernels"""
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
nl = Netlist.from_text(t._random_netlist(rs, rs.randint(3, 25), rs.randint(0, 8)))
eng = Engine(nl, 0)
B = 70
params = eng.mc_params(seed, 0.05, 0, B)
eng.set_kernel("general")
xg, itg, stg = eng.dc(params)
eng.set_kernel("auto")
eng.jit_scheduled(params, plan_steps=50)
print(eng.sched_info["text"][:300])
for dc_fast in (0, 1):
    eng.set_option("dc_fast", dc_fast)
    x, it, st = eng.dc(params)
    neq = (x != xg)
    print("dc_fast", dc_fast, "iters equal", bool((it == itg).all()), "entries not bitwise equal", int(neq.sum()), "in instances", sorted(set(neq.nonzero()[:, 1].tolist()))[:20],
          "status bits", sorted(set(int(v) for v in st.tolist())), "general status", sorted(set(int(v) for v in stg.tolist())))
    if neq.any():
        d = (x - xg).abs()
        i = int(d.argmax()); n, b = divmod(i, B)
        print("   worst abs", float(d.max()), "unknown", n, nl.eq_names[n] if n < len(nl.eq_names) else "?", "instance", b, "values", float(x[n, b]), float(xg[n, b]), "iters", int(it[b]), int(itg[b]), "status", hex(int(st[b])))
        ph = params.cpu().numpy()
        xo, ito, sto = t._orc().dc(nl.ir_ptr, nl.n_unknowns, ph, b)
        print("   oracle: iters", ito, "x[n]", xo[n], " general == oracle bitwise:", bool((xg[:, b].cpu().numpy() == xo).all()), " generated == oracle bitwise:", bool((x[:, b].cpu().numpy() == xo).all()))

# ---- which DC alternative is responsible?  (generated kernels re-made with one recorded DC sequence at a time)
sched, dc_sched = eng.loaded_schedules()
print("transient schedules", sched, "dc schedules", dc_sched)
eng.set_option("dc_fast", 0)
for label, dcs in [("both", dc_sched)] + [("only alt %d" % i, [d]) for i, d in enumerate(dc_sched)] + [("reversed", list(reversed(dc_sched)))]:
    eng.jit_with_schedules(sched, dcs)
    x, it, st = eng.dc(params)
    kept = (st & 0x80) == 0
    neq = (x != xg) & kept[None, :]
    print(label, ": finished on the generated kernel", int(kept.sum()), "of", B, "; entries not bitwise equal among those", int(neq.sum()),
          "in", len(set(neq.nonzero()[:, 1].tolist())), "instances; iters equal", bool((it == itg).all()))

# ---- which sequences does the reference's pivoting take, per instance?
eng.jit_with_schedules(sched, dc_sched)
x, it, st = eng.dc(params)
kept = ((st & 0x80) == 0).cpu().numpy()
neq_inst = (x != xg).any(dim=0).cpu().numpy()
for b in list(np.nonzero(kept & neq_inst)[0][:3]) + list(np.nonzero(kept & ~neq_inst)[0][:2]) + list(np.nonzero(~kept)[0][:2]):
    seqs = eng.record_dc_pivot_schedules(params, instance=int(b), max_alts=8)
    print("instance", int(b), "finished on generated" if kept[b] else "replayed", "bitwise" if not neq_inst[b] else "NOT bitwise", "sequences:", seqs)
