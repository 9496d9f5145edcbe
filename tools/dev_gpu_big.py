import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from circuitsimulator_amd import Netlist, Engine
from circuitsimulator_amd.workloads import rc_ladder_netlist
from oracle import binding as orc
for nn, steps in ((70, 30), (256, 20)):
    nl = Netlist.from_text(rc_ladder_netlist(nn))
    print('N', nl.n_unknowns, 'elems', nl.n_elems, 'P', nl.n_params)
    eng = Engine(nl, 0)
    B = 4
    params = eng.mc_params(12345, 0.05, 0, B)
    t = time.time(); x, it, st = eng.dc(params); torch.cuda.synchronize(); print('dc time', time.time()-t)
    ph = params.cpu().numpy()
    for b in (0, 1):
        xo, ito, sto = orc.dc(nl.ir_ptr, nl.n_unknowns, ph, b)
        e = np.abs(x[:, b].cpu().numpy() - xo) / np.maximum(np.abs(xo), 1e-6)
        print(' DC b', b, 'iters', int(it[b]), ito, 'status', int(st[b]), sto, 'max rel', e.max())
    iters = torch.zeros(B, dtype=torch.int64, device='cuda:0')
    t = time.time(); eng.tran(params, x, nl.tstep, 0, steps, iters, st); torch.cuda.synchronize(); dt = time.time()-t
    print(' tran time', dt, 'iters', iters.tolist(), 'status', st.tolist())
    for b in (0, 1):
        o = orc.tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstep*steps, want_rows=False)
        e = np.abs(x[:, b].cpu().numpy() - o['x_final']) / np.maximum(np.abs(o['x_final']), 1e-6)
        print(' TRAN b', b, 'iters', int(iters[b]), o['iters'], 'max rel', e.max())
    sched, nlu, nd = eng.record_pivot_schedule(params, 0, None, 10)
    print(' schedule', sched[:80], nlu, nd)
