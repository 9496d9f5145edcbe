import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from circuitsimulator_amd import Netlist, Engine, lu_solve_batch
from oracle import binding as orc
for name in ['buffer', 'dbmixer']:
    nl = Netlist.from_file('tests/golden/%s.sp' % name)
    eng = Engine(nl, 0)
    x, it, st = eng.dc_host(B=2)
    xo, ito, sto = orc.dc(nl.ir_ptr, nl.n_unknowns, nl.nominal_params)
    rel = np.abs(x[0]-xo)/np.maximum(np.abs(xo),1e-9)
    print(name, 'DC iters gpu', it, 'oracle', ito, 'status', st, hex(sto), 'max rel', rel.max(), 'inst eq', np.array_equal(x[0],x[1]))
    ns = 300 if name=='buffer' else 500
    tstop = nl.tstep*ns
    t=time.time()
    wave, xf, itr, stt = eng.tran_host(B=2, tstop=tstop, probes=list(range(nl.n_unknowns)))
    dt=time.time()-t
    o = orc.tran(nl.ir_ptr, nl.n_unknowns, nl.nominal_params, 0, nl.tstep, tstop)
    rows = o['rows'][:,1:]
    floor = np.where(np.arange(nl.n_unknowns) < nl.n_node_eq, 1e-6, 1e-9)
    rel = np.abs(wave[0]-rows)/np.maximum(np.abs(rows), floor)
    print(name, 'TRAN iters gpu', itr, 'oracle', o['iters'], 'status', stt, 'rows', wave.shape, 'max rel', rel.max(), 'time', dt)
A = np.random.RandomState(0).randn(5, 7, 7); b = np.random.RandomState(1).randn(5,7)
x, fl = lu_solve_batch(A, b)
print('lu', np.abs(x - np.linalg.solve(A, b[...,None])[...,0]).max(), fl)
