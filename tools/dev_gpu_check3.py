import sys, numpy as np
sys.path.insert(0, '.')
from circuitsimulator_amd import Netlist, Engine
from oracle import binding as orc
nl = Netlist.from_file('tests/golden/dbmixer.sp'); eng = Engine(nl, 0)
wave, xf, it, st = eng.tran_host(B=1, probes=list(range(nl.n_unknowns)))
o = orc.tran(nl.ir_ptr, nl.n_unknowns, nl.nominal_params, 0, nl.tstep, nl.tstop)
rows = o['rows'][:,1:]
floor = np.where(np.arange(nl.n_unknowns) < nl.n_node_eq, 1e-6, 1e-9)
rel = np.abs(wave[0]-rows)/np.maximum(np.abs(rows), floor)
print('iters', it, o['iters'])
for c in np.argsort(rel.max(axis=0))[::-1][:6]:
    r = rel[:,c].argmax()
    print('col', c, nl.eq_names[c], 'row', r, 'rel', rel[r,c], 'gpu', wave[0,r,c], 'ref', rows[r,c], 'abs', abs(wave[0,r,c]-rows[r,c]), 'colmax', np.abs(rows[:,c]).max(), 'max abs err col', np.abs(wave[0][:,c]-rows[:,c]).max())
print('node voltages max rel (floor 1e-6):', rel[:, :nl.n_node_eq].max())
