import sys, time, numpy as np
sys.path.insert(0, '.')
from circuitsimulator_amd import Netlist, Engine
from oracle import binding as orc
for name in ['buffer', 'dbmixer']:
    nl = Netlist.from_file('tests/golden/%s.sp' % name)
    eng = Engine(nl, 0)
    ns = 300 if name=='buffer' else 2000
    tstop = nl.tstep*ns
    wave, xf, itr, stt = eng.tran_host(B=1, tstop=tstop, probes=list(range(nl.n_unknowns)))
    o = orc.tran(nl.ir_ptr, nl.n_unknowns, nl.nominal_params, 0, nl.tstep, tstop)
    rows = o['rows'][:,1:]
    floor = np.where(np.arange(nl.n_unknowns) < nl.n_node_eq, 1e-6, 1e-9)
    rel = np.abs(wave[0]-rows)/np.maximum(np.abs(rows), floor)
    print(name, 'iters', itr, o['iters'], 'max rel', rel.max())
    worst = np.argsort(rel.max(axis=0))[::-1][:5]
    for c in worst:
        r = rel[:,c].argmax()
        print('  col', c, nl.eq_names[c], 'row', r, 'rel', rel[r,c], 'gpu', wave[0,r,c], 'ref', rows[r,c], 'absdiff', abs(wave[0,r,c]-rows[r,c]))
    # pure relative on |v|>1e-3
    big = np.abs(rows) > 1e-3
    print('  max pure rel on |v|>1e-3:', (np.abs(wave[0]-rows)/np.abs(rows))[big].max())
