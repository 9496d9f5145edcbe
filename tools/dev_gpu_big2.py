import sys, time, numpy as np, os
sys.path.insert(0, '.')
import torch
from circuitsimulator_amd import Netlist, Engine
from circuitsimulator_amd.workloads import rc_ladder_netlist
from oracle import binding as orc
os.environ['CSIM_JIT_DIR'] = '/tmp/csim_jit_dev'
nn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nl = Netlist.from_text(rc_ladder_netlist(nn))
eng = Engine(nl, 0)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = 100
params = eng.mc_params(12345, 0.05, 0, B)
t = time.time(); eng.jit_scheduled(params, plan_steps=5); print('jit time', time.time()-t, eng.tran_kernel)
os.system('ls -la /tmp/csim_jit_dev | tail -4; grep -c . /tmp/csim_jit_dev/*.hip; tail -3 /tmp/csim_jit_dev/*.log')
x, it, st = eng.dc(params)
iters = torch.zeros(B, dtype=torch.int64, device='cuda:0')
torch.cuda.synchronize()
t = time.time(); eng.tran(params, x, nl.tstep, 0, steps, iters, st); torch.cuda.synchronize(); dt = time.time()-t
print('tran time', dt, 'iters', iters[:4].tolist(), 'status', st[:4].tolist(), 'rate %.3g' % (iters.sum().item()/dt))
ph = params.cpu().numpy()
for b in (0, 1):
    o = orc.tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstep*steps, want_rows=False)
    e = np.abs(x[:, b].cpu().numpy() - o['x_final']) / np.maximum(np.abs(o['x_final']), 1e-6)
    print(' TRAN b', b, 'iters', int(iters[b]), o['iters'], 'max rel', e.max())
