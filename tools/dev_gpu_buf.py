import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from circuitsimulator_amd import Netlist, Engine
from oracle import binding as orc
nl = Netlist.from_file('tests/golden/buffer.sp'); eng = Engine(nl, 0)
for B in (4096,):
    params = eng.mc_params(12345, 0.05, 0, B)
    for kern in ('scheduled',):
        eng.set_kernel(kern)
        x, it, st = eng.dc(params)
        iters = torch.zeros(B, dtype=torch.int64, device='cuda:0')
        torch.cuda.synchronize(); t = time.time()
        eng.tran(params, x, nl.tstep, 0, 300, iters, st); torch.cuda.synchronize(); dt = time.time() - t
        fb = int(((st & 0x20) != 0).sum())
        print('B', B, kern, 'time %.4f' % dt, 'rate %.3g' % (iters.sum().item() / dt), 'fallback instances', fb)
        if B == 8:
            ph = params.cpu().numpy()
            for b in (0, 7):
                o = orc.tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstop, want_rows=False)
                e = np.abs(x[:, b].cpu().numpy() - o['x_final']) / np.maximum(np.abs(o['x_final']), 1e-6)
                print('   b', b, 'iters', int(iters[b]), o['iters'], 'max rel', e.max())
