#!/usr/bin/env python3
"""scratch: DC pivot sequences of several Monte-Carlo instances of a linear test circuit"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from circuitsimulator_amd import Engine, Netlist
import test_linear_kernels as t
nl = Netlist.from_text(t.CIRCUITS["rlc_mesh"]())
eng = Engine(nl, 0)
B = 37
params = eng.mc_params(77, 0.05, 0, B)
for inst in (0, 12, 24, 36):
    print(inst, eng.record_dc_pivot_schedules(params, inst, max_alts=1), eng.record_dc_pivot_schedules(params, inst, max_alts=8))
