"""circuitsimulator_amd -- MI355X-native batched DC / transient MNA solve engine.

Drop-in for one path of ZyuRao/CircuitSimulator: stamp -> pivoted dense LU ->
damped Newton update, for a batch of independent circuit instances, behind
the C-ABI declared in include/csim.h (libcsim.so, HIP kernels for gfx950).
"""
from .capi import CsimError, lib  # noqa: F401
from .engine import Engine, Netlist, gs_solve_batch, lu_decompose_batch, lu_solve_batch  # noqa: F401

__all__ = ["Engine", "Netlist", "CsimError", "lu_solve_batch", "lu_decompose_batch", "gs_solve_batch", "lib"]
