"""Generated kernels for LINEAR circuits (BASELINE configs[3], csrc/engine/codegen_linear.cpp): factor once per
launch, substitute once per step, execute the damped passes -- with the reference's arithmetic (no contraction,
true divisions, sums in stamping order), so on the recorded pivot sequence they perform the general kernel's
operations: states and waveforms BIT FOR BIT, per-step NR counts equal."""
import os
import shutil

import numpy as np
import pytest

from conftest import rel_err
from test_gpu_parity import NOFB, TOL, _orc, _run_tran

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device: the engine has no CPU path")
    return torch


def _need_hipcc():
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available for the JIT")


def _linear_lanes(eng):
    text = eng.sched_info["text"] if eng.sched_info else ""
    return text


CIRCUITS = {
    # the configs[3] ladder at a quarter of its length: tridiagonal + the source's border row (pivoted first)
    "ladder64": lambda: __import__("circuitsimulator_amd.workloads", fromlist=["x"]).rc_ladder_netlist(64),
    # inductors (branch rows, exact +-1 incidence), a current source, PULSE and PWL waveforms, a floating-ish mesh
    "rlc_mesh": lambda: (
        "V1 a 0 PULSE(0 1 2e-9 1e-9 1e-9 5e-9 20e-9)\n"
        "I1 0 c PWL(0 0 5e-9 1e-3 30e-9 -1e-3)\n"
        "R1 a b 50\nL1 b c 2e-9\nC1 c 0 1e-12\nR2 c d 75\nL2 d e 5e-9\nC2 e 0 2e-12\nR3 e 0 1e3\n"
        "R4 b e 220\nC3 b d 0.5e-12\nV2 f 0 SIN 0.5 0.25 2e8 0\nR5 f d 330\n"
        ".TRAN 1e-10 8e-9\n"),
}


@pytest.mark.parametrize("name", sorted(CIRCUITS))
def test_linear_kernels_are_bitwise_the_general_kernel(torch_mod, tmp_path, monkeypatch, name):
    from circuitsimulator_amd import Engine, Netlist
    _need_hipcc()
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    nl = Netlist.from_text(CIRCUITS[name]())
    eng = Engine(nl, 0)
    B, steps = 37, 80                                   # ragged against 4 instances per wave and 64 lanes
    params = eng.mc_params(77, 0.05, 0, B)
    probes = sorted({0, nl.n_unknowns // 2, nl.n_unknowns - 1})
    eng.set_kernel("general")
    ref = _run_tran(torch_mod, eng, params, steps, nl.tstep, probes=probes, stride=3, want_step_iters=True, chunks=[30, 50])
    eng.set_kernel("auto")
    eng.jit_scheduled(params, plan_steps=5)
    assert eng.tran_kernel == "scheduled"
    got = _run_tran(torch_mod, eng, params, steps, nl.tstep, probes=probes, stride=3, want_step_iters=True, chunks=[30, 50])
    assert not (got["status"] & 0x27).any()
    # the operating point, same bits, signs of zeros included.  rlc_mesh: the generated direct-solve kernel
    # (dcSolveDirectLU on the recorded DC pivot sequence) did the work itself; the ladder's DC pivots differ from instance
    # to instance (a resistor chain with its capacitors open: pivots win by a fraction of a percent), so the JIT gives it
    # no DC kernel and the general kernel solves
    assert np.array_equal(got["dc_iters"], ref["dc_iters"]) and (got["dc_iters"] == 1).all()
    assert np.array_equal(got["x_dc"], ref["x_dc"]) and np.array_equal(np.signbit(got["x_dc"]), np.signbit(ref["x_dc"]))
    assert not (got["status"] & 0x80).any()
    src = open(tmp_path / "jit" / [f for f in os.listdir(tmp_path / "jit") if f.endswith(".hip")][0]).read()
    assert ("csim_dc_linear_kernel(" in src) == (name == "rlc_mesh")
    assert np.array_equal(got["step_iters"], ref["step_iters"])
    assert np.array_equal(got["status"] & NOFB, ref["status"])
    assert np.array_equal(got["x"], ref["x"]), np.abs(got["x"] - ref["x"]).max()
    assert np.array_equal(got["wave"], ref["wave"])
    # and the oracle (same operations; device sin() may differ from glibc's in the last bit)
    ph = params.cpu().numpy()
    for b in (0, B - 1):
        o = _orc().tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstep * steps, want_rows=False, want_step_iters=True)
        assert np.array_equal(got["step_iters"][:, b], o["step_iters"])
        assert rel_err(got["x"][:, b], o["x_final"]).max() < TOL
    # an asynchronous call gives the same bits
    eng.set_option("hybrid_sync", 0)
    again = _run_tran(torch_mod, eng, params, steps, nl.tstep, probes=probes, stride=3, want_step_iters=True, chunks=[30, 50])
    for key in ("x", "wave", "iters", "step_iters", "status"):
        assert np.array_equal(again[key], got[key]), key


def test_ladder_sixteen_lane_kernel_is_the_one_that_runs(torch_mod, tmp_path, monkeypatch):
    """configs[3] itself (N = 257): the library the JIT builds carries the sixteen-lanes-per-instance linear kernel
    (tape, iterate and x_raw in registers); a degenerate instance (an inductor-free ladder has none, so: a resistor of
    0 ohm, which the reference skips with a warning) still equals the general kernel."""
    from circuitsimulator_amd import Engine, Netlist
    from circuitsimulator_amd.workloads import rc_ladder_netlist
    _need_hipcc()
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    nl = Netlist.from_text(rc_ladder_netlist(256))
    eng = Engine(nl, 0)
    B = 9
    params = eng.mc_params(3, 0.05, 0, B)
    eng.jit_scheduled(params, plan_steps=5)
    jit_dir = tmp_path / "jit"
    src = [f for f in os.listdir(jit_dir) if f.endswith(".hip")]
    assert src and "csim_tran_linear16_kernel" in open(jit_dir / src[0]).read()
    eng.set_kernel("general")
    ref = _run_tran(torch_mod, eng, params, 100, nl.tstep, want_step_iters=True)
    eng.set_kernel("auto")
    got = _run_tran(torch_mod, eng, params, 100, nl.tstep, want_step_iters=True)
    assert got["iters"][0] == 1687
    assert np.array_equal(got["step_iters"], ref["step_iters"])
    assert np.array_equal(got["x"], ref["x"])
    assert np.array_equal(got["x_dc"], ref["x_dc"]) and not (got["status"] & 0x80).any()
