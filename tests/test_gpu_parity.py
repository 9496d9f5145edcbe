"""GPU parity: the HIP kernels behind the C-ABI against the CPU oracle on identical
parameter tables, against the golden anchors, and -- at BASELINE.json's batch sizes --
through size-independent properties (batch invariance, chunk invariance, shard invariance).

Bar: NR-iteration counts and status words EQUAL (integer trajectory fingerprint); node
voltages and branch currents within 1e-9 relative with an absolute floor of 1e-6 (V / A);
see conftest.rel_err for why the floor is what it is.
"""
import numpy as np
import pytest

from conftest import has_gpu, netlist_path, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-9
FALLBACK = 0x20          # CSIM_ST_SCHED_FALLBACK: informational (instance was re-run by the general kernel)
FALLBACK_DC = 0x80       # CSIM_ST_SCHED_FALLBACK_DC: the same for the DC operating point
FAITHFUL = 0x100         # CSIM_ST_SCHED_FAITHFUL: informational (>= 1 step ran on the faithful generated kernel)
NOFB = 0xFFFFFE5F        # mask that drops the three informational bits


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device: the engine has no CPU path")
    return torch


@pytest.fixture(scope="module")
def engines(torch_mod, buffer_nl, dbmixer_nl):
    from circuitsimulator_amd import Engine
    return {"buffer": (buffer_nl, Engine(buffer_nl, 0)), "dbmixer": (dbmixer_nl, Engine(dbmixer_nl, 0))}


def _orc():
    from oracle import binding
    return binding


def _run_tran(torch, eng, params, n_steps, tstep, probes=None, stride=1, want_step_iters=False, chunks=None):
    """DC + n_steps through the device-pointer ABI. Returns dict of numpy results."""
    B = params.shape[1]
    x, dc_it, st = eng.dc(params)
    x_dc = x.clone()
    iters = torch.zeros(B, dtype=torch.int64, device=x.device)
    wave = None
    if probes is not None:
        rows = n_steps // stride + 1
        wave = torch.zeros((rows, len(probes), B), dtype=torch.float64, device=x.device)
    si = torch.zeros((n_steps, B), dtype=torch.int32, device=x.device) if want_step_iters else None
    done = 0
    for n in (chunks or [n_steps]):
        eng.tran(params, x, tstep, done, n, iters, st, probes=probes, out_stride=stride, wave=wave,
                 step_iters=si[done:done + n] if si is not None else None)
        done += n
    assert done == n_steps
    torch.cuda.synchronize()
    return dict(x=x.cpu().numpy(), x_dc=x_dc.cpu().numpy(), dc_iters=dc_it.cpu().numpy(),
                iters=iters.cpu().numpy(), status=st.cpu().numpy().astype(np.uint32),
                wave=wave.cpu().numpy() if wave is not None else None,
                step_iters=si.cpu().numpy() if si is not None else None)


# ------------------------------------------------------------------ DC (K2)

@pytest.mark.parametrize("name", ["buffer", "dbmixer"])
def test_dc_nominal_vs_anchors_and_oracle(name, engines, anchors):
    nl, eng = engines[name]
    a = anchors[name]
    x, it, st = eng.dc_host(B=3)
    xo, ito, sto = _orc().dc(nl.ir_ptr, nl.n_unknowns, nl.nominal_params)
    assert list(it) == [a["dc_iters"]] * 3 == [ito] * 3
    assert list(st) == [sto] * 3
    assert np.array_equal(x[0], x[1]) and np.array_equal(x[0], x[2])
    assert rel_err(x[0], xo).max() < TOL
    for eqname, s in a["dc_x"].items():           # golden 17-digit anchors
        i = nl.eq_names.index(eqname)
        floor = 1e-6
        assert abs(x[0][i] - float(s)) <= TOL * max(abs(float(s)), floor), eqname


@pytest.mark.parametrize("name", ["buffer", "dbmixer"])
def test_dc_mc_batch_vs_oracle(name, engines, torch_mod):
    nl, eng = engines[name]
    B = 64
    params = eng.mc_params(12345, 0.05, 0, B)
    x, it, st = eng.dc(params)
    torch_mod.cuda.synchronize()
    ph = params.cpu().numpy()
    x, it, st = x.cpu().numpy(), it.cpu().numpy(), st.cpu().numpy().astype(np.uint32)
    for b in (0, 1, 2, 3, 17, B - 1):
        xo, ito, sto = _orc().dc(nl.ir_ptr, nl.n_unknowns, ph, b)
        assert it[b] == ito and st[b] == sto, (b, it[b], ito, st[b], sto)
        assert rel_err(x[:, b], xo).max() < TOL, b


def test_mc_params_device_equals_host_bitwise(engines, torch_mod):
    nl, eng = engines["dbmixer"]
    dev = eng.mc_params(12345, 0.05, 1000, 513).cpu().numpy()
    host = nl.mc_params_host(12345, 0.05, 1000, 513)
    assert np.array_equal(dev, host)
    assert np.array_equal(eng.mc_params(12345, 0.05, 0, 4).cpu().numpy()[:, 0], nl.nominal_params)


# ----------------------------------------------------------- transient (K1)

def test_buffer_transient_as_shipped_full_waveform(engines, torch_mod, anchors):
    nl, eng = engines["buffer"]
    params = eng.upload_params(nl.nominal_table(2))
    r = _run_tran(torch_mod, eng, params, 300, nl.tstep, probes=list(range(nl.n_unknowns)), want_step_iters=True)
    o = _orc().tran(nl.ir_ptr, nl.n_unknowns, nl.nominal_params, 0, nl.tstep, nl.tstop, want_step_iters=True)
    assert r["iters"][0] == anchors["buffer"]["tran_iters"] == o["iters"]
    assert np.array_equal(r["step_iters"][:, 0], o["step_iters"])
    assert (r["status"][0] & NOFB) == o["status"]
    wave = np.transpose(r["wave"], (2, 0, 1))          # [B][rows][N]
    assert np.array_equal(wave[0], wave[1])
    assert rel_err(wave[0], o["rows"][:, 1:]).max() < TOL
    for eqname, s in anchors["buffer"]["last_row"].items():
        i = nl.eq_names.index(eqname)
        floor = 1e-6
        assert abs(wave[0][-1][i] - float(s)) <= TOL * max(abs(float(s)), floor)


def test_buffer_10k_steps_batch1_config(engines, torch_mod, anchors):
    """BASELINE.json configs[1]: buffer.sp transient 10 000 steps, batch = 1 (.TRAN 3e-11 300e-9)."""
    nl, eng = engines["buffer"]
    tstep, tstop = 3e-11, 300e-9
    assert nl.num_steps(tstep, tstop) == 10000
    params = eng.upload_params(nl.nominal_table(1))
    r = _run_tran(torch_mod, eng, params, 10000, tstep, probes=list(range(nl.n_unknowns)), stride=1,
                  want_step_iters=True, chunks=[4096, 4096, 1808])
    o = _orc().tran(nl.ir_ptr, nl.n_unknowns, nl.nominal_params, 0, tstep, tstop, want_step_iters=True)
    assert r["iters"][0] == anchors["buffer"]["tran10k_iters"] == o["iters"]
    assert np.array_equal(r["step_iters"][:, 0], o["step_iters"])
    assert rel_err(r["wave"][:, :, 0], o["rows"][:, 1:]).max() < TOL


@pytest.mark.parametrize("lanes", [1, 4, 16])
def test_dbmixer_mc_transient_vs_oracle(engines, torch_mod, lanes):
    """The generated transient kernels (one lane per instance; four and sixteen lanes per instance) against the oracle."""
    nl, eng = engines["dbmixer"]
    B, steps = 64, 1500
    params = eng.mc_params(12345, 0.05, 0, B)
    eng.set_option("lanes_per_instance", lanes)
    try:
        assert eng.lanes_for_batch(B) == lanes
        r = _run_tran(torch_mod, eng, params, steps, nl.tstep, probes=nl.probes, stride=100, want_step_iters=True)
    finally:
        eng.set_option("lanes_per_instance", 0)
    ph = params.cpu().numpy()
    for b in (0, 1, 2, 3, B - 1):
        o = _orc().tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstep * steps, want_step_iters=True)
        assert r["iters"][b] == o["iters"], b
        assert np.array_equal(r["step_iters"][:, b], o["step_iters"]), b
        assert (r["status"][b] & NOFB) == o["status"], b
        assert rel_err(r["x"][:, b], o["x_final"]).max() < TOL, b
        ref = o["rows"][::100, 1:][:, nl.probes]
        assert np.abs(r["wave"][:, :, b] - ref).max() <= TOL * np.abs(ref).max()


def test_dbmixer_full_run_nominal(engines, torch_mod, anchors):
    """tests/dbmixer.sp as shipped: 50 000 steps; totals and every row against the oracle."""
    nl, eng = engines["dbmixer"]
    a = anchors["dbmixer"]
    wave, xf, it, st = eng.tran_host(B=1, probes=list(range(nl.n_unknowns)))
    assert it[0] == a["tran_iters"] and st[0] == 0
    o = _orc().tran(nl.ir_ptr, nl.n_unknowns, nl.nominal_params, 0, nl.tstep, nl.tstop)
    assert wave.shape == (1, 50001, 31)
    assert rel_err(wave[0], o["rows"][:, 1:]).max() < TOL
    for eqname, s in a["last_row"].items():
        i = nl.eq_names.index(eqname)
        floor = 1e-6
        assert abs(xf[0][i] - float(s)) <= TOL * max(abs(float(s)), floor)


def test_dbmixer_full_run_monte_carlo_instances(engines, torch_mod):
    """BASELINE configs[2]/[4] at FULL length: Monte-Carlo instances through all 50 000 steps on the
    scheduled kernel; per-instance NR totals and the final state against the oracle (three instances;
    the oracle needs ~3 s each)."""
    torch = torch_mod
    nl, eng = engines["dbmixer"]
    B = 64
    params = eng.mc_params(12345, 0.05, 0, B)
    x, dc_it, st = eng.dc(params)
    iters = torch.zeros(B, dtype=torch.int64, device=x.device)
    for s0 in range(0, 50000, 5000):
        eng.tran(params, x, nl.tstep, s0, 5000, iters, st)
    torch.cuda.synchronize()
    assert not (st.cpu().numpy().astype(np.uint32) & 0x27).any()
    ph = params.cpu().numpy()
    xs, its = x.cpu().numpy(), iters.cpu().numpy()
    for b in (1, 37, 63):
        o = _orc().tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstop, want_rows=False)
        assert o["n_steps"] == 50000 and its[b] == o["iters"], (b, its[b], o["iters"])
        assert rel_err(xs[:, b], o["x_final"]).max() < TOL


# --------------------------------------- the generated kernel with the reference's arithmetic

def test_faithful_generated_kernel_is_bitwise_the_general_kernel(engines, torch_mod):
    """set_kernel("faithful"): the lane-per-instance generated kernel emitted with true divisions, without
    FMA contraction and without the slow-step rule.  On recorded pivot sequences it performs the reference's
    operations in the reference's order, so it must reproduce the general kernel (itself bit-identical to the
    oracle's LU) BIT FOR BIT -- state, NR counts, status -- on dbmixer and, through its alternatives and the
    hand-over for what they do not cover, on the switching buffer."""
    for name, B, steps in (("dbmixer", 96, 300), ("buffer", 64, 200)):
        nl, eng = engines[name]
        params = eng.mc_params(321, 0.05, 0, B)
        eng.set_kernel("general")
        slow = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True, probes=[0, 1], stride=7)
        eng.set_kernel("faithful")
        assert eng.tran_kernel == "faithful"
        try:
            fast = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True, probes=[0, 1], stride=7,
                             chunks=[steps // 3, steps - steps // 3])
        finally:
            eng.set_kernel("auto")
        assert np.array_equal(fast["step_iters"], slow["step_iters"]), name
        assert np.array_equal(fast["status"] & NOFB, slow["status"]), name
        assert np.array_equal(fast["x"], slow["x"]), name                      # bitwise
        assert np.array_equal(fast["wave"], slow["wave"]), name
        assert ((fast["status"] & FAITHFUL) != 0).all(), name
        if name == "dbmixer":
            assert not (fast["status"] & FALLBACK).any()                       # one sequence: nothing reached the general kernel


# --------------------------------------- sixteen lanes per instance (group kernel)

def test_group_kernel_ragged_batches_chunks_and_probes(engines, torch_mod):
    """The sixteen-lanes-per-instance kernel packs 4 instances per wavefront: batch sizes that do not
    fill the last wave, launches cut into chunks, probe rows -- all must equal the lane-per-instance
    kernel (same NR counts per step, states within 1e-9) and, for the first and last instance, the oracle."""
    nl, eng = engines["dbmixer"]
    assert "group16" in eng.sched_info["text"] and "group4" in eng.sched_info["text"]
    for B in (1, 3, 5, 17, 67, 256):
        params = eng.mc_params(777, 0.05, 0, B)
        eng.set_option("lanes_per_instance", 1)
        ref = _run_tran(torch_mod, eng, params, 240, nl.tstep, probes=[1, 2, 30], stride=40, want_step_iters=True)
        ph = params.cpu().numpy()
        orc = {b: _orc().tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstep * 240, want_step_iters=True) for b in {0, B - 1}}
        for lanes in (16, 4):            # 4 resp. 16 instances per wavefront
            eng.set_option("lanes_per_instance", lanes)
            try:
                got = _run_tran(torch_mod, eng, params, 240, nl.tstep, probes=[1, 2, 30], stride=40, want_step_iters=True,
                                chunks=[1, 99, 140])
            finally:
                eng.set_option("lanes_per_instance", 0)
            assert np.array_equal(got["step_iters"], ref["step_iters"]), (B, lanes)
            assert np.array_equal(got["status"], ref["status"]) and not (got["status"] & 0x27).any(), (B, lanes)
            assert rel_err(got["x"].T, ref["x"].T).max() < TOL, (B, lanes)
            assert rel_err(got["wave"].reshape(-1, B).T, ref["wave"].reshape(-1, B).T).max() < TOL, (B, lanes)
            for b, o in orc.items():
                assert np.array_equal(got["step_iters"][:, b], o["step_iters"]), (B, b, lanes)
                assert rel_err(got["x"][:, b], o["x_final"]).max() < TOL, (B, b, lanes)


def _amplifier_line(stages):
    """Resistively loaded NMOS stages, RC coupled and DC biased: stages MOSFETs, 2*stages + 3 nodes; a
    well-behaved circuit (5-13 Newton passes per step)."""
    t = ["* amplifier line", "VDD vdd 0 DC 2.5", "Vin in 0 SIN 0.9 0.05 200e6 0", "Rg in g0 100"]
    for k in range(stages):
        t += ["MN%d d%d g%d 0 n 4e-6 1e-6 2" % (k, k, k), "RD%d vdd d%d %g" % (k, k, 4000 + 100 * k),
              "RC%d d%d g%d %g" % (k, k, k + 1, 3000 + 50 * k), "RB%d g%d 0 %g" % (k, k + 1, 6000 + 100 * k),
              "CG%d g%d 0 %ge-15" % (k, k + 1, 10 + k)]
    t += [".MODEL 2 VT 0.55 MU 3e-2 COX 2e-3 LAMBDA 0.04 CJ0 1e-14", ".TRAN 5e-12 2e-9", ".plotnv d%d" % (stages - 1)]
    return "\n".join(t) + "\n"


def test_group_kernel_three_slots_and_two_mosfet_rounds(torch_mod, tmp_path, monkeypatch):
    """The sixteen-lanes-per-instance kernel beyond the shipped netlists' shape: 33 < N <= 48 unknowns (three
    rows per lane) and more than 16 MOSFETs (two evaluation rounds per Newton pass), JIT-generated."""
    from circuitsimulator_amd import Engine, Netlist
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    nl = Netlist.from_text(_amplifier_line(17))             # 39 unknowns, 17 MOSFETs
    assert 32 < nl.n_unknowns <= 48 and nl.n_elems > 80
    eng = Engine(nl, 0)
    B, steps = 40, 250
    params = eng.mc_params(11, 0.03, 0, B)
    slow = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
    eng.jit_scheduled(params, plan_steps=steps)
    assert eng.tran_kernel == "scheduled" and "group16" in eng.sched_info["text"]
    res = {}
    for lanes in (1, 16):
        eng.set_option("lanes_per_instance", lanes)
        res[lanes] = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
        assert np.array_equal(res[lanes]["step_iters"], slow["step_iters"]), lanes
        assert np.array_equal(res[lanes]["status"] & NOFB, slow["status"]), lanes
        assert rel_err(res[lanes]["x"].T, slow["x"].T).max() < TOL, lanes
    eng.set_option("lanes_per_instance", 0)
    assert ((res[16]["status"] & (FALLBACK | FAITHFUL)) == 0).sum() > B // 2      # the group kernel itself did the work
    o = _orc().tran(nl.ir_ptr, nl.n_unknowns, params.cpu().numpy(), 7, nl.tstep, nl.tstep * steps, want_rows=False)
    assert res[16]["iters"][7] == o["iters"] and rel_err(res[16]["x"][:, 7], o["x_final"]).max() < TOL


def test_group_kernel_four_slots_55_unknowns(torch_mod, tmp_path, monkeypatch):
    """49 <= N <= 64: four rows per lane.  A 55-unknown amplifier line (25 MOSFETs: two evaluation rounds) through the
    sixteen-lanes-per-instance kernel -- until round 3 such circuits had the wave-per-instance LDS kernel only -- against the
    general kernel (per-step NR counts, status, states) and the oracle."""
    from circuitsimulator_amd import Engine, Netlist
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    nl = Netlist.from_text(_amplifier_line(25))
    assert nl.n_unknowns == 55
    eng = Engine(nl, 0)
    B, steps = 24, 200
    params = eng.mc_params(5, 0.03, 0, B)
    slow = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
    eng.jit_scheduled(params, plan_steps=steps)
    assert eng.tran_kernel == "scheduled" and "group16" in eng.sched_info["text"]
    eng.set_option("lanes_per_instance", 16)
    import time
    t0 = time.perf_counter()
    fast = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
    dt = time.perf_counter() - t0
    print("55 unknowns, 16 lanes per instance: %d instances x %d steps in %.3f s" % (B, steps, dt))
    assert np.array_equal(fast["step_iters"], slow["step_iters"])
    assert np.array_equal(fast["status"] & NOFB, slow["status"])
    assert rel_err(fast["x"].T, slow["x"].T).max() < TOL
    assert ((fast["status"] & (FALLBACK | FAITHFUL)) == 0).sum() > B // 2        # the group kernel itself did the work
    o = _orc().tran(nl.ir_ptr, nl.n_unknowns, params.cpu().numpy(), 3, nl.tstep, nl.tstep * steps, want_rows=False)
    assert fast["iters"][3] == o["iters"] and rel_err(fast["x"][:, 3], o["x_final"]).max() < TOL


@pytest.mark.parametrize("stages", [30, 45])
def test_group_kernel_five_and_six_slots(torch_mod, tmp_path, monkeypatch, stages):
    """65 <= N <= 96: five and six rows per lane (amplifier lines of 30 / 45 stages: N = 65 / 95).  From five rows on the
    sixteen-lane kernel takes the small LDS image of the four-lane kernel (sources' parameters only, compact per-step
    terms, term table sharing its place with the staging rows) and spills registers; it is still the fastest kernel for
    such circuits up to mid-size batches.  Per-step NR counts, status and states against the general kernel, one instance
    against the oracle."""
    from circuitsimulator_amd import Engine, Netlist
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    nl = Netlist.from_text(_amplifier_line(stages))
    assert nl.n_unknowns == 2 * stages + 5 and 64 < nl.n_unknowns <= 96
    eng = Engine(nl, 0)
    B, steps = 24, 120
    params = eng.mc_params(5, 0.03, 0, B)
    slow = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
    eng.jit_scheduled(params, plan_steps=steps)
    assert eng.tran_kernel == "scheduled" and "group16" in eng.sched_info["text"] and "group4" not in eng.sched_info["text"]
    assert eng.lanes_for_batch(B) == 16
    fast = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True, chunks=[50, 70])
    assert np.array_equal(fast["step_iters"], slow["step_iters"])
    assert np.array_equal(fast["status"] & NOFB, slow["status"])
    assert rel_err(fast["x"].T, slow["x"].T).max() < TOL
    assert ((fast["status"] & (FALLBACK | FAITHFUL)) == 0).sum() > B // 2        # the group kernel itself did the work
    o = _orc().tran(nl.ir_ptr, nl.n_unknowns, params.cpu().numpy(), 3, nl.tstep, nl.tstep * steps, want_rows=False)
    assert fast["iters"][3] == o["iters"] and rel_err(fast["x"][:, 3], o["x_final"]).max() < TOL


def test_group_kernel_carries_every_recorded_schedule(engines, torch_mod):
    """buffer.sp at its shipped step alternates between ten pivot schedules.  The group kernel has one
    solve body per schedule, all over the first schedule's row placement (pivot rows at arbitrary lanes,
    explicit lane masks): it must flag no more instances than the lane-per-instance kernel, and agree with
    the general kernel on every NR count."""
    nl, eng = engines["buffer"]
    B = 96
    params = eng.mc_params(4, 0.05, 0, B)
    eng.set_kernel("general")
    slow = _run_tran(torch_mod, eng, params, 300, nl.tstep, want_step_iters=True)
    eng.set_kernel("auto")
    assert eng.lanes_for_batch(B) == 16
    runs = {}
    for lanes in (1, 4, 16):
        eng.set_option("lanes_per_instance", lanes)
        try:
            runs[lanes] = _run_tran(torch_mod, eng, params, 300, nl.tstep, want_step_iters=True, chunks=[150, 150])
        finally:
            eng.set_option("lanes_per_instance", 0)
        assert np.array_equal(runs[lanes]["step_iters"], slow["step_iters"])
        assert np.array_equal(runs[lanes]["status"] & NOFB, slow["status"])
        assert rel_err(runs[lanes]["x"].T, slow["x"].T).max() < TOL
    flagged = {lanes: int(((r["status"] & FALLBACK) != 0).sum()) for lanes, r in runs.items()}
    assert flagged[16] <= flagged[1] and flagged[4] <= flagged[1], flagged
    assert flagged[16] < B // 4 and flagged[4] < B // 4, flagged          # the alternatives are in use, not handed over
    # one instance against the oracle at a step where the first schedule holds throughout
    eng.set_option("lanes_per_instance", 16)
    try:
        r = _run_tran(torch_mod, eng, params[:, :9].contiguous(), 1000, 3e-11)
    finally:
        eng.set_option("lanes_per_instance", 0)
    assert not (r["status"] & 0x27).any()
    o = _orc().tran(nl.ir_ptr, nl.n_unknowns, params.cpu().numpy(), 8, 3e-11, 3e-11 * 1000, want_rows=False)
    assert r["iters"][8] == o["iters"] and rel_err(r["x"][:, 8], o["x_final"]).max() < TOL


def test_refining_schedules_from_flagged_instances(torch_mod, tmp_path, monkeypatch):
    """Instances whose factorisations use pivot sequences the generated kernels do not carry are handed to the
    general kernel and finish there, a few waves alone on the chip (buffer.sp with the seven sequences of its
    nominal run: 31 of 4 096 Monte-Carlo instances, most of the wall time).  Engine.refine_schedules replays some
    of them through the planner, appends what it finds and re-specialises (generator + hipcc + load): fewer
    instances leave the fast kernels, the results do not move (every factorisation verifies the sequence it uses).
    Here the kernels are first rebuilt with the three most frequent shipped sequences only."""
    import shutil
    from circuitsimulator_amd import Engine, Netlist
    if not (shutil.which("hipcc") or __import__("os").path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available for the JIT")
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    nl = Netlist.from_file(netlist_path("buffer.sp"))
    eng = Engine(nl, 0)
    shipped, shipped_dc = eng.loaded_schedules()
    assert len(shipped) == 10 and not shipped_dc
    eng.jit_with_schedules(shipped[:3])
    assert eng.loaded_schedules()[0] == shipped[:3]
    B, steps = 1024, 300
    params = eng.mc_params(12345, 0.05, 0, B)
    before = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
    n_before = int(((before["status"] & FALLBACK) != 0).sum())
    assert n_before > 8
    added = eng.refine_schedules(params, before["status"], nl.tstep, n_steps=steps)
    assert added > 0
    assert eng.loaded_schedules()[0][:3] == shipped[:3] and len(eng.loaded_schedules()[0]) == 3 + added
    after = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
    n_after = int(((after["status"] & FALLBACK) != 0).sum())
    assert n_after < n_before // 2, (n_before, n_after)
    assert np.array_equal(after["step_iters"], before["step_iters"])
    assert np.array_equal(after["status"] & NOFB, before["status"] & NOFB)
    assert rel_err(after["x"].T, before["x"].T).max() < TOL
    eng.close()


# --------------------------------------- scheduled (lane-per-instance) kernels

def test_shipped_netlists_have_scheduled_kernels(engines):
    for name in ("buffer", "dbmixer"):
        nl, eng = engines[name]
        assert eng.tran_kernel == "scheduled", name
        # 16 lanes per instance up to one round of the chip (4096), four up to ITS one round (16 384), one beyond
        assert [eng.lanes_for_batch(b) for b in (1, 4096, 4097, 16384, 16385, 65536)] == [16, 16, 4, 4, 1, 1]
        assert len(eng.loaded_schedules()[0]) == {"buffer": 10, "dbmixer": 1}[name]


def test_scheduled_equals_general_on_mc_batch(engines, torch_mod):
    """The generated kernel against the dynamic-pivoting general kernel on the same table:
    equal NR counts per step, states within the parity bar; dbmixer needs no fallback."""
    nl, eng = engines["dbmixer"]
    B, steps = 256, 400
    params = eng.mc_params(777, 0.05, 0, B)
    eng.set_kernel("scheduled")
    fast = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
    eng.set_kernel("general")
    slow = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
    eng.set_kernel("auto")
    assert np.array_equal(fast["step_iters"], slow["step_iters"])
    assert np.array_equal(fast["iters"], slow["iters"])
    assert not (fast["status"] & FALLBACK).any()
    assert np.array_equal(fast["status"], slow["status"])
    assert rel_err(fast["x"].T, slow["x"].T).max() < TOL


def test_switching_circuit_alternatives_and_hybrid_stepping(engines, torch_mod):
    """buffer.sp at its shipped 1 ns step switches hard and alternates between several pivot
    sequences (7 recorded alternatives in schedules/buffer.sched).  Every factorisation verifies
    the alternative it uses; a factorisation no alternative fits makes the scheduled kernel
    checkpoint the step and hand the instance to the general kernel, which returns it once a
    whole step ran on recorded sequences again.  Whatever the path, results must match."""
    nl, eng = engines["buffer"]
    B = 512
    params = eng.mc_params(12345, 0.05, 0, B)
    eng.set_kernel("general")
    slow = _run_tran(torch_mod, eng, params, 300, nl.tstep, want_step_iters=True)
    eng.set_kernel("auto")
    fast = _run_tran(torch_mod, eng, params, 300, nl.tstep, want_step_iters=True, chunks=[120, 180])
    assert np.array_equal(fast["step_iters"], slow["step_iters"])
    assert np.array_equal(fast["iters"], slow["iters"])
    assert np.array_equal(fast["status"] & NOFB, slow["status"])
    assert rel_err(fast["x"].T, slow["x"].T).max() < TOL
    n_fb = int(((fast["status"] & FALLBACK) != 0).sum())
    assert n_fb < B // 4                      # the recorded alternatives cover almost every instance
    ph = params.cpu().numpy()
    flagged = np.where((fast["status"] & FALLBACK) != 0)[0]
    for b in [0, 3, B - 1] + list(flagged[:2]):
        o = _orc().tran(nl.ir_ptr, nl.n_unknowns, ph, int(b), nl.tstep, nl.tstop, want_rows=False, want_step_iters=True)
        assert fast["iters"][b] == o["iters"] and np.array_equal(fast["step_iters"][:, b], o["step_iters"])
        assert rel_err(fast["x"][:, b], o["x_final"]).max() < TOL
    # at 3e-11 s the first alternative holds for every factorisation: no hand-over at all
    r = _run_tran(torch_mod, eng, params[:, :8].contiguous(), 2000, 3e-11)
    assert not (r["status"] & FALLBACK).any()
    o = _orc().tran(nl.ir_ptr, nl.n_unknowns, ph, 1, 3e-11, 3e-11 * 2000, want_rows=False)
    assert r["iters"][1] == o["iters"]
    assert rel_err(r["x"][:, 1], o["x_final"]).max() < TOL


def test_hybrid_stepping_when_no_alternative_fits(torch_mod, tmp_path, monkeypatch):
    """Force the hand-over path: JIT a kernel from a plan that saw only the quiet start of a
    switching circuit, then run through the switching; flagged instances must still be exact."""
    import shutil, os
    from circuitsimulator_amd import Engine, Netlist
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available for the JIT")
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    text = open(netlist_path("buffer.sp")).read().replace("Rin 101 102 10", "Rin 101 102 11")   # new topology hash? no: new constants only
    nl = Netlist.from_text(text + "\nRextra 118 0 1e9\n")          # one more element: no prebuilt kernel
    eng = Engine(nl, 0)
    assert eng.tran_kernel == "general"
    B = 64
    params = eng.mc_params(5, 0.05, 0, B)
    slow = _run_tran(torch_mod, eng, params, 300, nl.tstep, want_step_iters=True)
    eng.jit_scheduled(params, plan_steps=3)                          # 3 quiet steps: one sequence only
    assert eng.tran_kernel == "scheduled"
    for rounds in ("4", "1", "0"):          # 0 rounds: straight from the first violation to the final launch
        eng.set_option("hybrid_rounds", rounds)
        fast = _run_tran(torch_mod, eng, params, 300, nl.tstep, want_step_iters=True)
        assert ((fast["status"] & FALLBACK) != 0).any(), rounds      # the hand-over really happened
        assert np.array_equal(fast["step_iters"], slow["step_iters"]), rounds
        assert np.array_equal(fast["iters"], slow["iters"]), rounds
        assert np.array_equal(fast["status"] & NOFB, slow["status"]), rounds
        assert rel_err(fast["x"].T, slow["x"].T).max() < TOL, rounds


def test_scheduled_kernel_ragged_batch(engines, torch_mod):
    """Batch sizes that do not fill the last wavefront (lane-per-instance packs 64 per wave)."""
    nl, eng = engines["dbmixer"]
    ref = None
    for B in (1, 63, 65, 130):
        params = eng.mc_params(12345, 0.05, 0, B)
        r = _run_tran(torch_mod, eng, params, 50, nl.tstep)
        if ref is None:
            ref = r
        assert np.array_equal(r["x"][:, 0], ref["x"][:, 0]) and r["iters"][0] == ref["iters"][0]
        assert not (r["status"] & 0x27).any()


# ------------------------------------- full-size, size-independent properties

def test_batch4096_invariances(engines, torch_mod):
    """BASELINE.json configs[2] size (dbmixer, B = 4096): results of an instance do not depend
    on the batch it is in, on how the time axis is chunked into launches, or on the shard."""
    torch = torch_mod
    nl, eng = engines["dbmixer"]
    B, steps = 4096, 60
    params = eng.mc_params(12345, 0.05, 0, B)
    # make column 7 a duplicate of column 4000: equal inputs -> bitwise equal outputs
    params[:, 7] = params[:, 4000]
    whole = _run_tran(torch, eng, params, steps, nl.tstep)
    assert not (whole["status"] & 0x27).any()
    assert np.array_equal(whole["x"][:, 7], whole["x"][:, 4000]) and whole["iters"][7] == whole["iters"][4000]
    # chunked time axis == single launch, bitwise
    chunked = _run_tran(torch, eng, params, steps, nl.tstep, chunks=[1, 29, 30])
    assert np.array_equal(chunked["x"], whole["x"]) and np.array_equal(chunked["iters"], whole["iters"])
    # two shards (as two ranks would run them) == whole batch, bitwise
    lo = _run_tran(torch, eng, params[:, :2048].contiguous(), steps, nl.tstep)
    hi = _run_tran(torch, eng, params[:, 2048:].contiguous(), steps, nl.tstep)
    assert np.array_equal(np.concatenate([lo["x"], hi["x"]], axis=1), whole["x"])
    assert np.array_equal(np.concatenate([lo["iters"], hi["iters"]]), whole["iters"])
    # instance 0 is the nominal circuit: equals a batch-of-one run of the netlist
    one = _run_tran(torch, eng, eng.upload_params(nl.nominal_table(1)), steps, nl.tstep)
    assert np.array_equal(one["x"][:, 0], whole["x"][:, 0]) and one["iters"][0] == whole["iters"][0]
    # spot-check against the oracle at this size
    ph = params[:, [0, 1, 4095]].cpu().numpy()
    for j, b in enumerate((0, 1, 4095)):
        o = _orc().tran(nl.ir_ptr, nl.n_unknowns, ph, j, nl.tstep, nl.tstep * steps, want_rows=False)
        assert whole["iters"][b] == o["iters"]
        assert rel_err(whole["x"][:, b], o["x_final"]).max() < TOL


def test_mid_batch_invariances(engines, torch_mod):
    """12 288 dbmixer instances: the batch size at which the engine takes the four-lanes-per-instance kernel by itself
    (16 instances per wavefront).  Same properties as at 4096: an instance's result does not depend on its neighbours,
    on the chunking of the time axis or on the shard (6 144 each: the same kernel), and equals the oracle's."""
    torch = torch_mod
    nl, eng = engines["dbmixer"]
    B, steps = 12288, 40
    assert eng.lanes_for_batch(B) == 4 and eng.lanes_for_batch(B // 2) == 4
    params = eng.mc_params(2024, 0.05, 0, B)
    params[:, 13] = params[:, 12001]
    whole = _run_tran(torch, eng, params, steps, nl.tstep)
    assert not (whole["status"] & 0x27).any()
    assert np.array_equal(whole["x"][:, 13], whole["x"][:, 12001]) and whole["iters"][13] == whole["iters"][12001]
    chunked = _run_tran(torch, eng, params, steps, nl.tstep, chunks=[1, 19, 20])
    assert np.array_equal(chunked["x"], whole["x"]) and np.array_equal(chunked["iters"], whole["iters"])
    lo = _run_tran(torch, eng, params[:, :B // 2].contiguous(), steps, nl.tstep)
    hi = _run_tran(torch, eng, params[:, B // 2:].contiguous(), steps, nl.tstep)
    assert np.array_equal(np.concatenate([lo["x"], hi["x"]], axis=1), whole["x"])
    assert np.array_equal(np.concatenate([lo["iters"], hi["iters"]]), whole["iters"])
    # the same instances on the other fast kernels: equal NR totals, states within the bar
    for lanes in (16, 1):
        eng.set_option("lanes_per_instance", lanes)
        try:
            other = _run_tran(torch, eng, params, steps, nl.tstep)
        finally:
            eng.set_option("lanes_per_instance", 0)
        assert np.array_equal(other["iters"], whole["iters"]), lanes
        assert rel_err(other["x"].T, whole["x"].T).max() < TOL, lanes
    cols = (0, 6143, 6144, B - 1)
    ph = params[:, list(cols)].cpu().numpy()
    for j, b in enumerate(cols):
        o = _orc().tran(nl.ir_ptr, nl.n_unknowns, ph, j, nl.tstep, nl.tstep * steps, want_rows=False)
        assert whole["iters"][b] == o["iters"], b
        assert rel_err(whole["x"][:, b], o["x_final"]).max() < TOL, b


# ------------------------------------------------ host-pointer API, edge cases

def test_host_api_tstart_and_stride(engines):
    nl, eng = engines["buffer"]
    wave, xf, it, st = eng.tran_host(B=2, tstep=1e-9, tstop=20e-9, tstart=5e-9, probes=[1, 8], out_stride=2)
    o = _orc().tran(nl.ir_ptr, nl.n_unknowns, nl.nominal_params, 0, 1e-9, 20e-9)
    keep = [r for r in range(0, 21, 2) if (0.0 if r == 0 else r * 1e-9) >= 5e-9]
    assert wave.shape == (2, len(keep), 2)
    ref = o["rows"][keep][:, [2, 9]]
    assert np.abs(wave[0] - ref).max() <= TOL * np.abs(ref).max()
    assert it[0] == o["iters"]


def test_host_api_instance_major_params(engines):
    nl, eng = engines["dbmixer"]
    tab = nl.mc_params_host(12345, 0.05, 0, 5)             # [P][B]
    x, it, st = eng.dc_host(np.ascontiguousarray(tab.T))   # [B][P]
    for b in range(5):
        xo, ito, sto = _orc().dc(nl.ir_ptr, nl.n_unknowns, tab, b)
        assert it[b] == ito and st[b] == sto
        assert rel_err(x[b], xo).max() < TOL


def test_error_codes(engines):
    from circuitsimulator_amd import capi, CsimError, Engine, Netlist
    nl, eng = engines["buffer"]
    with pytest.raises(CsimError) as e:
        eng.tran_host(B=1, tstep=0.0, tstop=1e-9)
    assert e.value.code == capi.CSIM_ERR_CONFIG          # tanalisis.cpp:94-97
    with pytest.raises(CsimError) as e:
        Engine(Netlist.from_text("* empty\n"), 0)
    assert e.value.code == capi.CSIM_ERR_EMPTY           # tanalisis.cpp:103-107
    with pytest.raises(CsimError) as e:
        Engine(nl, 99)
    assert e.value.code == capi.CSIM_ERR_NO_DEVICE
    big = "V1 n0 0 1\n" + "".join("R%d n%d n%d 1\n" % (i, i, i + 1) for i in range(1030)) + "R9999 n1030 0 1\n"
    with pytest.raises(CsimError) as e:
        Engine(Netlist.from_text(big), 0)
    assert e.value.code == capi.CSIM_ERR_UNSUPPORTED
    x, it, st = eng.dc_host(np.zeros((0, nl.n_params)))  # empty batch is a no-op
    assert x.shape == (0, nl.n_unknowns)


def test_linear_circuit_direct_dc_and_rc_transient(torch_mod):
    """Linear circuits take the direct DC path: one solve, no gmin (dcanalysis.cpp:46-68)."""
    from circuitsimulator_amd import Engine, Netlist
    text = ("V1 n1 0 SIN 0 1 1e6 0\n" + "".join("R%d n%d n%d 100\n" % (k, k, k + 1) for k in range(1, 12)) +
            "".join("C%d n%d 0 1e-12\n" % (k, k) for k in range(2, 13)) + ".TRAN 1e-9 100e-9\n")
    nl = Netlist.from_text(text)
    eng = Engine(nl, 0)
    x, it, st = eng.dc_host(B=2)
    xo, ito, sto = _orc().dc(nl.ir_ptr, nl.n_unknowns, nl.nominal_params)
    assert it[0] == ito == 1 and st[0] == sto
    assert rel_err(x[0], xo).max() < TOL
    wave, xf, itr, stt = eng.tran_host(B=2, probes=list(range(nl.n_unknowns)))
    o = _orc().tran(nl.ir_ptr, nl.n_unknowns, nl.nominal_params, 0, nl.tstep, nl.tstop)
    assert itr[0] == o["iters"] and (stt[0] & NOFB) == o["status"]
    assert rel_err(wave[0], o["rows"][:, 1:]).max() < TOL


def test_floating_node_gives_zero_vector_and_flag(torch_mod):
    from circuitsimulator_amd import Engine, Netlist
    nl = Netlist.from_text("V1 a 0 1\nR1 a 0 1k\nC1 b 0 1p\n")
    x, it, st = Engine(nl, 0).dc_host(B=1)
    assert it[0] == 1 and (st[0] & 0x4) and np.array_equal(x[0], np.zeros(3))


# ------------------------------------------------------- batched LU (a1, a2)

def test_lu_known_answers():
    from circuitsimulator_amd import lu_solve_batch
    A = np.array([[[0.0, 2.0], [3.0, 1.0]],          # forced row swap
                  [[1.0, 2.0], [2.0, 4.0]],          # singular: second pivot exactly 0 -> zero vector
                  [[1e-16, 0.0], [0.0, 1.0]],        # column maximum below 1e-15 -> zero vector
                  [[2.0, 0.0], [0.0, 4.0]]])
    b = np.array([[4.0, 5.0], [1.0, 1.0], [1.0, 1.0], [2.0, 2.0]])
    x, flags = lu_solve_batch(A, b)
    assert np.allclose(x[0], [1.0, 2.0], rtol=0, atol=1e-15) and flags[0] == 0
    assert np.array_equal(x[1], [0.0, 0.0]) and flags[1] & 0x4
    assert np.array_equal(x[2], [0.0, 0.0]) and flags[2] & 0x4
    assert np.array_equal(x[3], [1.0, 0.5]) and flags[3] == 0


def test_lu_random_systems_bitwise_equal_to_oracle():
    """Same pivot rule, same operation order, no FMA contraction -> the same bits.  n <= 32 runs on the register
    LU of the packed general kernels (four systems per wave; 16 / 17 = one / two rows per lane, odd sizes above 16
    are padded to even), larger n on the LDS version; B = 9 leaves a wave with one system only."""
    from circuitsimulator_amd import lu_solve_batch
    rs = np.random.RandomState(3)
    for n in (1, 2, 5, 13, 15, 16, 17, 18, 24, 31, 32, 33, 47, 63):
        B = 9
        A = rs.randn(B, n, n)
        A[rs.rand(B, n, n) < 0.6] = 0.0                 # sparse like an MNA matrix
        A += np.eye(n)[None] * rs.choice([1.0, 1e-3, 5.0], size=(B, n))[:, :, None]
        A[0, 0, :] = 0.0; A[0, 0, n - 1] = 1.0           # make pivoting happen
        b = rs.randn(B, n)
        x, flags = lu_solve_batch(A, b)
        for i in range(B):
            xo, fo = _orc().solve_lu(A[i], b[i])
            assert flags[i] == fo
            assert np.array_equal(x[i], xo), (n, i, np.abs(x[i] - xo).max())


def test_lu_pivot_tie_takes_first_row():
    from circuitsimulator_amd import lu_solve_batch
    A = np.array([[[1.0, 2.0, 3.0], [-1.0, 1.0, 0.0], [1.0, 0.0, 5.0]],
                  [[0.5, 2.0, 3.0], [-1.0, 1.0, 0.0], [1.0, 0.0, 5.0]]])
    b = np.array([[1.0, 2.0, 3.0], [1.0, 2.0, 3.0]])
    x, flags = lu_solve_batch(A, b)
    for i in range(2):
        xo, fo = _orc().solve_lu(A[i], b[i])
        assert np.array_equal(x[i], xo) and flags[i] == fo


def test_lu_decompose_factors_bitwise_equal_to_oracle():
    """Solver::luDecompose: packed L\\U and the permutation, same bits as the oracle."""
    from circuitsimulator_amd import lu_decompose_batch
    rs = np.random.RandomState(11)
    for n in (1, 3, 13, 31, 63):
        B = 6
        A = rs.randn(B, n, n)
        A[rs.rand(B, n, n) < 0.5] = 0.0
        A += np.eye(n)[None] * 0.5
        if n > 1:
            A[1] = A[1][::-1].copy()                     # force pivoting
        LU, perm, flags = lu_decompose_batch(A)
        for i in range(B):
            ok, LUo, permo = _orc().lu_decompose(A[i])
            assert bool(flags[i] & 0x4) == (not ok)
            if ok:
                assert np.array_equal(perm[i], permo), (n, i)
                assert np.array_equal(LU[i], LUo), (n, i, np.abs(LU[i] - LUo).max())
    LU, perm, flags = lu_decompose_batch(np.array([[[1.0, 2.0], [2.0, 4.0]]]))
    assert flags[0] & 0x4


def test_lu_dense_systems_beyond_lds_bitwise_equal_to_oracle():
    """Solver::solveLinearSystemLU / luDecompose for n > 63 (BASELINE configs[3] size: n = 257,
    528 KB per matrix): the in-place global-memory kernel, fully dense and MNA-like sparse inputs,
    forced pivoting, a first-maximum tie, a singular system."""
    from circuitsimulator_amd import lu_decompose_batch, lu_solve_batch
    rs = np.random.RandomState(5)
    for n in (64, 100, 257):
        B = 4
        A = rs.randn(B, n, n)
        A[1][rs.rand(n, n) < 0.9] = 0.0                   # sparse like an MNA matrix
        A[1] += np.eye(n) * 0.7
        A[2] = A[2][::-1].copy()                          # reversed rows: a swap in almost every column
        A[3, 5, :] = A[3, 9, :]                           # two equal rows: exact ties, then a tiny pivot
        A[0, 3, 0] = -A[0, :, 0].__abs__().max()          # tie in column 0 between an early and a later row
        A[0, 7, 0] = -A[0, 3, 0]
        b = rs.randn(B, n)
        x, flags = lu_solve_batch(A, b)
        LU, perm, dflags = lu_decompose_batch(A)
        for i in range(B):
            xo, fo = _orc().solve_lu(A[i], b[i])
            assert flags[i] == fo, (n, i, flags[i], fo)
            assert np.array_equal(x[i], xo), (n, i, np.abs(x[i] - xo).max())
            ok, LUo, permo = _orc().lu_decompose(A[i])
            assert bool(dflags[i] & 0x4) == (not ok), (n, i)
            if ok:
                assert np.array_equal(perm[i], permo), (n, i)
                assert np.array_equal(LU[i], LUo), (n, i, np.abs(LU[i] - LUo).max())
    # exactly singular: zero column -> the reference returns the zero vector
    A = rs.randn(1, 80, 80)
    A[0, :, 17] = 0.0
    x, flags = lu_solve_batch(A, rs.randn(1, 80))
    assert flags[0] & 0x4 and not x.any()


def test_cli_writes_the_reference_csv(tmp_path, buffer_nl):
    """csim_cli = the reference's main.cpp phases over the C++ shims (parseNetlist,
    computeDcOperatingPoint, runTransientAnalysisBackwardEuler) -> the reference's CSV."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "circuitsimulator_amd", "csim_cli")
    out = str(tmp_path / "tran.csv")
    p = subprocess.run([exe, netlist_path("buffer.sp"), out], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "Unknowns     : 13  (nodeEq=9, branchEq=4)" in p.stdout
    assert "V(107) = 2.490044 V   [eqIndex=3]" in p.stdout
    assert "I(L2, 117 -> 118)" in p.stdout and "[branchEq=12]" in p.stdout
    lines = open(out).read().strip().split("\n")
    assert lines[0] == buffer_nl.csv_header
    rows = np.array([[float(v) for v in l.split(",")] for l in lines[1:]])
    o = _orc().tran(buffer_nl.ir_ptr, 13, buffer_nl.nominal_params, 0, buffer_nl.tstep, buffer_nl.tstop)
    assert rows.shape == o["rows"].shape
    assert np.array_equal(rows[:, 0], np.array([float("%.9e" % t) for t in o["rows"][:, 0]]))
    assert np.abs(rows[:, 1:] - o["rows"][:, 1:]).max() <= 6e-10 * np.abs(o["rows"][:, 1:]).max()
    assert all(len(l.split(",")[1].split("e")[0]) in (11, 12) for l in lines[1:4])   # 9 decimals, scientific


def test_committed_schedules_match_the_planner(engines, torch_mod):
    """csrc/schedules/*.sched are what the general kernel's planner records on this GPU, and their
    first lines equal SURVEY.md Appendix F; dbmixer keeps ONE sequence through the run, also under
    MC perturbation; buffer at its shipped 1 ns step alternates between the committed ones."""
    import os
    from conftest import ROOT

    def committed(name):
        txt = open(os.path.join(ROOT, "circuitsimulator_amd", "csrc", "schedules", name + ".sched")).read()
        return [l.split("#")[0].strip() for l in txt.splitlines() if l.split("#")[0].strip()]

    nl, eng = engines["dbmixer"]
    params = eng.mc_params(12345, 0.05, 0, 4)
    tran_lines = [l for l in committed("dbmixer") if not l.startswith("dc")]
    dc_lines = [l[2:].strip() for l in committed("dbmixer") if l.startswith("dc")]
    assert len(tran_lines) == 1 and len(dc_lines) == 1
    for b in range(4):
        sched, nlu, ndiff = eng.record_pivot_schedule(params, b, None, 1500)
        assert sched == tran_lines[0] and ndiff == 0 and nlu > 10000
        dcs, other = eng.record_dc_pivot_schedules(params, b)            # the DC ramp keeps one sequence too
        assert [a for a, _ in dcs] == dc_lines and other == 0 and 400 < dcs[0][1] < 520
    nl, eng = engines["buffer"]
    params = eng.mc_params(12345, 0.05, 0, 2)
    sched, nlu, ndiff = eng.record_pivot_schedule(params, 0, 3e-11, 3000)
    assert sched == committed("buffer")[0] and ndiff == 0
    alts, other = eng.record_pivot_schedules(params, 0, 1e-9, 300)       # as shipped: several sequences
    assert len(alts) >= 4 and other == 0
    assert alts[0][0] == committed("buffer")[0]
    assert all(a in committed("buffer") for a, _ in alts)


INVERTER_CHAIN = """* three CMOS inverters driving an RC line (not one of the shipped netlists)
VDD vdd 0 DC 2.5
Vin in 0 SIN 1.25 1.0 50e6 0
Rg in g1 100
M1 o1 g1 vdd p 40e-6 0.5e-6 1
M2 o1 g1 0   n 20e-6 0.5e-6 2
M3 o2 o1 vdd p 40e-6 0.5e-6 1
M4 o2 o1 0   n 20e-6 0.5e-6 2
M5 o3 o2 vdd p 40e-6 0.5e-6 1
M6 o3 o2 0   n 20e-6 0.5e-6 2
R1 o3 l1 50
L1 l1 l2 2e-9
C1 l2 0 0.2e-12
R2 l2 out 75
C2 out 0 0.5e-12
.MODEL 1 VT -0.6 MU 2e-2 COX 2e-3 LAMBDA 0.04 CJ0 2e-14
.MODEL 2 VT 0.55 MU 6e-2 COX 2e-3 LAMBDA 0.04 CJ0 2e-14
.TRAN 2e-12 4e-9
.plotnv out
"""


def test_jit_scheduled_kernel_for_a_new_netlist(torch_mod, tmp_path, monkeypatch):
    """A netlist with no prebuilt kernel: plan with the general kernel, generate, hipcc, load;
    then scheduled == general == oracle."""
    import shutil
    from circuitsimulator_amd import Engine, Netlist
    if not (shutil.which("hipcc") or __import__("os").path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available for the JIT")
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    nl = Netlist.from_text(INVERTER_CHAIN)
    eng = Engine(nl, 0)
    assert eng.tran_kernel == "general"
    B, steps = 96, 600
    params = eng.mc_params(4242, 0.05, 0, B)
    slow = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
    eng.jit_scheduled(params, plan_steps=300)
    assert eng.tran_kernel == "scheduled"
    fast = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
    assert np.array_equal(fast["step_iters"], slow["step_iters"])
    assert np.array_equal(fast["status"] & NOFB, slow["status"])
    assert rel_err(fast["x"].T, slow["x"].T).max() < TOL
    assert ((fast["status"] & FALLBACK) != 0).sum() < B // 2      # the fast path really ran for most
    ph = params.cpu().numpy()
    o = _orc().tran(nl.ir_ptr, nl.n_unknowns, ph, 5, nl.tstep, nl.tstep * steps, want_rows=False)
    assert fast["iters"][5] == o["iters"]
    assert rel_err(fast["x"][:, 5], o["x_final"]).max() < TOL
    # second engine: the cached library is reused (no compile)
    eng2 = Engine(nl, 0)
    eng2.jit_scheduled(params, plan_steps=300)
    assert eng2.tran_kernel == "scheduled"


def test_non_convergent_instances_leave_the_fast_path(torch_mod, tmp_path, monkeypatch):
    """tanalisis.cpp:369-376: a step whose Newton iteration ends at the cap (50) is kept with a WARNING.
    Such instances are chaotic -- the generated kernel's FMA contraction moves them by +-1-2 iterations
    per step -- so the generated kernel treats 'cap reached' like a failed pivot check and the
    bit-faithful general kernel redoes the step.  Here the kernel is generated FROM the pivot sequences
    of a non-convergent instance, so nothing but that rule keeps those instances off the fast path.
    Bar: per-step NR counts and status equal to the general kernel on all 96 instances, state within
    1e-9; the non-convergent ones also against the oracle."""
    from circuitsimulator_amd import Engine, Netlist
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    nl = Netlist.from_text(INVERTER_CHAIN)
    eng = Engine(nl, 0)
    B, steps = 96, 600
    params = eng.mc_params(4242, 0.05, 0, B)
    slow = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
    nonconv = np.where((slow["status"] & 0x02) != 0)[0]
    assert len(nonconv) >= 3, "the seed no longer yields non-convergent samples: pick another"
    assert (slow["step_iters"][:, nonconv].max(axis=0) == 50).all()
    seqs, other = eng.record_pivot_schedules(params, int(nonconv[0]), nl.tstep, steps)
    assert other == 0
    eng.jit_with_schedules([s for s, _ in seqs[:4]])
    assert eng.tran_kernel == "scheduled"
    fast = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True, chunks=[250, 350])
    assert np.array_equal(fast["step_iters"], slow["step_iters"])
    assert np.array_equal(fast["status"] & NOFB, slow["status"])
    assert rel_err(fast["x"].T, slow["x"].T).max() < TOL
    assert ((fast["status"][nonconv] & (FALLBACK | FAITHFUL)) != 0).all()          # every one of them left the fast kernel
    conv = np.setdiff1d(np.arange(B), nonconv)
    assert ((fast["status"][conv] & (FALLBACK | FAITHFUL)) == 0).sum() > len(conv) // 2      # and the fast path still ran
    ph = params.cpu().numpy()
    for b in list(nonconv[:3]) + [0]:
        o = _orc().tran(nl.ir_ptr, nl.n_unknowns, ph, int(b), nl.tstep, nl.tstep * steps, want_rows=False,
                        want_step_iters=True)
        assert np.array_equal(fast["step_iters"][:, b], o["step_iters"]), b
        assert (fast["status"][b] & 0x03) == (o["status"] & 0x03), b        # non-finite / non-convergence bits
        assert rel_err(fast["x"][:, b], o["x_final"]).max() < TOL, b


def test_run_shorter_than_one_step_yields_the_t0_row(engines):
    """tstop < tstep is legal upstream (nSteps = 0, tanalisis.cpp:238): only the t = 0 row, which is
    the DC operating point -- also when a scheduled kernel is loaded."""
    nl, eng = engines["dbmixer"]
    assert eng.tran_kernel == "scheduled"
    probes = [1, 2]
    wave, xf, it, st = eng.tran_host(B=3, tstep=nl.tstep, tstop=0.4 * nl.tstep, tstart=0.0, probes=probes)
    x_dc, _, _ = eng.dc_host(B=3)
    assert wave.shape == (3, 1, 2) and (it == 0).all()
    assert np.array_equal(wave[:, 0, :], x_dc[:, probes]) and np.array_equal(xf, x_dc)


# ------------------------------- BASELINE configs[3]: RC ladder, N = 257 unknowns

@pytest.fixture(scope="module")
def ladder(torch_mod):
    from circuitsimulator_amd import Engine, Netlist
    from circuitsimulator_amd.workloads import rc_ladder_netlist
    nl = Netlist.from_text(rc_ladder_netlist(256))
    assert (nl.n_unknowns, nl.n_elems, nl.n_params) == (257, 511, 516)
    return nl, Engine(nl, 0)


def test_rc_ladder_large_n_general_kernels(ladder, torch_mod):
    """N = 257 > 63: the bit-guided global-scratch kernels.  Linear circuit -> direct DC (one LU,
    no gmin); 100 transient steps = 1 687 NR iterations (SURVEY.md 8d #4 fingerprint)."""
    nl, eng = ladder
    eng.set_kernel("general")
    B = 3
    params = eng.mc_params(12345, 0.05, 0, B)
    r = _run_tran(torch_mod, eng, params, 100, nl.tstep, probes=[0, 128, 255, 256], stride=10, want_step_iters=True)
    eng.set_kernel("auto")
    assert list(r["dc_iters"]) == [1, 1, 1]
    assert r["iters"][0] == 1687
    ph = params.cpu().numpy()
    for b in (0, 2):
        xo, ito, sto = _orc().dc(nl.ir_ptr, 257, ph, b)
        assert rel_err(r["x_dc"][:, b], xo).max() < TOL
        o = _orc().tran(nl.ir_ptr, 257, ph, b, nl.tstep, nl.tstop, want_step_iters=True)
        assert r["iters"][b] == o["iters"] and np.array_equal(r["step_iters"][:, b], o["step_iters"])
        assert r["status"][b] == o["status"]
        assert rel_err(r["x"][:, b], o["x_final"]).max() < TOL
        ref = o["rows"][::10, 1:][:, [0, 128, 255, 256]]
        assert np.abs(r["wave"][:, :, b] - ref).max() <= TOL * max(np.abs(ref).max(), 1e-6)


def test_circuit_beyond_320_unknowns(torch_mod, tmp_path, monkeypatch):
    """The reference takes any N (src/circuit.cpp:38-40).  A 600-node RC ladder (601 unknowns) through the
    large-N general kernels (structure bit matrix in LDS, values in global memory) and through the generated
    linear-circuit kernel (16 instances per workgroup at this size), against the oracle; and the size at
    which the engine does refuse."""
    from circuitsimulator_amd import CsimError, Engine, Netlist
    from circuitsimulator_amd.workloads import rc_ladder_netlist
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    nl = Netlist.from_text(rc_ladder_netlist(600))
    N = nl.n_unknowns
    assert N == 601
    eng = Engine(nl, 0)
    B, steps = 20, 3
    params = eng.mc_params(5, 0.05, 0, B)
    slow = _run_tran(torch_mod, eng, params[:, :2].contiguous(), steps, nl.tstep, want_step_iters=True)
    ph = params.cpu().numpy()
    o = _orc().tran(nl.ir_ptr, N, ph, 1, nl.tstep, nl.tstep * steps, want_rows=False, want_step_iters=True)
    assert np.array_equal(slow["step_iters"][:, 1], o["step_iters"]) and slow["status"][1] == o["status"]
    assert rel_err(slow["x"][:, 1], o["x_final"]).max() < TOL
    xo, ito, _ = _orc().dc(nl.ir_ptr, N, ph, 1)
    assert slow["dc_iters"][1] == ito == 1 and rel_err(slow["x_dc"][:, 1], xo).max() < TOL
    eng.jit_scheduled(params, plan_steps=2)
    assert eng.tran_kernel == "scheduled"
    fast = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
    assert not (fast["status"] & 0x27).any()
    assert np.array_equal(fast["step_iters"][:, :2], slow["step_iters"])
    assert rel_err(fast["x"][:, :2].T, slow["x"].T).max() < TOL
    o = _orc().tran(nl.ir_ptr, N, ph, B - 1, nl.tstep, nl.tstep * steps, want_rows=False, want_step_iters=True)
    assert np.array_equal(fast["step_iters"][:, B - 1], o["step_iters"])
    assert rel_err(fast["x"][:, B - 1], o["x_final"]).max() < TOL
    with pytest.raises(CsimError) as e:                     # 1025 unknowns: beyond the large-N kernels
        Engine(Netlist.from_text(rc_ladder_netlist(1024)), 0)
    assert e.value.code == -5


def test_rc_ladder_full_waveform_host_api(ladder):
    """All 257 unknowns probed (more probes than lanes) through the instance-major host API."""
    nl, eng = ladder
    eng.set_kernel("general")
    wave, xf, it, st = eng.tran_host(B=1, tstop=20e-9, probes=list(range(257)))
    eng.set_kernel("auto")
    o = _orc().tran(nl.ir_ptr, 257, nl.nominal_params, 0, nl.tstep, 20e-9)
    assert wave.shape == (1, 21, 257) and it[0] == o["iters"]
    assert rel_err(wave[0], o["rows"][:, 1:]).max() < TOL


def test_rc_ladder_scheduled_kernel_at_config_batch(ladder, torch_mod, tmp_path, monkeypatch):
    """configs[3] per-GPU share: B = 8192 instances of the N = 257 ladder, JIT-specialised kernel."""
    import shutil, os
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available for the JIT")
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    nl, eng = ladder
    B = 8192
    params = eng.mc_params(12345, 0.05, 0, B)
    eng.jit_scheduled(params, plan_steps=5)
    assert eng.tran_kernel == "scheduled"
    r = _run_tran(torch_mod, eng, params, 100, nl.tstep, chunks=[40, 60])
    assert not (r["status"] & 0x27).any()
    assert r["iters"][0] == 1687
    one = _run_tran(torch_mod, eng, params[:, :1].contiguous(), 100, nl.tstep)
    assert np.array_equal(one["x"][:, 0], r["x"][:, 0])                       # batch invariance
    ph = params[:, [0, 1, B - 1]].cpu().numpy()
    for j, b in enumerate((0, 1, B - 1)):
        o = _orc().tran(nl.ir_ptr, 257, ph, j, nl.tstep, nl.tstop, want_rows=False)
        assert r["iters"][b] == o["iters"]
        assert rel_err(r["x"][:, b], o["x_final"]).max() < TOL


def test_mid_size_nonlinear_circuit_big_kernels(torch_mod):
    """A nonlinear circuit with 63 < N <= 320: Newton DC ramp + transient on the large-N kernels."""
    from circuitsimulator_amd import Engine, Netlist
    stages = 20
    txt = ["VDD vdd 0 DC 2.5", "Vin n0 0 SIN 1.25 1.0 100e6 0"]
    for k in range(stages):
        txt += ["M%da n%d g%d vdd p 40e-6 0.5e-6 1" % (k, k + 1, k), "M%db n%d g%d 0 n 20e-6 0.5e-6 2" % (k, k + 1, k),
                "R%d n%d g%d 200" % (k, k, k), "C%d n%d 0 5e-15" % (k, k + 1), "L%d n%d t%d 1e-10" % (k, k + 1, k),
                "R%dt t%d 0 1e5" % (k, k)]
    txt += [".MODEL 1 VT -0.6 MU 2e-2 COX 2e-3 LAMBDA 0.04 CJ0 2e-14", ".MODEL 2 VT 0.55 MU 6e-2 COX 2e-3 LAMBDA 0.04 CJ0 2e-14",
            ".TRAN 5e-12 1e-9"]
    nl = Netlist.from_text("\n".join(txt) + "\n")
    assert 63 < nl.n_unknowns <= 320
    eng = Engine(nl, 0)
    B = 2
    params = eng.mc_params(9, 0.05, 0, B)
    r = _run_tran(torch_mod, eng, params, 40, nl.tstep, want_step_iters=True)
    ph = params.cpu().numpy()
    for b in range(B):
        xo, ito, sto = _orc().dc(nl.ir_ptr, nl.n_unknowns, ph, b)
        assert r["dc_iters"][b] == ito
        assert rel_err(r["x_dc"][:, b], xo).max() < TOL
        o = _orc().tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstep * 40, want_rows=False, want_step_iters=True)
        assert r["iters"][b] == o["iters"] and np.array_equal(r["step_iters"][:, b], o["step_iters"])
        assert (r["status"][b] & NOFB) == o["status"]
        assert rel_err(r["x"][:, b], o["x_final"]).max() < TOL


def test_dc_sweep_axis(torch_mod):
    """a .DC card executed as a batch axis -- the buffer's DC transfer curve,
    every sweep point checked against the oracle on the same parameter column."""
    from circuitsimulator_amd import Engine, Netlist
    text = open(netlist_path("buffer.sp")).read() + "\n.DC Vin -1.5 1.5 0.125\n"
    nl = Netlist.from_text(text)
    eng = Engine(nl, 0)
    values, x, it, st = eng.dc_sweep(0)
    torch_mod.cuda.synchronize()
    assert len(values) == 25 and values[0] == -1.5 and values[-1] == 1.5
    x, it, st = x.cpu().numpy(), it.cpu().numpy(), st.cpu().numpy().astype(np.uint32)
    _, table = nl.dc_sweep_table(0)
    for j in range(25):
        xo, ito, sto = _orc().dc(nl.ir_ptr, nl.n_unknowns, table, j)
        assert it[j] == ito and st[j] == sto, j
        assert rel_err(x[:, j], xo).max() < TOL, j
    # the input node follows dc + v0 (SIN offset 1.5 V): V(101) = value + 1.5
    assert np.allclose(x[nl.eq_names.index("101")], values + 1.5, rtol=0, atol=1e-8)
    out = x[nl.eq_names.index("118")]
    assert out[0] < 0.1 and out[-1] > 2.9          # non-inverting buffer: low in -> low out, high in -> high out


def test_cpp_batch_api(engines):
    """csim::BatchEngine (api/analysis.hpp) from C++: Monte-Carlo table, batched DC, batched transient."""
    import json
    import os
    import subprocess
    from conftest import ROOT
    nl, eng = engines["dbmixer"]
    exe = os.path.join(ROOT, "circuitsimulator_amd", "csim_batch_demo")
    B, steps = 5, 300
    p = subprocess.run([exe, netlist_path("dbmixer.sp"), str(B), str(steps)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    recs = [json.loads(l) for l in p.stdout.strip().splitlines()]
    assert len(recs) == B
    tab = nl.mc_params_host(12345, 0.05, 0, B)
    for r in recs:
        b = r["b"]
        xo, ito, sto = _orc().dc(nl.ir_ptr, nl.n_unknowns, tab, b)
        assert r["dc_iters"] == ito and r["dc_status"] == sto
        o = _orc().tran(nl.ir_ptr, nl.n_unknowns, tab, b, nl.tstep, nl.tstep * steps, want_rows=False)
        assert r["tran_iters"] == o["iters"] and (r["status"] & NOFB) == (o["status"] | sto)
        assert r["rows"] == 2                                   # out_stride = n_steps: t = 0 and the last row
        assert rel_err(np.array(r["x_final"]), o["x_final"]).max() < TOL
        assert abs(r["wave_last"] - o["x_final"][-1]) <= TOL * max(abs(o["x_final"][-1]), 1e-6)


@pytest.mark.parametrize("variant", ["r_zero", "c_nonpositive", "l_zero"])
def test_degenerate_element_values_follow_the_reference(variant, torch_mod, tmp_path, monkeypatch):
    """a zero-ohm resistor is skipped (element.cpp:20-23); C <= 0 and L <= 0
    contribute nothing in transient (tanalisis.cpp:65,296) -- an L <= 0 inductor leaves its branch row
    empty, the factorisation fails and every solve returns the ZERO vector (solver.hpp:94-97), which
    the damped update then relaxes towards.  General kernel, JIT kernel (whose +-1 folding of the
    inductor incidence is guarded by an L > 0 precondition) and oracle must agree."""
    import shutil, os
    from circuitsimulator_amd import Engine, Netlist
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    base = ["VDD vdd 0 DC 2", "Vin in 0 SIN 1 0.5 200e6 0", "Rg in g 50", "M1 d g 0 n 20e-6 0.5e-6 2", "Rd vdd d 500",
            "R2 d o 30", "L1 o p 1e-9", "C1 p 0 0.3e-12", "R3 p 0 2e3",
            ".MODEL 2 VT 0.55 MU 6e-2 COX 2e-3 LAMBDA 0.04 CJ0 2e-14", ".TRAN 5e-12 1e-9"]
    edit = {"r_zero": ("R2 d o 30", "R2 d o 0\nR2b d o 30"), "c_nonpositive": ("C1 p 0 0.3e-12", "C1 p 0 -1e-12"),
            "l_zero": ("L1 o p 1e-9", "L1 o p 0")}[variant]
    nl = Netlist.from_text("\n".join(base).replace(edit[0], edit[1]) + "\n")
    eng = Engine(nl, 0)
    B, steps = 6, 120
    params = eng.mc_params(3, 0.05, 0, B)
    gen = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
    ph = params.cpu().numpy()
    for b in (0, B - 1):
        xo, ito, sto = _orc().dc(nl.ir_ptr, nl.n_unknowns, ph, b)
        assert gen["dc_iters"][b] == ito
        assert rel_err(gen["x_dc"][:, b], xo).max() < TOL
        o = _orc().tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstep * steps, x0=xo, want_rows=False, want_step_iters=True)
        assert gen["iters"][b] == o["iters"] and np.array_equal(gen["step_iters"][:, b], o["step_iters"])
        assert gen["status"][b] == (o["status"] | sto)
        assert rel_err(gen["x"][:, b], o["x_final"]).max() < TOL
    if variant == "l_zero":
        assert (gen["status"] & 0x4).all()                       # CSIM_ST_LU_TINY_PIVOT on every instance
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        try:
            eng.jit_scheduled(params, plan_steps=20)
        except Exception:
            assert variant == "l_zero"                           # no successful factorisation to plan from
            return
        fast = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
        assert np.array_equal(fast["step_iters"], gen["step_iters"])
        assert np.array_equal(fast["status"] & NOFB, gen["status"])
        assert rel_err(fast["x"].T, gen["x"].T).max() < TOL


def test_batch4096_every_instance_against_the_oracle(engines, torch_mod):
    """configs[2] size: all 4096 Monte-Carlo instances (not a sample) against the CPU oracle on the
    same parameter table -- NR counts per instance equal, every unknown within the parity bar."""
    nl, eng = engines["dbmixer"]
    B, steps = 4096, 30
    params = eng.mc_params(12345, 0.05, 0, B)
    r = _run_tran(torch_mod, eng, params, steps, nl.tstep)
    assert not (r["status"] & 0x27).any()
    ph = params.cpu().numpy()
    worst = 0.0
    for b in range(B):
        o = _orc().tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstep * steps, want_rows=False)
        assert r["iters"][b] == o["iters"], b
        worst = max(worst, rel_err(r["x"][:, b], o["x_final"]).max())
    assert worst < TOL, worst


# ------------------------------------------------- PULSE / PWL sources

def test_pulse_pwl_sources_general_and_scheduled(torch_mod, tmp_path, monkeypatch):
    """TranWaveform::eval PULSE / PWL (reference include/sim.hpp:80-138) on V and I sources: the
    general kernel, the JIT-generated scheduled kernel and the host API against the oracle."""
    from circuitsimulator_amd import Engine, Netlist
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    nl = Netlist.from_file(netlist_path("pulse_pwl.sp"))
    eng = Engine(nl, 0)
    assert eng.tran_kernel == "general"
    B, steps = 70, 400
    params = eng.mc_params(99, 0.05, 0, B)
    probes = [nl.eq_names.index(n) for n in ("101", "110", "104", "111", "120")]
    slow = _run_tran(torch_mod, eng, params, steps, nl.tstep, probes=probes, want_step_iters=True)
    ph = params.cpu().numpy()
    orc = _orc()
    for b in (0, 33, 69):
        o = orc.tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstop, want_step_iters=True)
        assert o["n_steps"] == steps
        assert np.array_equal(slow["step_iters"][:, b], o["step_iters"])
        assert slow["status"][b] == o["status"] == 0
        want = o["rows"][:, [1 + q for q in probes]]
        assert rel_err(slow["wave"][:, :, b], want).max() < TOL
        assert rel_err(slow["x"][:, b], o["x_final"]).max() < TOL
    eng.jit_scheduled(params, plan_steps=steps)
    assert eng.tran_kernel == "scheduled"
    fast = _run_tran(torch_mod, eng, params, steps, nl.tstep, probes=probes, want_step_iters=True)
    assert np.array_equal(fast["step_iters"], slow["step_iters"])
    assert np.array_equal(fast["status"] & NOFB, slow["status"])
    assert rel_err(fast["wave"].transpose(2, 0, 1).reshape(-1, len(probes)),
                   slow["wave"].transpose(2, 0, 1).reshape(-1, len(probes))).max() < TOL
    assert ((fast["status"] & FALLBACK) != 0).sum() < B // 2
    for lanes in (4, 1):                 # (the run above: sixteen lanes per instance; four keep the sources' parameters packed)
        eng.set_option("lanes_per_instance", lanes)
        assert eng.lanes_for_batch(B) == lanes
        other = _run_tran(torch_mod, eng, params, steps, nl.tstep, probes=probes, want_step_iters=True)
        assert np.array_equal(other["step_iters"], slow["step_iters"]), lanes
        assert np.array_equal(other["status"] & NOFB, slow["status"]), lanes
        assert rel_err(other["wave"].transpose(2, 0, 1).reshape(-1, len(probes)),
                       slow["wave"].transpose(2, 0, 1).reshape(-1, len(probes))).max() < TOL, lanes
    eng.set_option("lanes_per_instance", 0)
    # host API, nominal instance: every CSV column
    wave, xf, it, st = eng.tran_host(B=1, probes=list(range(nl.n_unknowns)))
    o = orc.tran(nl.ir_ptr, nl.n_unknowns, nl.nominal_params, 0, nl.tstep, nl.tstop)
    assert it[0] == o["iters"] and st[0] & NOFB == 0
    assert rel_err(wave[0], o["rows"][:, 1:]).max() < TOL


# --------------------------------------------- scheduled DC operating-point kernel

def test_scheduled_dc_kernel_equals_general_and_oracle(engines, torch_mod):
    """The generated lane-per-instance DC kernel (source ramp + ConvController on the recorded "dc"
    pivot sequences) against the general kernel on the whole batch and the oracle on a sample."""
    import time
    torch = torch_mod
    nl, eng = engines["dbmixer"]
    B = 4096
    params = eng.mc_params(12345, 0.05, 0, B)
    eng.set_kernel("general")
    x_g, it_g, st_g = eng.dc(params)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x_g, it_g, st_g = eng.dc(params)
    torch.cuda.synchronize()
    t_general = time.perf_counter() - t0
    eng.set_kernel("auto")
    x_s, it_s, st_s = eng.dc(params)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x_s, it_s, st_s = eng.dc(params)
    torch.cuda.synchronize()
    t_sched = time.perf_counter() - t0
    print("DC B=%d: general %.2f ms, scheduled %.2f ms" % (B, 1e3 * t_general, 1e3 * t_sched))
    xg, xs = x_g.cpu().numpy(), x_s.cpu().numpy()
    stg, sts = st_g.cpu().numpy().astype(np.uint32), st_s.cpu().numpy().astype(np.uint32)
    assert np.array_equal(it_g.cpu().numpy(), it_s.cpu().numpy())
    assert np.array_equal(stg, sts & NOFB)
    assert ((sts & FALLBACK_DC) != 0).sum() <= B // 100       # the scheduled kernel really produced the results
    assert rel_err(xs.T, xg.T).max() < TOL
    assert t_sched < t_general                                   # and it is the faster path
    ph = params.cpu().numpy()
    for b in (0, 1, 777, 4095):
        xo, ito, sto = _orc().dc(nl.ir_ptr, nl.n_unknowns, ph, b)
        assert it_s[b].item() == ito and (sts[b] & NOFB) == sto
        assert rel_err(xs[:, b], xo).max() < TOL


def test_scheduled_dc_violators_are_replayed_by_the_general_kernel(torch_mod, tmp_path, monkeypatch):
    """A circuit whose DC ramp walks through many pivot sequences (buffer.sp: 10 in the nominal
    instance, 15 over a Monte-Carlo batch).  By default the JIT gives such a circuit no DC kernel;
    forced to keep a partial cover, instances violate and must come back from the general kernel
    with identical results."""
    from circuitsimulator_amd import Engine, Netlist
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    monkeypatch.setenv("CSIM_JIT_DC_FORCE", "1")
    nl = Netlist.from_file(netlist_path("buffer.sp"))
    eng = Engine(nl, 0)
    B = 256
    params = eng.mc_params(7, 0.05, 0, B)
    eng.set_kernel("general")
    x_g, it_g, st_g = eng.dc(params)
    eng.set_kernel("auto")
    eng.jit_scheduled(params, plan_steps=300)
    x_s, it_s, st_s = eng.dc(params)
    sts = st_s.cpu().numpy().astype(np.uint32)
    nfb = int(((sts & FALLBACK_DC) != 0).sum())
    print("buffer DC: %d of %d instances replayed by the general kernel" % (nfb, B))
    assert 0 < nfb <= B
    assert np.array_equal(it_g.cpu().numpy(), it_s.cpu().numpy())
    assert np.array_equal(st_g.cpu().numpy().astype(np.uint32), sts & NOFB)
    assert rel_err(x_s.cpu().numpy().T, x_g.cpu().numpy().T).max() < TOL


# --------------------------------------------------- randomized netlists (general kernels)

def _random_netlist(rs, n_nodes, n_mos):
    """A random RLC + MOSFET + source circuit in the reference dialect.  Every node gets a resistive
    path to ground most of the time; now and then one is left floating on purpose."""
    nodes = ["n%d" % i for i in range(1, n_nodes + 1)]
    lines = ["* random circuit", "VDD vdd 0 DC %.3g" % rs.choice([1.8, 2.5, 3.3])]
    k = 0
    for nd in nodes:
        if rs.rand() < 0.9:
            k += 1
            lines.append("R%d %s %s %.4g" % (k, nd, rs.choice(["0", "vdd"] + nodes), 10 ** rs.uniform(1.5, 5)))
    for _ in range(n_nodes):
        a, b = rs.choice(nodes + ["0", "vdd"], 2, replace=False)
        k += 1
        lines.append("R%d %s %s %.4g" % (k, a, b, 10 ** rs.uniform(2, 5)))
    for i, nd in enumerate(nodes):
        if rs.rand() < 0.7:
            lines.append("C%d %s 0 %.4g" % (i + 1, nd, 10 ** rs.uniform(-14, -11)))
    for i in range(rs.randint(0, 3)):
        a, b = rs.choice(nodes, 2, replace=False)
        lines.append("L%d %s %s %.4g" % (i + 1, a, b, 10 ** rs.uniform(-10, -8)))
    src = rs.choice(nodes)
    lines.append("VIN %s 0 SIN %.3g %.3g %.4g 0" % (src, rs.uniform(0.5, 1.5), rs.uniform(0.1, 1.0), 10 ** rs.uniform(7, 9)))
    if rs.rand() < 0.5:
        lines.append("I1 %s %s %.4g" % (rs.choice(nodes), rs.choice(["0"] + nodes), 10 ** rs.uniform(-6, -4)))
    for i in range(n_mos):
        d, g, s_ = rs.choice(nodes + ["vdd"], 3, replace=True)
        if rs.rand() < 0.5:
            lines.append("M%d %s %s %s n %.3ge-6 0.35e-6 2" % (i + 1, d, g, rs.choice([s_, "0"]), rs.uniform(5, 40)))
        else:
            lines.append("M%d %s %s %s p %.3ge-6 0.35e-6 1" % (i + 1, d, g, rs.choice([s_, "vdd"]), rs.uniform(5, 60)))
    lines += [".MODEL 1 VT -0.75 MU 5e-2 COX 0.3e-4 LAMBDA 0.05 CJ0 4.0e-14",
              ".MODEL 2 VT 0.83 MU 1.5e-1 COX 0.3e-4 LAMBDA 0.05 CJ0 4.0e-14",
              ".TRAN %.3g %.3g" % (1e-10, 4e-9)]
    return "\n".join(lines) + "\n"


def test_random_netlists_general_kernels_vs_oracle(torch_mod):
    """40 seeded random circuits (3..18 nodes, 0..5 MOSFETs, inductors, floating nodes now and then)
    through the general DC and transient kernels, three Monte-Carlo instances each, against the oracle:
    status words and NR counts equal, states within the parity bar.  Covers what the two shipped
    netlists cannot: arbitrary stamping orders, pivot ties, singular systems, non-convergence."""
    from circuitsimulator_amd import Engine, Netlist
    orc = _orc()
    worst = 0.0
    n_flagged = 0
    for seed in range(40):
        rs = np.random.RandomState(1000 + seed)
        text = _random_netlist(rs, rs.randint(3, 19), rs.randint(0, 6))
        nl = Netlist.from_text(text)
        eng = Engine(nl, 0)
        B, steps = 3, 40
        params = eng.mc_params(seed, 0.05, 0, B)
        r = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
        ph = params.cpu().numpy()
        for b in range(B):
            xo, ito, sto = orc.dc(nl.ir_ptr, nl.n_unknowns, ph, b)
            assert r["dc_iters"][b] == ito, (seed, b, "dc iters", r["dc_iters"][b], ito)
            assert rel_err(r["x_dc"][:, b], xo).max() < TOL, (seed, b, "dc x")
            o = orc.tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstep * steps, want_step_iters=True)
            dc_bits = sto & 0x1C                                    # DC flags live in the same status word
            assert (r["status"][b] & NOFB) == (o["status"] | dc_bits), (seed, b, hex(r["status"][b]), hex(o["status"]), hex(sto))
            n_flagged += int(r["status"][b] != 0)
            if o["status"] & 0x1:                                   # the reference would have thrown: stopped early
                continue
            assert np.array_equal(r["step_iters"][:, b], o["step_iters"]), (seed, b, "tran iters")
            e = rel_err(r["x"][:, b], o["x_final"]).max()
            worst = max(worst, e)
            assert e < TOL, (seed, b, e)
    print("random netlists: worst relative deviation %.2e, %d flagged instance runs" % (worst, n_flagged))


def test_random_netlists_generated_kernels(torch_mod, tmp_path, monkeypatch):
    """The kernel generator on six of the random circuits: plan, generate, hipcc, load; the generated
    transient (and, where a few sequences cover the ramp, DC) kernels against the general kernels on a
    Monte-Carlo batch -- per-step NR counts and status equal, states within the parity bar."""
    import shutil
    from circuitsimulator_amd import Engine, Netlist
    if not (shutil.which("hipcc") or __import__("os").path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available for the JIT")
    monkeypatch.setenv("CSIM_JIT_DIR", str(tmp_path / "jit"))
    n_dc_kernels = 0
    for seed in (3, 8, 14, 20, 27, 36):
        rs = np.random.RandomState(1000 + seed)
        nl = Netlist.from_text(_random_netlist(rs, rs.randint(3, 19), rs.randint(0, 6)))
        eng = Engine(nl, 0)
        B, steps = 70, 60
        params = eng.mc_params(seed, 0.05, 0, B)
        slow = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
        eng.jit_scheduled(params, plan_steps=steps)
        assert eng.tran_kernel == "scheduled", seed
        fast = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
        assert np.array_equal(fast["dc_iters"], slow["dc_iters"]), seed
        assert rel_err(fast["x_dc"].T, slow["x_dc"].T).max() < TOL, seed
        assert np.array_equal(fast["step_iters"], slow["step_iters"]), seed
        assert np.array_equal(fast["status"] & NOFB, slow["status"]), seed
        assert rel_err(fast["x"].T, slow["x"].T).max() < TOL, seed
        assert "group4" in eng.sched_info["text"], seed
        eng.set_option("lanes_per_instance", 4)
        quad = _run_tran(torch_mod, eng, params, steps, nl.tstep, want_step_iters=True)
        assert np.array_equal(quad["step_iters"], slow["step_iters"]), seed
        assert np.array_equal(quad["status"] & NOFB, slow["status"]), seed
        assert rel_err(quad["x"].T, slow["x"].T).max() < TOL, seed
        n_dc_kernels += int(((fast["status"] & FALLBACK_DC) != 0).any())
    print("random netlists: %d of 6 circuits had DC instances replayed by the general kernel" % n_dc_kernels)


def test_auto_jit_through_the_reference_shaped_cli(tmp_path):
    """CSIM_AUTO_JIT=1: csim_cli (= the reference's main.cpp over the C++ shims) on a netlist with no
    prebuilt kernel specialises it on first use; the CSV equals the general kernel's within the parity bar."""
    import os
    import shutil
    import subprocess
    from conftest import ROOT
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available for the JIT")
    cli = os.path.join(ROOT, "circuitsimulator_amd", "csim_cli")
    net = tmp_path / "chain.sp"
    net.write_text(INVERTER_CHAIN)
    outs = {}
    for mode in ("general", "auto_jit"):
        env = dict(os.environ, CSIM_JIT_DIR=str(tmp_path / "jit"))
        if mode == "auto_jit":
            env["CSIM_AUTO_JIT"] = "1"
        out = tmp_path / (mode + ".csv")
        p = subprocess.run([cli, str(net), str(out)], capture_output=True, text=True, env=env, timeout=300)
        assert p.returncode == 0, p.stderr[-500:]
        outs[mode] = np.loadtxt(out, delimiter=",", skiprows=1)
    assert any(f.endswith(".so") for f in os.listdir(tmp_path / "jit"))          # a kernel really was generated
    a, b = outs["general"], outs["auto_jit"]
    assert a.shape == b.shape and a.shape[0] > 1000
    nl_nodes = a.shape[1] - 1
    assert rel_err(b[:, 1:], a[:, 1:]).max() < 1e-8      # CSV carries 10 significant digits


def test_config4_share_of_rank_7(engines, torch_mod):
    """BASELINE configs[4] (dbmixer.sp, 2^20 Monte-Carlo samples on 8 GPUs): the share of the LAST rank, instances
    7 * 131 072 ... 8 * 131 072 - 1, 20 time steps on the kernel the engine picks for that batch.  The device's
    parameter table equals the host mirror at that offset bit for bit; launches cut differently and duplicated
    columns give the same bits; first / middle / last instance against the oracle; nothing flagged."""
    torch = torch_mod
    nl, eng = engines["dbmixer"]
    B, steps = 131072, 20
    b_first = 7 * B
    params = eng.mc_params(12345, 0.05, b_first, B)
    host = nl.mc_params_host(12345, 0.05, b_first, B)
    assert np.array_equal(params.cpu().numpy(), host)
    assert eng.lanes_for_batch(B) == 1                                  # one lane per instance fills the chip at this size
    one = _run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True)
    assert not (one["status"] & 0xA7).any()
    cut = _run_tran(torch, eng, params, steps, nl.tstep, want_step_iters=True, chunks=[7, 13])
    for key in ("x", "iters", "step_iters", "status", "dc_iters", "x_dc"):
        assert np.array_equal(cut[key], one[key]), key
    dup = params.clone()
    dup[:, 1::2] = dup[:, 0::2]                                          # every odd column repeats its left neighbour
    two = _run_tran(torch, eng, dup, steps, nl.tstep)
    assert np.array_equal(two["x"][:, 1::2], two["x"][:, 0::2]) and np.array_equal(two["iters"][1::2], two["iters"][0::2])
    assert np.array_equal(two["x"][:, 0::2], one["x"][:, 0::2])           # and a column does not care who its neighbours are
    for b in (0, B // 2, B - 1):
        xo, ito, sto = _orc().dc(nl.ir_ptr, nl.n_unknowns, host, b)
        assert one["dc_iters"][b] == ito and rel_err(one["x_dc"][:, b], xo).max() < TOL
        o = _orc().tran(nl.ir_ptr, nl.n_unknowns, host, b, nl.tstep, nl.tstep * steps, want_rows=False, want_step_iters=True)
        assert np.array_equal(one["step_iters"][:, b], o["step_iters"]) and (one["status"][b] & NOFB) == o["status"]
        assert rel_err(one["x"][:, b], o["x_final"]).max() < TOL
