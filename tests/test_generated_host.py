"""The generator's arithmetic, checked WITHOUT a GPU: the bit-faithful generated DC kernel (csim_dc_faithful_kernel: the
reference's operations on the recorded DC pivot sequences, codegen.cpp) is plain C++ apart from a handful of HIP names,
so it is compiled for the host with g++ (-ffp-contract=off, one "lane") and run against the oracle -- operating point and
NR pass count must be equal BIT FOR BIT (reference: src/dcanalysis.cpp:95-163,264-307, include/solver.hpp:30-131,
src/element.cpp:207-274).  This is how the round-3 gmin bug was separated from the generator (tools/dev/k2f_trace/): the
host build matched the oracle through all passes, the GPU build did not.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import netlist_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "circuitsimulator_amd", "csrc")
SCHED = os.path.join(CSRC, "schedules")

HOST_PRELUDE = r"""
#include <cmath>
#include <cstdio>
#include <cstdint>
#define ST_TRAN_NONFINITE 0x0001u
#define ST_TRAN_NONCONV 0x0002u
#define ST_DC_NONCONV 0x0008u
#define ST_SCHED_FAITHFUL 0x0100u
#define Q(k) lds[(k) * 64 + lane]
#define X(i) Q(i)
#define __shared__ static
#define __restrict__
struct Idx { int x; };
static Idx threadIdx{0}, blockIdx{0};
static inline bool __any(bool v) { return v; }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline double csim_mul_rn(double a, double b) { return a * b; }
static inline double csim_add_rn(double a, double b) { return a + b; }
using std::fabs; using std::fmax; using std::fmin; using std::sqrt; using std::isfinite;
"""

HOST_MAIN = r"""
int main(int argc, char** argv)
{
    // stdin: P, N, then P parameters (hex floats); stdout: iters, status, fallback, N hex floats
    int P = 0, N = 0;
    if (std::scanf("%d %d", &P, &N) != 2) return 2;
    static double params[4096];
    for (int i = 0; i < P; ++i) if (std::scanf("%la", &params[i]) != 1) return 2;
    static double xout[2048];
    int iters = 0, viol = 0; unsigned status = 0; unsigned char fallback = 0;
    csim_dc_faithful_kernel(params, 1, xout, &iters, &status, &fallback, &viol, nullptr);
    std::printf("%d %u %d\n", iters, status, (int)fallback);
    for (int i = 0; i < N; ++i) std::printf("%a\n", xout[i]);
    return 0;
}
"""


def _host_binary(codegen, netlist, sched_text, workdir):
    sched = os.path.join(workdir, "s.sched")
    with open(sched, "w") as f:
        f.write(sched_text)
    hip = os.path.join(workdir, "k.hip")
    p = subprocess.run([codegen, netlist, sched, hip], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    lines = open(hip).read().split("\n")
    at = next(i for i, l in enumerate(lines) if l.startswith("csim_dc_faithful_kernel(const double*"))
    assert lines[at - 1].startswith('extern "C" __global__') and lines[at - 2] == "#pragma clang fp contract(off)"
    end = next(i for i in range(at, len(lines)) if lines[i] == "#pragma clang fp contract(fast)")
    body = "void\n" + "\n".join(lines[at:end])
    cpp = os.path.join(workdir, "k_host.cpp")
    with open(cpp, "w") as f:
        f.write(HOST_PRELUDE + body + HOST_MAIN)
    exe = os.path.join(workdir, "k_host")
    p = subprocess.run(["g++", "-O1", "-ffp-contract=off", "-std=c++17", "-w", cpp, "-o", exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return exe


def _run(exe, params, N):
    text = "%d %d\n" % (len(params), N) + "\n".join(float(v).hex() for v in params) + "\n"
    p = subprocess.run([exe], input=text, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    out = p.stdout.split()
    return int(out[0]), int(out[1]), int(out[2]), np.array([float.fromhex(v) for v in out[3:3 + N]])


HOST_MAIN_TRAN = r"""
int main(int argc, char** argv)
{
    // stdin: P, N, nSteps, dt, P parameters, N start values; stdout: total iters, status, fallback, done, N states, nSteps per-step counts
    int P = 0, N = 0, nSteps = 0; double dt = 0.0;
    if (std::scanf("%d %d %d %la", &P, &N, &nSteps, &dt) != 4) return 2;
    static double params[4096], x[2048];
    for (int i = 0; i < P; ++i) if (std::scanf("%la", &params[i]) != 1) return 2;
    for (int i = 0; i < N; ++i) if (std::scanf("%la", &x[i]) != 1) return 2;
    long long iters = 0; unsigned status = 0; unsigned char fallback = 0; int done = 0; int viol[8] = {0};
    static int stepIters[200000];
    csim_tran_faithful_kernel(params, 1, dt, 0LL, (long long)nSteps, nullptr, 0, 1, nullptr, x, &iters, &status, stepIters,
                              &fallback, &done, viol, nullptr, nullptr, nullptr, nullptr);
    std::printf("%lld %u %d %d\n", iters, status, (int)fallback, done);
    for (int i = 0; i < N; ++i) std::printf("%a\n", x[i]);
    for (int s = 0; s < nSteps; ++s) std::printf("%d\n", stepIters[s]);
    return 0;
}
"""

HOST_PRELUDE_TRAN = r"""
using std::sin; using std::fmod;
static inline double clamp01_cg(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }
static inline int __shfl_xor(int v, int) { return v; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
"""


def _host_binary_tran(codegen, netlist, sched_text, workdir):
    sched = os.path.join(workdir, "s.sched")
    with open(sched, "w") as f:
        f.write(sched_text)
    hip = os.path.join(workdir, "k.hip")
    p = subprocess.run([codegen, netlist, sched, hip], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    lines = open(hip).read().split("\n")
    at = next(i for i, l in enumerate(lines) if l.startswith("csim_tran_faithful_kernel(const double*"))
    assert lines[at - 1].startswith('extern "C" __global__') and lines[at - 2] == "#pragma clang fp contract(off)"
    end = next(i for i in range(at, len(lines)) if lines[i] == "#pragma clang fp contract(fast)")
    body = "void\n" + "\n".join(lines[at:end])
    cpp = os.path.join(workdir, "t_host.cpp")
    with open(cpp, "w") as f:
        f.write(HOST_PRELUDE + HOST_PRELUDE_TRAN + body + HOST_MAIN_TRAN)
    exe = os.path.join(workdir, "t_host")
    p = subprocess.run(["g++", "-O1", "-ffp-contract=off", "-std=c++17", "-w", cpp, "-o", exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return exe


@pytest.fixture(scope="module")
def codegen():
    path = os.path.join(CSRC, "build", "csim_codegen")
    if not os.path.exists(path):
        pytest.fail("csim_codegen not built (python -c 'import __graft_entry__ as g; g.build()')")
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    return path


def test_generated_faithful_dc_kernel_on_the_host_equals_the_oracle_bitwise(codegen, tmp_path):
    """dbmixer.sp with its shipped schedules: the nominal instance and two Monte-Carlo instances; 460 / 459 passes."""
    from circuitsimulator_amd import Netlist
    from oracle import binding as orc
    nl = Netlist.from_file(netlist_path("dbmixer.sp"))
    exe = _host_binary(codegen, netlist_path("dbmixer.sp"), open(os.path.join(SCHED, "dbmixer.sched")).read(), str(tmp_path))
    ph = nl.mc_params_host(12345, 0.05, 0, 8)
    cols = [nl.nominal_params] + [ph[:, b] for b in (3, 4)]
    for params in cols:
        it, st, fb, x = _run(exe, params, nl.n_unknowns)
        xo, ito, sto = orc.dc(nl.ir_ptr, nl.n_unknowns, np.ascontiguousarray(params).reshape(-1, 1), 0)
        assert fb == 0                                   # the recorded sequence covered every factorisation
        assert it == ito and st == sto
        assert np.array_equal(x, xo), np.abs(x - xo).max()


def test_generated_faithful_dc_kernel_with_two_sequences_and_a_gate_only_node(codegen, tmp_path):
    """The circuit of tools/fuzz_generated.py seed 10266 (tests/golden/gate_only_node.sp: node n8 is nothing but a MOSFET
    gate, its row is gmin alone) with its two DC pivot sequences, as the GPU planner recorded them."""
    from circuitsimulator_amd import Netlist
    from oracle import binding as orc
    path = netlist_path("gate_only_node.sp")
    nl = Netlist.from_file(path)
    assert nl.n_unknowns == 13
    sched = "0:10,4:12,5:11,8:11\ndc 0:10,4:12,5:11,8:11,9:10,10:12\ndc 0:10,4:12,5:11,8:11\n"
    exe = _host_binary(codegen, path, sched, str(tmp_path))
    ph = nl.mc_params_host(10266, 0.05, 0, 70)
    finished = 0
    for b in (41, 26, 0, 5, 69):
        it, st, fb, x = _run(exe, ph[:, b], nl.n_unknowns)
        if fb:                                           # a third sequence turned up: the engine would replay on the general kernel
            continue
        finished += 1
        xo, ito, sto = orc.dc(nl.ir_ptr, nl.n_unknowns, ph, b)
        assert it == ito and st == sto, b
        assert np.array_equal(x, xo), (b, np.abs(x - xo).max())
    assert finished >= 2


def test_generated_faithful_transient_kernel_on_the_host_equals_the_oracle_bitwise(codegen, tmp_path):
    """csim_tran_faithful_kernel of dbmixer.sp (K1f: what slow steps and near-threshold decisions are redone with) on the host:
    400 backward-Euler steps from the oracle's operating point, two Monte-Carlo instances -- per-step NR counts, status and
    the final state equal the oracle's bit for bit (on the host the kernel's sin() is the oracle's; the device's differs
    in the last bit now and then: DESIGN.md section 8)."""
    from circuitsimulator_amd import Netlist
    from oracle import binding as orc
    nl = Netlist.from_file(netlist_path("dbmixer.sp"))
    exe = _host_binary_tran(codegen, netlist_path("dbmixer.sp"), open(os.path.join(SCHED, "dbmixer.sched")).read(), str(tmp_path))
    ph = nl.mc_params_host(12345, 0.05, 0, 8)
    steps = 400
    for b in (0, 5):
        xdc, _, _ = orc.dc(nl.ir_ptr, nl.n_unknowns, ph, b)
        text = "%d %d %d %s\n" % (ph.shape[0], nl.n_unknowns, steps, float(nl.tstep).hex())
        text += "\n".join(float(v).hex() for v in ph[:, b]) + "\n" + "\n".join(float(v).hex() for v in xdc) + "\n"
        p = subprocess.run([exe], input=text, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        out = p.stdout.split()
        iters, st, fb, done = int(out[0]), int(out[1]), int(out[2]), int(out[3])
        x = np.array([float.fromhex(v) for v in out[4:4 + nl.n_unknowns]])
        per_step = np.array([int(v) for v in out[4 + nl.n_unknowns:4 + nl.n_unknowns + steps]])
        o = orc.tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstep * steps, want_rows=False, want_step_iters=True)
        assert fb == 0 and done == steps, (b, fb, done)
        assert iters == o["iters"] and np.array_equal(per_step, o["step_iters"]), b
        assert (st & ~0x100) == o["status"], (b, hex(st))        # 0x100: "ran on the faithful kernel"
        assert np.array_equal(x, o["x_final"]), (b, np.abs(x - o["x_final"]).max())


def test_generated_faithful_transient_kernel_with_ten_alternatives_on_the_host(codegen, tmp_path):
    """buffer.sp carries ten recorded pivot sequences (one solve body each, tried in order).  At the 3e-11 s step of
    BASELINE configs[1] the nominal circuit stays on recorded sequences: 1000 steps on the host-compiled K1f, per-step NR
    counts and the final state bit for bit the oracle's."""
    from circuitsimulator_amd import Netlist
    from oracle import binding as orc
    nl = Netlist.from_file(netlist_path("buffer.sp"))
    exe = _host_binary_tran(codegen, netlist_path("buffer.sp"), open(os.path.join(SCHED, "buffer.sched")).read(), str(tmp_path))
    params = np.ascontiguousarray(nl.nominal_params).reshape(-1, 1)
    steps, dt = 1000, 3e-11
    xdc, _, _ = orc.dc(nl.ir_ptr, nl.n_unknowns, params, 0)
    text = "%d %d %d %s\n" % (params.shape[0], nl.n_unknowns, steps, float(dt).hex())
    text += "\n".join(float(v).hex() for v in params[:, 0]) + "\n" + "\n".join(float(v).hex() for v in xdc) + "\n"
    p = subprocess.run([exe], input=text, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    out = p.stdout.split()
    iters, st, fb, done = int(out[0]), int(out[1]), int(out[2]), int(out[3])
    x = np.array([float.fromhex(v) for v in out[4:4 + nl.n_unknowns]])
    per_step = np.array([int(v) for v in out[4 + nl.n_unknowns:4 + nl.n_unknowns + steps]])
    o = orc.tran(nl.ir_ptr, nl.n_unknowns, params, 0, dt, dt * steps, want_rows=False, want_step_iters=True)
    assert fb == 0 and done == steps, (fb, done)
    assert iters == o["iters"] and np.array_equal(per_step, o["step_iters"])
    assert np.array_equal(x, o["x_final"]), np.abs(x - o["x_final"]).max()


def test_generated_fast_transient_kernel_on_the_host_against_the_oracle(codegen, tmp_path):
    """The FAST lane-per-instance kernel (csim_tran_sched_kernel: reciprocal pivots, contraction allowed, near-threshold
    guard) compiled for the host with FMA contraction on: not bit-faithful by design -- per-step NR counts equal to the
    oracle's and states within the 1e-9 bar, 400 steps of two dbmixer Monte-Carlo instances, nothing handed over."""
    from circuitsimulator_amd import Netlist
    from oracle import binding as orc
    from conftest import rel_err
    nl = Netlist.from_file(netlist_path("dbmixer.sp"))
    workdir = str(tmp_path)
    hip = os.path.join(workdir, "k.hip")
    p = subprocess.run([codegen, netlist_path("dbmixer.sp"), os.path.join(SCHED, "dbmixer.sched"), hip], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    lines = open(hip).read().split("\n")
    at = next(i for i, l in enumerate(lines) if l.startswith("csim_tran_sched_kernel(const double*"))
    assert lines[at - 1].startswith('extern "C" __global__')
    end = next(i for i in range(at + 1, len(lines)) if lines[i].startswith('extern "C" __global__') or lines[i].startswith("#pragma clang fp"))
    body = "void\n" + "\n".join(lines[at:end])
    main = HOST_MAIN_TRAN.replace("csim_tran_faithful_kernel", "csim_tran_sched_kernel").replace(
        "viol, nullptr, nullptr, nullptr, nullptr);", "viol, nearX, &nearStep, &nearIt, &nearItAfter);").replace(
        "static int stepIters[200000];", "static int stepIters[200000]; static double nearX[2048]; int nearStep = 0, nearIt = 0; long long nearItAfter = 0;")
    rcp = ("static inline double __builtin_amdgcn_rcp(double a) { return 1.0 / a; }\nusing std::fma;\n"
           "static inline double rcp_nr(double a) { const double r = __builtin_amdgcn_rcp(a); const double e = fma(-a, r, 1.0); return fma(fma(e, e, e), r, r); }\n")
    cpp = os.path.join(workdir, "s_host.cpp")
    with open(cpp, "w") as f:
        f.write(HOST_PRELUDE + HOST_PRELUDE_TRAN + rcp + body + main)
    exe = os.path.join(workdir, "s_host")
    p = subprocess.run(["g++", "-O1", "-ffp-contract=fast", "-mfma", "-std=c++17", "-w", cpp, "-o", exe], capture_output=True, text=True)
    if p.returncode != 0 and "mfma" in p.stderr:
        pytest.skip("no FMA on this host")
    assert p.returncode == 0, p.stderr[-2000:]
    ph = nl.mc_params_host(12345, 0.05, 0, 8)
    steps = 400
    for b in (1, 6):
        xdc, _, _ = orc.dc(nl.ir_ptr, nl.n_unknowns, ph, b)
        text = "%d %d %d %s\n" % (ph.shape[0], nl.n_unknowns, steps, float(nl.tstep).hex())
        text += "\n".join(float(v).hex() for v in ph[:, b]) + "\n" + "\n".join(float(v).hex() for v in xdc) + "\n"
        p = subprocess.run([exe], input=text, capture_output=True, text=True)
        if p.returncode != 0 and p.returncode < 0:
            pytest.skip("host cannot run FMA code")
        assert p.returncode == 0, p.stderr
        out = p.stdout.split()
        iters, st, fb, done = int(out[0]), int(out[1]), int(out[2]), int(out[3])
        x = np.array([float.fromhex(v) for v in out[4:4 + nl.n_unknowns]])
        per_step = np.array([int(v) for v in out[4 + nl.n_unknowns:4 + nl.n_unknowns + steps]])
        o = orc.tran(nl.ir_ptr, nl.n_unknowns, ph, b, nl.tstep, nl.tstep * steps, want_rows=False, want_step_iters=True)
        assert fb == 0 and done == steps, (b, fb, done)
        assert iters == o["iters"] and np.array_equal(per_step, o["step_iters"]), b
        assert rel_err(x, o["x_final"]).max() < 1e-9, b


# ------------------------------------------------------------------ random circuits, no GPU anywhere
def _fuzz_netlist(rs):
    """The random circuit of the GPU fuzzers; with CSIM_FUZZ_STRESS=1 (tools/fuzz_host.py --stress) plus a node that is only
    a MOSFET gate, one that is a gate and a capacitor, and one that hangs on capacitors only."""
    from test_gpu_parity import _random_netlist
    text = _random_netlist(rs, rs.randint(3, 25), rs.randint(0, 8))
    if os.environ.get("CSIM_FUZZ_STRESS") == "1":
        nodes = sorted({w for ln in text.splitlines() if ln[:1] in "RCL" for w in ln.split()[1:3] if w.startswith("n")})
        if len(nodes) >= 2:
            d_, s_ = rs.choice(nodes, 2, replace=False)
            extra = ["MG1 %s gx1 %s n %.3ge-6 0.35e-6 2" % (d_, s_, rs.uniform(5, 40)),
                     "MG2 %s gx2 vdd p %.3ge-6 0.35e-6 1" % (rs.choice(nodes), rs.uniform(5, 40)),
                     "CG2 gx2 %s %.4g" % (rs.choice(nodes), 10 ** rs.uniform(-14, -12)),
                     "CX1 cx1 0 %.4g" % 10 ** rs.uniform(-14, -12),
                     "CX2 cx1 %s %.4g" % (rs.choice(nodes), 10 ** rs.uniform(-14, -12))]
            text = text.replace(".MODEL 1", "\n".join(extra) + "\n.MODEL 1", 1)
    return text


def _sched_line(seq):
    return ",".join("%d:%d" % (k, p) for k, p in seq) if seq else "-"


def _extract(lines, name):
    at = next((i for i, l in enumerate(lines) if l.startswith(name + "(const double*")), None)
    if at is None:
        return None
    assert lines[at - 1].startswith('extern "C" __global__') and lines[at - 2] == "#pragma clang fp contract(off)"
    end = next(i for i in range(at, len(lines)) if lines[i] == "#pragma clang fp contract(fast)")
    return "void\n" + "\n".join(lines[at:end])


HOST_MAIN_BOTH = r"""
int main(int argc, char** argv)
{
    // stdin: mode (0 = DC, 1 = transient), P, N, nSteps, dt, P parameters, N start values
    int mode = 0, P = 0, N = 0, nSteps = 0; double dt = 0.0;
    if (std::scanf("%d %d %d %d %la", &mode, &P, &N, &nSteps, &dt) != 5) return 2;
    static double params[4096], x[2048];
    for (int i = 0; i < P; ++i) if (std::scanf("%la", &params[i]) != 1) return 2;
    for (int i = 0; i < N; ++i) if (std::scanf("%la", &x[i]) != 1) return 2;
    unsigned status = 0; unsigned char fallback = 0; int viol[8] = {0};
    if (mode == 0) {
#ifdef HAVE_DC
        int iters = 0;
        csim_dc_faithful_kernel(params, 1, x, &iters, &status, &fallback, viol, nullptr);
        std::printf("%d %u %d 0\n", iters, status, (int)fallback);
#else
        return 3;
#endif
    } else {
        long long iters = 0; int done = 0;
        static int stepIters[200000];
        csim_tran_faithful_kernel(params, 1, dt, 0LL, (long long)nSteps, nullptr, 0, 1, nullptr, x, &iters, &status, stepIters,
                                  &fallback, &done, viol, nullptr, nullptr, nullptr, nullptr);
        std::printf("%lld %u %d %d\n", iters, status, (int)fallback, done);
    }
    for (int i = 0; i < N; ++i) std::printf("%a\n", x[i]);
    return 0;
}
"""


def _call(exe, mode, params, x0, n_steps, dt):
    text = "%d %d %d %d %s\n" % (mode, len(params), len(x0), n_steps, float(dt).hex())
    text += "\n".join(float(v).hex() for v in params) + "\n" + "\n".join(float(v).hex() for v in x0) + "\n"
    p = subprocess.run([exe], input=text, capture_output=True, text=True)
    assert p.returncode == 0, (p.returncode, p.stderr[-500:])
    out = p.stdout.split()
    return int(out[0]), int(out[1]), int(out[2]), int(out[3]), np.array([float.fromhex(v) for v in out[4:4 + len(x0)]])


@pytest.mark.parametrize("seed", [11, 23, 31, 47, 7006, 10266, 20010, 20041])
def test_random_circuit_generated_faithful_kernels_on_the_host(codegen, tmp_path, seed):
    """Generator fuzz without a GPU: a seeded random circuit (the generator of the GPU fuzzers), pivot sequences recorded
    by the ORACLE's own LU (operating point and 50 transient steps of instance 0), csim_codegen, g++ -- the faithful DC
    and transient kernels then reproduce the oracle bit for bit on every instance whose factorisations stay on the
    recorded sequences (an instance that leaves them reports a hand-over, which is the kernel's contract)."""
    rs = np.random.RandomState(seed)
    _check_faithful_on_host(codegen, tmp_path, _fuzz_netlist(rs), seed)


def test_pulse_and_pwl_sources_generated_faithful_kernels_on_the_host(codegen, tmp_path):
    """tests/golden/pulse_pwl.sp: TranWaveform::eval PULSE / PWL (reference include/sim.hpp:80-138) on voltage and current
    sources, through the generated faithful kernels on the host: bit for bit the oracle."""
    _check_faithful_on_host(codegen, tmp_path, open(netlist_path("pulse_pwl.sp")).read(), 99, steps=400)


def _check_faithful_on_host(codegen, tmp_path, text, seed, steps=50):
    from circuitsimulator_amd import Netlist
    from oracle import binding as orc
    nl = Netlist.from_text(text)
    if not nl.has_nonlinear:
        pytest.skip("linear circuit: no K1f / K2f (the linear kernels are fuzzed on the GPU, tools/fuzz_linear.py)")
    path = str(tmp_path / "c.sp")
    with open(path, "w") as f:
        f.write(text)
    N = nl.n_unknowns
    tstep = nl.tstep * float(os.environ.get("CSIM_FUZZ_TSTEP_SCALE", "1"))      # tools/fuzz_host.py --tstep-scale: hard switching
    ph = nl.mc_params_host(seed, 0.05, 0, 4)
    orc.pivot_log(True)
    xdc0, _, _ = orc.dc(nl.ir_ptr, N, ph, 0)
    dc_seqs = orc.pivot_sequences()
    orc.pivot_log(True)
    orc.tran(nl.ir_ptr, N, ph, 0, tstep, tstep * steps, want_rows=False)
    tr_seqs = orc.pivot_sequences()
    orc.pivot_log(False)
    # (the transient log includes the operating point the oracle computes first: harmless extra alternatives)
    sched = "\n".join(_sched_line(q) for q in tr_seqs[:12]) + "\n" + "\n".join("dc " + _sched_line(q) for q in dc_seqs[:12]) + "\n"
    sfile = str(tmp_path / "c.sched")
    with open(sfile, "w") as f:
        f.write(sched)
    hip = str(tmp_path / "k.hip")
    p = subprocess.run([codegen, path, sfile, hip], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    lines = open(hip).read().split("\n")
    tran_body = _extract(lines, "csim_tran_faithful_kernel")
    dc_body = _extract(lines, "csim_dc_faithful_kernel")
    assert tran_body is not None
    cpp = str(tmp_path / "k_host.cpp")
    with open(cpp, "w") as f:
        f.write(HOST_PRELUDE + HOST_PRELUDE_TRAN + "#define S(j) Q(%d + (j))\n" % N + ("#define HAVE_DC 1\n" + dc_body if dc_body else "")
                + "\n" + tran_body + HOST_MAIN_BOTH)
    exe = str(tmp_path / "k_host")
    p = subprocess.run(["g++", "-O1", "-ffp-contract=off", "-std=c++17", "-w", cpp, "-o", exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    n_dc = n_tr = 0
    for b in range(4):
        xo, ito, sto = orc.dc(nl.ir_ptr, N, ph, b)
        if dc_body:
            it, st, fb, _, x = _call(exe, 0, ph[:, b], np.zeros(N), 0, 0.0)
            if not fb:
                n_dc += 1
                assert it == ito and st == sto, (seed, b)
                assert np.array_equal(x, xo), (seed, b, np.abs(x - xo).max())
        o = orc.tran(nl.ir_ptr, N, ph, b, tstep, tstep * steps, want_rows=False)
        it, st, fb, done, x = _call(exe, 1, ph[:, b], xo, steps, tstep)
        if not fb and done == steps and not (st & 0x2 and not o["status"] & 0x2):
            n_tr += 1
            # (the oracle's transient status carries its operating point's flags; the kernel started from that point)
            assert it == o["iters"] and ((st & ~0x100) | sto) == o["status"], (seed, b, it, o["iters"], hex(st), hex(o["status"]))
            assert np.array_equal(x, o["x_final"]), (seed, b, np.abs(x - o["x_final"]).max())
    assert n_tr >= 1 and (n_dc >= 1 or not dc_body), (n_dc, n_tr)


@pytest.mark.parametrize("seed", [11, 47, 7006])
def test_random_circuit_generated_fast_kernel_on_the_host(codegen, tmp_path, seed):
    """The same, for the FAST lane-per-instance transient kernel (reciprocal pivots, contraction, slow-step rule,
    near-threshold guard), compiled with FMA contraction on: per-step NR totals equal to the oracle's and states within
    the 1e-9 bar on every instance the kernel keeps (a step it hands over -- unrecorded pivots, a slow step, a guarded
    decision -- ends the comparison for that instance: the hand-over targets are GPU kernels)."""
    from circuitsimulator_amd import Netlist
    from oracle import binding as orc
    from conftest import rel_err
    from test_gpu_parity import _random_netlist
    rs = np.random.RandomState(seed)
    text = _fuzz_netlist(rs)
    nl = Netlist.from_text(text)
    if not nl.has_nonlinear:
        pytest.skip("linear circuit")
    path = str(tmp_path / "c.sp")
    with open(path, "w") as f:
        f.write(text)
    N, steps = nl.n_unknowns, 50
    tstep = nl.tstep * float(os.environ.get("CSIM_FUZZ_TSTEP_SCALE", "1"))
    ph = nl.mc_params_host(seed, 0.05, 0, 4)
    orc.pivot_log(True)
    orc.tran(nl.ir_ptr, N, ph, 0, tstep, tstep * steps, want_rows=False)
    tr_seqs = orc.pivot_sequences()
    orc.pivot_log(False)
    sfile = str(tmp_path / "c.sched")
    with open(sfile, "w") as f:
        f.write("\n".join(_sched_line(q) for q in tr_seqs[:12]) + "\n")
    hip = str(tmp_path / "k.hip")
    p = subprocess.run([codegen, path, sfile, hip], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    lines = open(hip).read().split("\n")
    at = next(i for i, l in enumerate(lines) if l.startswith("csim_tran_sched_kernel(const double*"))
    end = next(i for i in range(at + 1, len(lines)) if lines[i].startswith('extern "C" __global__') or lines[i].startswith("#pragma clang fp"))
    body = "void\n" + "\n".join(lines[at:end])
    main = HOST_MAIN_TRAN.replace("csim_tran_faithful_kernel", "csim_tran_sched_kernel").replace(
        "viol, nullptr, nullptr, nullptr, nullptr);", "viol, nearX, &nearStep, &nearIt, &nearItAfter);").replace(
        "static int stepIters[200000];", "static int stepIters[200000]; static double nearX[2048]; int nearStep = 0, nearIt = 0; long long nearItAfter = 0;")
    rcp = ("static inline double __builtin_amdgcn_rcp(double a) { return 1.0 / a; }\nusing std::fma;\n"
           "static inline double rcp_nr(double a) { const double r = __builtin_amdgcn_rcp(a); const double e = fma(-a, r, 1.0); return fma(fma(e, e, e), r, r); }\n")
    cpp = str(tmp_path / "s_host.cpp")
    with open(cpp, "w") as f:
        f.write(HOST_PRELUDE + HOST_PRELUDE_TRAN + "#define S(j) Q(%d + (j))\n" % N + rcp + body + main)
    exe = str(tmp_path / "s_host")
    p = subprocess.run(["g++", "-O1", "-ffp-contract=fast", "-mfma", "-std=c++17", "-w", cpp, "-o", exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    kept = 0
    for b in range(4):
        xo, _, sto = orc.dc(nl.ir_ptr, N, ph, b)
        o = orc.tran(nl.ir_ptr, N, ph, b, tstep, tstep * steps, want_rows=False, want_step_iters=True)
        text_in = "%d %d %d %s\n" % (ph.shape[0], N, steps, float(tstep).hex())
        text_in += "\n".join(float(v).hex() for v in ph[:, b]) + "\n" + "\n".join(float(v).hex() for v in xo) + "\n"
        p = subprocess.run([exe], input=text_in, capture_output=True, text=True)
        if p.returncode < 0:
            pytest.skip("host cannot run FMA code")
        assert p.returncode == 0, p.stderr
        out = p.stdout.split()
        iters, st, fb, done = int(out[0]), int(out[1]), int(out[2]), int(out[3])
        x = np.array([float.fromhex(v) for v in out[4:4 + N]])
        per_step = np.array([int(v) for v in out[4 + N:4 + N + steps]])
        # the steps the kernel completed itself must carry the oracle's pass counts, whatever happened afterwards
        assert np.array_equal(per_step[:done], o["step_iters"][:done]), (seed, b, done)
        if fb == 0 and done == steps:
            kept += 1
            assert iters == o["iters"], (seed, b)
            assert rel_err(x, o["x_final"]).max() < 1e-9, (seed, b, rel_err(x, o["x_final"]).max())
    print("seed %d: %d of 4 instances kept by the fast kernel" % (seed, kept))
