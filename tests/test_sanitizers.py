"""CPU-side code under AddressSanitizer + UndefinedBehaviorSanitizer (GPU ASan is not available on this
pool): front-end, assembly plan, the three kernel generators with the shipped schedules, the sixteen-lane plan's host
interpreter, and the CPU oracle including its threaded batch driver (tests/sanitize/)."""
import os
import subprocess

import pytest

from conftest import ROOT, netlist_path

CSRC = os.path.join(ROOT, "circuitsimulator_amd", "csrc")


@pytest.fixture(scope="module")
def checker():
    here = os.path.join(ROOT, "tests", "sanitize")
    p = subprocess.run(["make", "-s", "-C", here], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    return os.path.join(here, "_build", "csim_sanitize_check")


@pytest.mark.parametrize("name", ["buffer", "dbmixer"])
def test_cpu_side_code_is_clean_under_asan_and_ubsan(checker, name):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([checker, netlist_path(name + ".sp"), os.path.join(CSRC, "schedules", name + ".sched")],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.returncode, p.stdout[-500:], p.stderr[-3000:])
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    assert "oracle: dc" in p.stdout


def test_linear_circuit_generators_under_the_sanitizers(checker, tmp_path):
    from circuitsimulator_amd.workloads import rc_ladder_netlist
    net = tmp_path / "ladder.sp"
    net.write_text(rc_ladder_netlist(64))
    sched = tmp_path / "ladder.sched"
    sched.write_text("0:64\n")
    p = subprocess.run([checker, str(net), str(sched)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.returncode, p.stderr[-3000:])
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
