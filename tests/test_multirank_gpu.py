"""The multi-rank path with the REAL engine: two fresh child processes (one per rank, started with
subprocess; nothing here re-executes a process that has touched the GPU) run `bench.py --gpus 2`
sharing the box's one GPU (CSIM_SHARE_GPU=1; gloo for the collectives because RCCL refuses two
ranks on one device).  Everything else is the measured path: rendezvous, netlist broadcast, sharded
parameter regeneration from (seed, global index), barriers, max-over-ranks timing, all-gather of
the probe voltages, all-reduce of the counters.  SURVEY.md 8(e): the G-way result must equal the
1-way result of the same global batch bitwise."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(args, env_extra, timeout=600):
    env = dict(os.environ, **env_extra)
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _finish(proc, timeout=600):
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, err = proc.communicate()
        pytest.fail("bench.py timed out\n" + err[-2000:])
    assert proc.returncode == 0, err[-4000:]
    return out, err


def test_two_ranks_share_the_gpu_and_equal_one_rank(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    B = 256
    common = ["--steps", "2", "--warmup", "1", "--tsteps", "40", "--no-cpu", "--large-batch", "0"]
    two = str(tmp_path / "two.npz")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CSIM_SHARE_GPU="1", CSIM_DIST_BACKEND="gloo")
        procs.append(_bench(["--gpus", "2", "--batch", str(B), "--dump-gathered", two] + common, env))
    outs = [_finish(p) for p in procs]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]   # rank 0 alone prints
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak"
    assert rec["gathered_shape"][1] == 2 * B
    assert rec["config"]["kernel"] == "scheduled" and rec["config"]["flagged_instances"] == 0

    one = str(tmp_path / "one.npz")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", str(2 * B),
                          "--dump-gathered", one] + common, env=env, cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True)
    out1, _ = _finish(p)
    rec1 = json.loads([l for l in out1.splitlines() if l.startswith("{")][0])
    a, b = np.load(two), np.load(one)
    assert a["gathered"].shape == (len(rec["gathered_shape"]) and rec["gathered_shape"][0], 2 * B)
    assert np.array_equal(a["gathered"], b["gathered"])            # bitwise: same kernel, same per-instance data
    assert float(a["total_iters"]) == float(b["total_iters"])     # all-reduced NR count == one-rank count
    assert rec1["n_gpus"] == 1


def test_bare_gpus_2_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher environment (how a driver may start it): the process starts two
    fresh child ranks itself, relays rank 0's single JSON line and exits 0."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(CSIM_SHARE_GPU="1", CSIM_DIST_BACKEND="gloo")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "128", "--steps", "2",
                          "--warmup", "1", "--tsteps", "30", "--no-cpu", "--large-batch", "0"], env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    out, err = _finish(p)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["gathered_shape"][1] == 256 and rec["dist_backend"] == "gloo"
    assert rec["config"]["flagged_instances"] == 0


def test_collectives_run_through_rccl():
    """One rank with a forced process group on the nccl backend (= RCCL on ROCm): the netlist broadcast, the
    all-gather of the probe voltages and the int64 all-reduce of the counters go through RCCL on this box."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(CSIM_FORCE_DIST="1", CSIM_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "256", "--steps", "2",
                          "--warmup", "1", "--tsteps", "30", "--no-cpu", "--large-batch", "0"], env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    out, err = _finish(p)
    rec = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert rec["dist_backend"] == "nccl" and rec["n_gpus"] == 1
    assert rec["gathered_shape"][1] == 256 and rec["value"] > 0
    assert isinstance(rec["config"]["flagged_instances"], int) and rec["config"]["flagged_instances"] == 0
