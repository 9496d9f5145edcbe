"""Multi-rank plumbing on CPU (gloo, world_size 2): netlist broadcast, instance sharding,
gather and counter all-reduce.  The solve itself needs a GPU; here a deterministic
function of the parameter table stands in for it, so shard == whole-batch can be checked."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from circuitsimulator_amd import shard
from conftest import ROOT, netlist_path


def test_shard_range_partitions():
    for total in (0, 1, 7, 4096, 65536, 1000001):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and b >= a
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_solve(params):
    """stand-in for the GPU solve: any deterministic per-instance function"""
    return np.stack([params.sum(axis=0), (params ** 2).sum(axis=0)])


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from circuitsimulator_amd import Netlist
    assert shard.init_process_group("gloo") == (rank, rank, world)
    text = open(netlist_path("dbmixer.sp")).read() if rank == 0 else "garbage on rank %d" % rank
    text = shard.broadcast_netlist_text(text, src=0)
    nl = Netlist.from_text(text)
    assert nl.n_unknowns == 31
    lo, hi = shard.shard_range(total, rank, world)
    params = nl.mc_params_host(12345, 0.05, lo, hi - lo)      # regenerated, not scattered
    local = torch.from_numpy(_fake_solve(params))
    full = shard.all_gather_instances(local, total)
    iters = shard.all_reduce_sum(float(hi - lo))
    tmax = shard.all_reduce_max(float(rank + 1))
    shard.barrier()
    np.save(os.path.join(out_dir, "r%d.npy" % rank), full.numpy())
    assert iters == total and tmax == world
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [64, 101])
def test_two_ranks_equal_single_rank(tmp_path, total):
    from circuitsimulator_amd import Netlist
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    nl = Netlist.from_file(netlist_path("dbmixer.sp"))
    whole = _fake_solve(nl.mc_params_host(12345, 0.05, 0, total))
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "r%d.npy" % r))
        assert got.shape == (2, total)
        assert np.array_equal(got, whole)          # G-way result == 1-way result, bitwise


def test_bare_multi_gpu_bench_is_its_own_launcher():
    """`python bench.py --gpus 2` without a launcher environment must not exit 2 ("use torch.distributed.run"):
    it starts one child per rank itself.  On a box without a GPU the children stop with "no GPU visible"
    (status 3) and the parent reports which ranks failed; it never imports torch or touches a device itself."""
    import subprocess
    import sys
    from conftest import ROOT, has_gpu
    if has_gpu():
        pytest.skip("covered by tests/test_multirank_gpu.py on a GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 1, (p.returncode, p.stderr[-500:])
    # (the first rank to fail stops the run: the other one is reported with 3 too, or as killed)
    assert "ranks failed" in p.stderr and "no GPU visible" in p.stderr and ", 3)" in p.stderr
    assert p.stdout.strip() == ""
