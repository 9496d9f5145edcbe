"""Multi-GPU sharding of the instance batch (one process per GPU).

The hot path has NO exchange step: circuit instances are independent, so the
batch is cut into contiguous instance ranges and every rank solves its own.
Collectives happen once per run, never per time step:
  1. broadcast of the netlist text from rank 0 (a few hundred bytes; every
     rank then parses and flattens it itself),
  2. no scatter -- each rank regenerates its instances' parameters from
     (seed, global instance index, slot),
  3. all-gather of the per-instance results (probe voltages / final state),
  4. all-reduce(sum) of the NR-iteration and status counters.
Backend "nccl" is RCCL over xGMI on the GPU box; "gloo" is used by the CPU
tests (tests/test_shard_gloo.py).
"""
import os

import numpy as np


def dist_env():
    """(rank, local_rank, world_size) from the launcher's environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(total, rank, world):
    """Contiguous instance range [lo, hi) of `rank`; sizes differ by at most one."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def init_process_group(backend=None):
    """Initialise torch.distributed from the environment if WORLD_SIZE > 1."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    # a single rank needs no process group; CSIM_FORCE_DIST=1 creates one anyway so that the
    # RCCL calls can be rehearsed on a one-GPU box
    if dist.is_initialized() or (world <= 1 and os.environ.get("CSIM_FORCE_DIST") != "1"):
        return rank, local_rank, world
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        if world > 1:
            raise RuntimeError("WORLD_SIZE > 1 needs MASTER_PORT in the environment (the launcher sets it)")
        import socket                         # a forced single-rank group: any free port will do
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        s.close()
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def _dev(device):
    """Device the collectives' tensors live on: the caller's, except under gloo (CPU tensors)."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return torch.device(device) if device is not None else torch.device("cpu")


def broadcast_netlist_text(text, src=0, device=None):
    """Rank `src` sends the netlist text; every rank returns the same str."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return text
    dev = _dev(device)
    rank = dist.get_rank()
    if rank == src:
        payload = np.frombuffer(text.encode("utf-8"), dtype=np.uint8).copy()
        n = torch.tensor([payload.size], dtype=torch.int64, device=dev)
    else:
        payload = None
        n = torch.zeros(1, dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    buf = torch.zeros(int(n.item()), dtype=torch.uint8, device=dev)
    if rank == src:
        buf.copy_(torch.from_numpy(payload))
    dist.broadcast(buf, src=src)
    return bytes(buf.cpu().numpy().tobytes()).decode("utf-8")


def all_gather_instances(local, total, device=None):
    """Concatenate per-rank tables along the LAST axis (instances) in rank order.

    local: tensor [..., B_local]; ranks may hold different B_local (shard_range).
    Returns a tensor [..., total] on every rank.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    sizes = [shard_range(total, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    lead = tuple(local.shape[:-1])
    home = local.device
    if dist.get_backend() == "gloo":
        local = local.cpu()
    pad = torch.zeros(lead + (bmax,), dtype=local.dtype, device=local.device)
    pad[..., :local.shape[-1]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad.contiguous())
    return torch.cat([p[..., :hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=-1).to(home)


def all_reduce_sum(value, device=None):
    """Sum of a python int/float (or tensor) over all ranks."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    if torch.is_tensor(value):
        home = value.device
        t = value.to(_dev(home))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.to(home)
    t = torch.tensor([value], dtype=torch.float64, device=_dev(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.item()


def all_reduce_sum_int(value, device=None):
    """Sum of a python int over all ranks as int64 (NR-iteration and status counters)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=_dev(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def backend_name():
    """"nccl" (= RCCL), "gloo", or "none" when this run has no process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_backend()
    return "none"


def all_reduce_max(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_dev(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
