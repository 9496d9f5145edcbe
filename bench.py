#!/usr/bin/env python3
"""bench.py -- NR-iteration x instances / second of the batched transient solve.

    python bench.py --gpus N --steps K --warmup W
    N > 1: either under a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (python -m
    torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...), or bare: the process then
    starts N child processes of itself, one per GPU, with that environment (it makes no GPU call of its own),
    relays rank 0's JSON line and exits non-zero if any rank failed.

Workload (BASELINE.json configs[2], the configuration the metric is quoted
on): tests/dbmixer.sp transient, B = 4096 Monte-Carlo-perturbed instances per
GPU (sigma = 5 %, seed 12345, instance 0 nominal), every
instance started from its own DC operating point.  One bench "step" = one
launch of the transient kernel that advances ALL instances of the rank by
`--tsteps` backward-Euler time steps (tstep = 1e-13 s from the netlist), with
parameters, state and counters resident in HBM.  Weak scaling: every GPU gets
its own B instances (global instance index = rank*B + i), no data-path
collective; the netlist is broadcast once and results are gathered once,
outside the timed region.

value = NR iterations summed over all instances of all ranks in the K timed
steps / max-over-ranks wall time of those steps.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6 # MI355X FP64 vector peak (256 CUs x 4 SIMDs x 16 FMA lanes x 2 flop x 2.4 GHz)
N_SIMDS = 1024                 # 256 CUs x 4 SIMDs
PEAK_CLOCK_HZ = 2.4e9
VALU_ISSUE_CYCLES = 4          # one wave64 VALU instruction (FP64 FMA: 16 lanes/clk/SIMD) holds a SIMD's issue for 4 cycles


def dense_flops(n):
    """LU + substitution flops of the reference per NR iteration."""
    return n * (n - 1) // 2 + (n - 1) * n * (2 * n - 1) // 3 + 2 * n * (n - 1) + n


def algorithmic_bytes_per_iter(N):
    """Dense system the reference materialises per NR iteration + x in/out."""
    return 8 * (N * N + 3 * N)


def profiled_counters(nl, kernel, B):
    """Counters of this kernel at this batch size from the committed rocprofv3 PMC passes
    (profiles/kernel_counters.json, written by tools/summarize_profile.py; FETCH_SIZE / WRITE_SIZE /
    SQ_INSTS_VALU collected in separate --pmc passes, per NR iteration x instance so that they carry
    over to any launch length).  None when this exact (circuit, kernel, batch) was not profiled."""
    path = os.path.join(HERE, "profiles", "kernel_counters.json")
    if not os.path.exists(path):
        return None
    try:
        table = json.load(open(path))
    except Exception:
        return None
    return table.get("N%d|%s|B%d" % (nl.n_unknowns, kernel, B))


def executed_flops_per_unit(eng, nl):
    """FP64 operations one NR iteration of one instance EXECUTES in the generated kernel: the solve's
    counts from the generator (csim_engine_sched_info; an FMA = 2, a Newton-refined reciprocal = 1 + 4 FMA)
    plus the damped update and norm (5 per unknown) -- device evaluation and assembly (a few hundred more)
    are not counted, so this is a lower bound."""
    info = eng.sched_info
    if not info or not info["ops"]:
        return None
    o = info["ops"]
    return 2 * o.get("fma", 0) + o.get("mul", 0) + o.get("addsub", 0) + 9 * o.get("recip", 0) + 5 * nl.n_unknowns


def valu_roof(units_per_s, counters, B, lanes_per_instance, flops_exec):
    """The roof that binds these kernels: FP64 VALU issue.  achieved_frac = wave-level VALU instructions
    the kernel issues per second x 4 cycles each / (1024 SIMDs x 2.4 GHz); waves = what the batch can keep
    resident (the lane-per-instance kernel needs 64 instances per wave, one wave per SIMD)."""
    waves = -(-B * lanes_per_instance // 64)
    rec = {"bound": "fp64 VALU issue (%d cycles per wave instruction, %d SIMDs, %.1f GHz)"
                    % (VALU_ISSUE_CYCLES, N_SIMDS, PEAK_CLOCK_HZ / 1e9),
           "lanes_per_instance": lanes_per_instance,
           "waves": waves, "simd_occupancy": min(1.0, waves / N_SIMDS),
           "executed_flops_per_unit": flops_exec,
           "executed_tflops": units_per_s * flops_exec / 1e12 if flops_exec else None,
           "valu_wave_insts_per_unit": None, "achieved_frac": None, "source": None}
    if counters and counters.get("valu_wave_insts_per_unit"):
        ipu = counters["valu_wave_insts_per_unit"]
        rec["valu_wave_insts_per_unit"] = ipu
        rec["achieved_frac"] = units_per_s * ipu * VALU_ISSUE_CYCLES / (N_SIMDS * PEAK_CLOCK_HZ)
        rec["source"] = counters.get("source")
    return rec


def self_launch(n):
    """`python bench.py --gpus N` started bare: one fresh child process per rank (nothing re-executes a process
    that touched the GPU: this parent never imports torch).  Rank 0's stdout is relayed; every child's stderr
    goes to ours.  Returns the exit status for the parent."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread so that the loop below can watch all ranks: when one fails the
    # others would wait in a collective for ever, so they are stopped (the exact PIDs started here)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    sys.stdout.write(b"".join(chunks).decode("utf-8", "replace"))
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print("bench.py: ranks failed (rank, exit status): %s" % bad, file=sys.stderr)
        return 1
    return 0


def cpu_share():
    """CPUs this process may USE: its affinity mask, cut down to the cgroup's CPU quota when there is one (a GPU box
    gives one GPU's job a share of the host: 256 cores visible, 16 cores' worth of time -- 256 threads then run slower
    than 16)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:                                                   # cgroup v2
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:                                               # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.999)))
    return n, quota


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(nl, params_host, n_inst, tstep, n_tsteps, threads=1):
    """Oracle (CPU restatement) on a bounded sample of the same workload: instances 0..n_inst-1,
    one instance per call, `threads` host threads over instances."""
    from oracle import binding as orc

    def one(b):
        return orc.tran(nl.ir_ptr, nl.n_unknowns, params_host, b, tstep, tstep * n_tsteps, want_rows=False)["iters"]

    t0 = time.perf_counter()
    if threads <= 1:
        iters = sum(one(b) for b in range(n_inst))
    else:
        # POSIX threads inside the oracle library, one instance at a time from a shared counter (a Python thread
        # pool of 256 ctypes callers spends its time on the interpreter lock: measured 7.7x on 256 cores)
        per, _ = orc.tran_batch_mt(nl.ir_ptr, params_host, 0, n_inst, tstep, tstep * n_tsteps, threads)
        iters = int(per.sum())
    dt = time.perf_counter() - t0
    return iters, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--tsteps", type=int, default=6000,
                    help="time steps per bench step (one csim_tran_batch_dev call); 6000 x (5 + 20) steps = three times "
                         "the netlist's 50 000-step run (~2 s of timed GPU work at B = 4096 with 20 timed steps)")
    ap.add_argument("--lanes", type=int, default=0, choices=[0, 1, 4, 16],
                    help="scheduled kernel: lanes per instance (0 = the engine picks by batch size)")
    ap.add_argument("--dump-gathered", default="",
                    help="rank 0 writes the gathered probe voltages and per-rank NR totals to this .npz (tests)")
    ap.add_argument("--netlist", default=os.path.join(HERE, "tests", "golden", "dbmixer.sp"))
    ap.add_argument("--ladder", type=int, default=0,
                    help="use the synthetic RC ladder with this many nodes instead of --netlist (configs[3]: 256)")
    ap.add_argument("--no-refine", dest="refine", action="store_false",
                    help="do not re-specialise the generated kernels with the pivot sequences of flagged instances")
    ap.add_argument("--no-jit", action="store_true", help="do not JIT-specialise a netlist without a prebuilt kernel")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--sigma", type=float, default=0.05)
    ap.add_argument("--kernel", default="auto", choices=["auto", "general", "scheduled", "faithful"])
    ap.add_argument("--cpu-iters", type=float, default=3.0e6, help="approx. NR iterations of the CPU sample (~15 s)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the all-cores CPU leg (0 = every core this process may run on; 1 = skip the leg)")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="target run time of the all-cores CPU leg")
    ap.add_argument("--gen-opts", default="",
                    help="re-generate the circuit's kernels with these generator options (engine option jit_gen_opts, "
                         "e.g. near_band=0) before the run: A/B measurements of generator choices")
    ap.add_argument("--overlap-gather", action="store_true",
                    help="copy every step's probe voltages to pinned host memory on a second stream while the next step "
                         "runs (with --async the host never waits inside a step: launch k+1 is enqueued under copy k)")
    ap.add_argument("--async", dest="async_calls", action="store_true",
                    help="engine option hybrid_sync=0: the transient calls only enqueue, the host never waits inside them")
    ap.add_argument("--large-batch", type=int, default=65536,
                    help="also time this many instances per GPU (one wave per SIMD needs >= 65536); 0 = skip")
    ap.add_argument("--mid-batch", type=int, default=16384,
                    help="also time this many instances per GPU (the four-lanes-per-instance kernel's range); 0 = skip")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:       # started bare: be the launcher (no GPU call in this process)
        sys.exit(self_launch(args.gpus))

    import torch
    from circuitsimulator_amd import Engine, Netlist, shard

    rank, local_rank, world = shard.dist_env()
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the engine has no CPU path", file=sys.stderr)
        sys.exit(3)
    # Rehearsal on a one-GPU box (not the measured configuration): CSIM_SHARE_GPU=1 maps every rank to device 0
    # and CSIM_DIST_BACKEND=gloo replaces RCCL, which refuses two ranks on one device.  The data path is the same.
    if os.environ.get("CSIM_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    shard.init_process_group(os.environ.get("CSIM_DIST_BACKEND", "nccl"))

    # ---- netlist: rank 0 reads, RCCL broadcast, every rank parses ------------
    t0 = time.perf_counter()
    if args.ladder:
        from circuitsimulator_amd.workloads import rc_ladder_netlist
        text = rc_ladder_netlist(args.ladder) if rank == 0 else ""
    else:
        text = open(args.netlist).read() if rank == 0 else ""
    text = shard.broadcast_netlist_text(text, src=0, device=dev)
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t0) * 1e3
    nl = Netlist.from_text(text)
    eng = Engine(nl, local_rank)
    if args.kernel != "auto":
        eng.set_kernel(args.kernel)
    if args.lanes:
        eng.set_option("lanes_per_instance", args.lanes)
    if args.async_calls:
        eng.set_option("hybrid_sync", 0)
    if args.gen_opts:
        eng.set_option("jit_gen_opts", args.gen_opts)          # also what a later JIT of a netlist without a shipped library uses
        if eng.tran_kernel != "general":
            sched, dc_sched = eng.loaded_schedules()
            eng.jit_with_schedules(sched, dc_sched)
    N, B, S = nl.n_unknowns, args.batch, args.tsteps
    tstep = nl.tstep

    # ---- per-rank shard of the global batch, regenerated from (seed, b, slot)
    params = eng.mc_params(args.seed, args.sigma, rank * B, B)
    if eng.tran_kernel == "general" and args.kernel != "general" and not args.no_jit:
        try:                                  # setup, untimed: plan + generate + hipcc + load
            eng.jit_scheduled(params, plan_steps=20)
        except Exception as e:                # no hipcc on the box, circuit too large, ...: stay general
            print("bench.py: JIT specialisation unavailable (%s); using the general kernel" % e, file=sys.stderr)
    x, dc_it, status = eng.dc(params)
    refined, probe_steps = 0, 0
    if args.refine and eng.tran_kernel == "scheduled" and args.kernel in ("auto", "scheduled") and not args.no_jit:
        # setup, untimed: one probe step; instances it flags used pivot sequences the generated kernels do not
        # carry and would finish on the general kernel (a long tail) -- record those sequences and re-specialise
        xp_, itp_, stp_ = x.clone(), torch.zeros(B, dtype=torch.int64, device=dev), status.clone()
        eng.tran(params, xp_, tstep, 0, S, itp_, stp_)
        probe_steps = 1
        torch.cuda.synchronize()
        if int((stp_ & 0x20).ne(0).sum().item()):
            try:
                refined = eng.refine_schedules(params, stp_, tstep, n_steps=min(S, 300))
            except Exception as e:
                print("bench.py: schedule refinement unavailable (%s)" % e, file=sys.stderr)
        del xp_, itp_, stp_
    iters = torch.zeros(B, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    step_idx = 0
    for _ in range(args.warmup):
        eng.tran(params, x, tstep, step_idx, S, iters, status)
        step_idx += S
    torch.cuda.synchronize()
    shard.barrier()
    torch.cuda.synchronize()

    it_before = iters.clone()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    copy_stream = torch.cuda.Stream() if args.overlap_gather else None
    probe_rows = nl.probes if nl.probes else [0]
    host_probe = [torch.empty((len(probe_rows), B), dtype=torch.float64).pin_memory() for _ in range(args.steps)] \
        if args.overlap_gather else []
    staged = [torch.empty((len(probe_rows), B), dtype=torch.float64, device=dev) for _ in range(2)] if args.overlap_gather else []
    t_start = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        eng.tran(params, x, tstep, step_idx, S, iters, status)
        ev[k][1].record()
        if copy_stream is not None:
            # step k's probe voltages: snapshot on the compute stream (x is overwritten by step k+1), D2H on the copy
            # stream under step k+1
            if k >= 2:
                torch.cuda.current_stream().wait_stream(copy_stream)      # the snapshot buffer is free again
            staged[k & 1].copy_(x[probe_rows, :])
            copy_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(copy_stream):
                host_probe[k].copy_(staged[k & 1], non_blocking=True)
        step_idx += S
    torch.cuda.synchronize()
    shard.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_start

    kern_ms = [a.elapsed_time(b) for a, b in ev]
    local_iters = int((iters - it_before).sum().item())
    total_iters = shard.all_reduce_sum_int(local_iters, device=dev)          # int64
    wall_max = shard.all_reduce_max(wall, device=dev)
    n_bad = int((status & 0xA7).ne(0).sum().item())           # non-finite / non-converged / tiny pivot / fallback
    n_bad = shard.all_reduce_sum_int(n_bad, device=dev)

    # ---- gather of node voltages (probes of the netlist: V(102), V(103)) -----
    t0 = time.perf_counter()
    probes = nl.probes if nl.probes else [0]
    local_v = x[probes, :].contiguous()
    all_v = shard.all_gather_instances(local_v, B * world, device=dev)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t0) * 1e3

    # ---- second leg: a batch that gives every SIMD a wave (same circuit, same kernels) ------
    def second_leg(BL):
        pl = eng.mc_params(args.seed, args.sigma, rank * BL, BL)
        xl, _, stl = eng.dc(pl)
        itl = torch.zeros(BL, dtype=torch.int64, device=dev)
        eng.tran(pl, xl, tstep, 0, S, itl, stl)                  # warm-up step
        torch.cuda.synchronize()
        shard.barrier()
        before = itl.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nl_steps = max(1, args.steps // 2)
        t0 = time.perf_counter()
        e0.record()
        for k in range(nl_steps):
            eng.tran(pl, xl, tstep, S * (1 + k), S, itl, stl)
        e1.record()
        torch.cuda.synchronize()
        shard.barrier()
        wl = shard.all_reduce_max(time.perf_counter() - t0, device=dev)
        il = shard.all_reduce_sum_int(int((itl - before).sum().item()), device=dev)
        return {"batch_per_gpu": BL, "value": il / wl, "steps": nl_steps,
                "kernel_avg_ms": e0.elapsed_time(e1) / nl_steps,
                "flagged_instances": int((stl & 0xA7).ne(0).sum().item())}

    mid = second_leg(args.mid_batch) if args.mid_batch and args.mid_batch != B else None
    large = second_leg(args.large_batch) if args.large_batch and args.large_batch != B else None

    if args.dump_gathered and rank == 0:
        np.savez(args.dump_gathered, gathered=all_v.cpu().numpy(), total_iters=np.int64(total_iters))

    if rank == 0:
        value = total_iters / wall_max
        avg_kern_s = float(np.mean(kern_ms)) * 1e-3
        iters_per_launch = local_iters / args.steps
        abytes = algorithmic_bytes_per_iter(N)
        achieved = iters_per_launch * abytes / avg_kern_s / 1e9
        kernel = eng.tran_kernel
        lanes = eng.lanes_for_batch(B) if kernel == "scheduled" else (1 if kernel == "faithful" else 64)
        ckey = kernel if lanes in (1, 64) else "%s%d" % (kernel, lanes)
        flagged_mask = 0xA7
        counters = profiled_counters(nl, ckey, B)
        flops_exec = executed_flops_per_unit(eng, nl) if kernel == "scheduled" else None
        frac = achieved / HBM_PEAK_GBS
        rec = {
            "metric": "NR-iteration x instances / sec (transient)",
            "value": value,
            "unit": "NR-iter*inst/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s transient (N=%d unknowns), batch=%d MC-perturbed instances per GPU "
                            "(sigma=%g, seed=%d), %d time steps of %.3g s per bench step"
                            % ("synthetic RC ladder" if args.ladder else "tests/" + os.path.basename(args.netlist),
                               N, B, args.sigma, args.seed, S, tstep),
                "batch_per_gpu": B,
                "time_steps_per_step": S,
                "kernel": kernel,
                "lanes_per_instance": lanes,
                "nr_iters_per_step": iters_per_launch,
                "flagged_instances": n_bad, "refined_schedules": refined, "setup_probe_steps": probe_steps,
                "generator_options": args.gen_opts,
            },
            # SURVEY.md 8(d) accounting: the bytes of the DENSE system the reference materialises per NR
            # iteration, 8(N^2+3N), not bytes this kernel moves (it keeps the sparse system on chip; what it
            # really moves is `traffic`).  A fraction above 1 is therefore possible and says only that the
            # dense work was not done; the roof that binds is `roofline_valu`.
            "roofline": {
                "bound": "hbm",
                "accounting": "dense_equivalent_bytes (SURVEY.md 8d: 8*(N^2+3N) per NR iteration x instance)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": frac,
                "artefact": frac > 1.0,
                "traffic": (counters["hbm_bytes_per_unit"] * iters_per_launch) if counters and counters.get("hbm_bytes_per_unit") else None,
                "traffic_source": counters.get("source") if counters else None,
                "algorithmic_bytes_per_unit": abytes,
                "kernel_avg_ms": avg_kern_s * 1e3,
            },
            "roofline_valu": valu_roof(iters_per_launch / avg_kern_s, counters, B, 1 if lanes == 64 else lanes, flops_exec)
                             if kernel == "scheduled" else None,
            # measured HBM bytes against the HBM roof (the figure `roofline` cannot give: its bytes are the dense system's)
            "hbm_measured": ({"bytes_per_launch": counters["hbm_bytes_per_unit"] * iters_per_launch,
                              "achieved_GBs": counters["hbm_bytes_per_unit"] * iters_per_launch / avg_kern_s / 1e9,
                              "frac_of_peak": counters["hbm_bytes_per_unit"] * iters_per_launch / avg_kern_s / 1e9 / HBM_PEAK_GBS,
                              "source": counters.get("source")}
                             if counters and counters.get("hbm_bytes_per_unit") else None),
            # the dense LU + substitution flops the reference spends per NR iteration,
            # n(n-1)/2 + (n-1)n(2n-1)/3 + 2n(n-1) + n, against the FP64 vector peak (the kernels execute far fewer:
            # structural zeros are never touched)
            "fp64_dense_equivalent": {
                "flops_per_unit": dense_flops(nl.n_unknowns),
                "achieved_tflops": (iters_per_launch / avg_kern_s) * dense_flops(nl.n_unknowns) / 1e12,
                "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
            },
            "dist_backend": shard.backend_name(),
            "hybrid_sync": 0 if args.async_calls else 1,
            "overlap_gather": bool(args.overlap_gather),
            "near_threshold": {"verified": eng.stat("near_verified"), "rolled_back": eng.stat("near_rolled_back")},
            "netlist_bcast_ms": bcast_ms,
            "result_gather_ms": gather_ms,
            "gathered_shape": list(all_v.shape),
        }
        for key, large in (("mid_batch", mid), ("large_batch", large)):
            if large is None:
                continue
            lf = large["value"] * abytes / 1e9 / HBM_PEAK_GBS
            large["roofline_frac"] = lf
            large["roofline_artefact"] = lf > 1.0
            ll = eng.lanes_for_batch(large["batch_per_gpu"]) if kernel == "scheduled" else 64
            lkey = kernel if ll in (1, 64) else "%s%d" % (kernel, ll)
            large["lanes_per_instance"] = ll
            if kernel == "scheduled":
                large["roofline_valu"] = valu_roof(large["value"] / world, profiled_counters(nl, lkey, large["batch_per_gpu"]),
                                                   large["batch_per_gpu"], ll, flops_exec)
            rec[key] = large
        if world == 1 and not args.no_cpu:
            # CPU baseline: the oracle (port of the reference algorithm), 1 thread,
            # on the first instances of the same parameter table
            ncores, quota = cpu_share()
            nthr = ncores if args.cpu_threads <= 0 else max(1, min(len(os.sched_getaffinity(0)), args.cpu_threads))
            ph = params[:, :min(B, 1024)].cpu().numpy()
            s_cpu = min(S * args.steps, 2000)                  # bounded sample: at most 2000 time steps per instance
            est_per_inst = 10.0 * s_cpu
            n_cpu = int(max(1, min(ph.shape[1], round(args.cpu_iters / est_per_inst))))
            ci, cdt = cpu_baseline(nl, ph, n_cpu, tstep, s_cpu)
            rec["cpu_baseline"] = {
                "value": ci / cdt,
                "unit": "NR-iter*inst/s",
                "cores": 1,
                "kind": "port",
                "cpu_model": cpu_model(),
                "sample": "oracle/mna_oracle.c, instances 0..%d of the same table, %d time steps from the DC "
                          "point (%d NR iterations, %.1f s, host has %d cores, affinity %d, cgroup CPU quota %s)"
                          % (n_cpu - 1, s_cpu, ci, cdt, os.cpu_count() or 0, len(os.sched_getaffinity(0)),
                             ("%.1f cores" % quota) if quota is not None else "none"),
            }
            # the same port on ALL host cores this process may run on (one thread per core over
            # instances; the C call releases the GIL), a second bounded sample sized for ~cpu_seconds of wall time
            if nthr > 1:
                # thread counts tried: every CPU this process may use, and -- when that is more than 16 -- the 16 a one-GPU
                # job is given on this pool (a share the cgroup files do not always show: 256 threads on a 16-core share
                # measured 1.8e6, 16 threads 3.9e6); the better one is reported, both are kept
                per_inst_s = cdt / n_cpu
                tried = []
                for thr in sorted({nthr, min(nthr, 16)}):
                    n_mt = int(max(thr, min(ph.shape[1], 1024, round(args.cpu_seconds * thr / per_inst_s))))
                    mi, mdt = cpu_baseline(nl, ph, n_mt, tstep, s_cpu, threads=thr)
                    tried.append({"value": mi / mdt, "cores": thr,
                                  "sample": "instances 0..%d, %d time steps (%d NR iterations, %.1f s, %d threads)"
                                            % (n_mt - 1, s_cpu, mi, mdt, thr)})
                best = max(tried, key=lambda t: t["value"])
                rec["cpu_baseline"]["all_cores"] = dict(best, host_cores=os.cpu_count() or 0, usable_cores=nthr,
                                                        cgroup_cpu_quota_cores=quota, tried=tried)
        print(json.dumps(rec))

    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
