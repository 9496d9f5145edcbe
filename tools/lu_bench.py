#!/usr/bin/env python3
"""Stand-alone batched LU (csim_lu_solve_batch = Solver::solveLinearSystemLU per system): run under
`rocprofv3 --kernel-trace --stats` for the kernels' own durations (the entry point takes host tables, so wall time is
dominated by the PCIe copies); prints the work per call so that the trace can be turned into rates.

    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/lu_bench.py
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    from circuitsimulator_amd import lu_solve_batch
    rs = np.random.RandomState(7)
    out = []
    for n, B in ((13, 65536), (31, 65536), (48, 32768), (63, 16384), (128, 2048), (256, 1024), (512, 256), (1024, 64)):
        A = rs.uniform(-1.0, 1.0, size=(B, n, n))
        A[:, np.arange(n), np.arange(n)] += 0.25 * n          # well conditioned, pivoting still happens
        b = rs.uniform(-1.0, 1.0, size=(B, n))
        lu_solve_batch(A[:8], b[:8])                          # warm-up
        t0 = time.perf_counter()
        x, flags = lu_solve_batch(A, b)
        dt = time.perf_counter() - t0
        r = np.abs(np.einsum("bij,bj->bi", A[:64], x[:64]) - b[:64]).max()
        out.append({"n": n, "batch": B, "wall_s": dt, "flops": B * (2.0 / 3.0 * n ** 3 + 2.0 * n * n), "bytes": B * 8.0 * (n * n + 2 * n),
                    "flagged": int((flags != 0).sum()), "residual": float(r)})
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
