#!/usr/bin/env python3
"""GPU-free fuzz of the kernel generator: tests/test_generated_host.py's random-circuit check over a range of seeds
(oracle LU records the pivot sequences, csim_codegen generates, g++ compiles the faithful DC and transient kernels for the
host, results must equal the oracle bit for bit).

    python tools/fuzz_host.py [--first 100] [--count 100]
"""
import argparse
import os
import sys
import tempfile
import pathlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=100)
    ap.add_argument("--count", type=int, default=100)
    ap.add_argument("--fast", action="store_true", help="check the FAST lane-per-instance kernel (counts equal, states within 1e-9) instead of the faithful ones")
    ap.add_argument("--stress", action="store_true", help="add gate-only and capacitor-only nodes to every circuit")
    ap.add_argument("--tstep-scale", type=float, default=1.0, help="multiply the circuits' time step (hard switching: several pivot sequences per run)")
    a = ap.parse_args()
    os.environ["CSIM_FUZZ_TSTEP_SCALE"] = repr(a.tstep_scale)
    if a.stress:
        os.environ["CSIM_FUZZ_STRESS"] = "1"
    import pytest
    import test_generated_host as T
    codegen = os.path.join(T.CSRC, "build", "csim_codegen")
    fn = T.test_random_circuit_generated_fast_kernel_on_the_host if a.fast else T.test_random_circuit_generated_faithful_kernels_on_the_host
    ok = skipped = bad = 0
    for seed in range(a.first, a.first + a.count):
        with tempfile.TemporaryDirectory() as d:
            try:
                fn(codegen, pathlib.Path(d), seed)
                ok += 1
            except pytest.skip.Exception:
                skipped += 1
            except AssertionError as e:
                bad += 1
                print("seed %d: %s" % (seed, str(e)[:300]), flush=True)
        if (seed - a.first) % 20 == 19:
            print("  ... %d seeds: %d ok, %d linear (skipped), %d FAILED" % (seed - a.first + 1, ok, skipped, bad), flush=True)
    print("fuzz (host, %s): %d circuits %s, %d linear circuits skipped, %d failed" % (
        "fast kernel" if a.fast else "faithful kernels", ok, "with equal NR counts and states within 1e-9" if a.fast else "bit for bit the oracle", skipped, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
