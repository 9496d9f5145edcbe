#!/usr/bin/env python3
"""Record the transient (and DC) pivot schedules of a netlist with the general kernel (needs a GPU).

    python tools/record_schedule.py tests/golden/dbmixer.sp [--steps 2000] [--tstep 1e-13] [--mc 16]

Prints every distinct "column:row,..." sequence with how many factorisations used it, for the nominal
circuit and a few Monte-Carlo instances, and the body of circuitsimulator_amd/csrc/schedules/<name>.sched
(one alternative per line, most frequent first).  The schedule is verified again at run time on every
factorisation, so this only decides how often the fast kernel is used, never what it computes.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("netlist")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--tstep", type=float, default=None)
    ap.add_argument("--mc", type=int, default=16)
    a = ap.parse_args()
    from circuitsimulator_amd import Engine, Netlist
    nl = Netlist.from_file(a.netlist)
    eng = Engine(nl, 0)
    params = eng.mc_params(12345, 0.05, 0, max(1, a.mc))
    total, total_dc = {}, {}
    for b in range(max(1, a.mc)):
        if nl.has_nonlinear and nl.n_unknowns <= 63:
            dcs, dother = eng.record_dc_pivot_schedules(params, b)
            print("instance %3d DC: %s%s" % (b, "  ".join("[%s] x%d" % sc for sc in dcs),
                                             ("  other/failed x%d" % dother) if dother else ""))
            for sched, n in dcs:
                total_dc[sched] = total_dc.get(sched, 0) + n
        alts, other = eng.record_pivot_schedules(params, b, a.tstep, a.steps)
        print("instance %3d: %s%s" % (b, "  ".join("[%s] x%d" % sc for sc in alts),
                                      ("  other/failed x%d" % other) if other else ""))
        for sched, n in alts:
            total[sched] = total.get(sched, 0) + n
    print("\n# schedule file body (most frequent first; one alternative per line):")
    for sched, n in sorted(total.items(), key=lambda kv: -kv[1]):
        print("%s    # %d factorisations" % (sched, n))
    if total_dc:
        print("# DC operating point (worth listing when a handful of sequences covers every instance):")
        for sched, n in sorted(total_dc.items(), key=lambda kv: -kv[1]):
            print("dc %s    # %d factorisations" % (sched, n))


if __name__ == "__main__":
    main()
