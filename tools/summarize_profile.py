#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_bench.sh on the GPU box) into the
committed artefacts profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.csv,
profiles/<tag>_summary.md and profiles/kernel_counters.json (read by bench.py for roofline.traffic
and roofline_valu).

FETCH_SIZE / WRITE_SIZE come from separate --pmc passes and are in KiB (x1024).  The gfx950
correction of MI355X_MICROARCH.md (FETCH_SIZE reads 1/2 for wide 16 B/lane streams) is NOT applied:
these kernels load 8 B/lane, and the raw figure reproduces the known compulsory bytes of the launch
(params + state) to within 3 %, which is the calibration that guide asks for.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    bench = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
    cmd = open(os.path.join(src, "command.txt")).read().strip()
    # calls of csim_tran_batch_dev in the traced run: timed + warm-up steps + the untimed set-up probe of
    # bench.py (one more call of the same length, when it looked for flagged instances)
    n_steps = bench["steps"] + bench["warmup"] + int(bench["config"].get("setup_probe_steps", 0))

    # kernel-trace stats (only this repo's kernels + the total)
    # (gpurun merges a call's files into gpurun_out/: an earlier profile of the same tag leaves its files next to the
    # new ones -- always take the newest)
    def newest(pattern):
        fs = sorted(glob.glob(pattern), key=os.path.getmtime)
        return fs[-1:] if fs else []

    stats = list(csv.DictReader(open(newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0])))
    ours = [r for r in stats if "csim" in r["Name"]]
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in ours:
            w.writerow([r["Name"].split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])

    # PMC passes: mean per dispatch per kernel
    pmc = collections.defaultdict(dict)
    meta = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        fs = newest(os.path.join(d, "*", "*_counter_collection.csv"))
        if not fs:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if "csim" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0]
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["Scratch_Size"], r["VGPR_Count"],
                       r["Accum_VGPR_Count"], r["SGPR_Count"])
        # one bench step = one call of csim_tran_batch_dev = several launches (the scheduled kernel, then
        # hand-back rounds that are idle unless an instance left its recorded schedules): report the
        # SUM over a step's launches, i.e. total / number of bench steps, not a per-dispatch mean
        for (k, c), v in acc.items():
            per = n_steps if "tran" in k else 1
            pmc[k][c] = sum(v) / max(per, 1)
    counters = sorted({c for k in pmc for c in pmc[k]})
    with open(os.path.join(dst, tag + "_pmc.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid", "wg", "lds_bytes", "scratch_bytes", "vgpr", "agpr", "sgpr"] + counters)
        for k in sorted(pmc):
            w.writerow([k] + list(meta[k]) + [pmc[k].get(c, "") for c in counters])

    # the dominant kernel of the timed region
    # (a JIT run also executes the planner's general-kernel launches, which can outweigh the timed ones:
    # when the bench ran a generated kernel, that is the kernel the numbers are about)
    tran = [r for r in ours if "tran" in r["Name"]]
    if bench["config"]["kernel"] == "scheduled" and any(r["Name"].startswith("csim_tran_") for r in tran):
        tran = [r for r in tran if r["Name"].startswith("csim_tran_")]
    dom = max(tran, key=lambda r: float(r["TotalDurationNs"]))
    dname = dom["Name"].split("(")[0]
    p = pmc.get(dname, {})
    fetch = p.get("FETCH_SIZE", 0.0) * 1024
    write = p.get("WRITE_SIZE", 0.0) * 1024
    cfg = bench["config"]
    # per NR iteration x instance, so that bench.py can scale to any launch length
    units = float(cfg["nr_iters_per_step"])
    lanes = cfg.get("lanes_per_instance", 1)
    kname = cfg["kernel"] if lanes in (1, 64) else "%s%d" % (cfg["kernel"], lanes)
    import re
    n_unknowns = int(re.search(r"N=(\d+)", cfg["workload"]).group(1))
    key = "N%d|%s|B%d" % (n_unknowns, kname, cfg["batch_per_gpu"])
    tpath = os.path.join(dst, "kernel_counters.json")
    table = json.load(open(tpath)) if os.path.exists(tpath) else {}
    m = meta.get(dname, ("",) * 7)
    table[key] = {"hbm_bytes_per_unit": (fetch + write) / units, "fetch_bytes_per_unit": fetch / units,
                  "write_bytes_per_unit": write / units,
                  "valu_wave_insts_per_unit": p.get("SQ_INSTS_VALU", 0.0) / units,
                  "lds_wave_insts_per_unit": p.get("SQ_INSTS_LDS", 0.0) / units,
                  "waves_per_launch": p.get("SQ_WAVES", 0.0), "vgpr": m[4], "agpr": m[5], "scratch_bytes": m[3],
                  "lds_bytes": m[2], "time_steps_per_launch": cfg["time_steps_per_step"],
                  "source": "profiles/%s_pmc.csv (%s)" % (tag, dname)}
    json.dump(table, open(tpath, "w"), indent=1, sort_keys=True)

    wave_cyc = p.get("SQ_WAVE_CYCLES", 0.0)
    lines = [
        "# rocprofv3 summary %s" % tag, "",
        "command: `python bench.py %s` on one MI355X (gfx950)" % cmd.replace("bench args: ", ""), "",
        "bench line of the traced run: value = %.4g %s, kernel = %s, kernel_avg_ms (HIP events) = %.3f, "
        "roofline.frac = %.3f" % (bench["value"], bench["unit"], cfg["kernel"], bench["roofline"]["kernel_avg_ms"],
                                  bench["roofline"]["frac"]), "",
        "## kernel trace (`--kernel-trace --stats`)", "",
        "| kernel | calls | avg ms | share |", "|---|---|---|---|"]
    for r in ours:
        lines.append("| %s | %s | %.3f | %s %% |" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e6,
                                                    r["Percentage"]))
    tran_total_ms = sum(float(r["TotalDurationNs"]) for r in ours if "tran" in r["Name"]) / 1e6
    lines += ["", "One bench step = one `csim_tran_batch_dev` call = the scheduled launch plus hand-back rounds (idle "
              "here). rocprofv3: all transient launches of a step sum to %.3f ms (%.3f ms of it `%s`); bench.py's "
              "HIP-event time of the same step: %.3f ms." % (tran_total_ms / n_steps, float(dom["TotalDurationNs"]) / 1e6 / n_steps,
                                                            dname, bench["roofline"]["kernel_avg_ms"]), "",
              "## PMC (separate `--pmc` passes; transient kernels: sum over the launches of one bench step)", ""]
    for k in sorted(pmc):
        lines.append("**%s** grid=%s wg=%s LDS=%s B scratch=%s B VGPR=%s AGPR=%s" % ((k,) + meta[k][:6]))
        for c in counters:
            if c in pmc[k]:
                lines.append("- %s = %.6g" % (c, pmc[k][c]))
        lines.append("")
    if wave_cyc:
        lines += ["## reading (dominant kernel `%s`)" % dname, "",
                  "- HBM traffic per launch: FETCH_SIZE %.2f MB + WRITE_SIZE %.2f MB (compulsory: parameters "
                  "P*8*B + state N*8*B in, state out)." % (fetch / 1e6, write / 1e6),
                  "- LDS bank conflicts: %.3g of %.3g LDS-active cycles (%.2f %%)."
                  % (p.get("SQ_LDS_BANK_CONFLICT", 0), p.get("SQ_LDS_IDX_ACTIVE", 0),
                     100.0 * p.get("SQ_LDS_BANK_CONFLICT", 0) / max(p.get("SQ_LDS_IDX_ACTIVE", 1), 1)),
                  "- wave cycles: VALU active %.1f %%, waiting (s_waitcnt) %.1f %%, issue stalls %.1f %% of "
                  "SQ_WAVE_CYCLES." % (100 * p.get("SQ_ACTIVE_INST_VALU", 0) / wave_cyc,
                                       100 * p.get("SQ_WAIT_ANY", 0) / wave_cyc,
                                       100 * p.get("SQ_WAIT_INST_ANY", 0) / wave_cyc),
                  "- L2: %.3g hits, %.3g misses per launch." % (p.get("TCC_HIT_sum", 0), p.get("TCC_MISS_sum", 0))]
    open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-8:]))


if __name__ == "__main__":
    main()
