#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of the bench command.
# Outputs under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into profiles/<tag>_*.
# PMC passes never combine with trace domains other than --kernel-trace (pool rule).
set -u
TAG=${1:-r01}
shift || true
ARGS=${*:---steps 3 --warmup 1 --tsteps 100 --no-cpu --large-batch 0}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
echo "bench args: $ARGS" > "$OUT/command.txt"
# one un-profiled run first: it fills the JIT cache (a netlist without a shipped kernel library is specialised with
# hipcc on first use), so that no compiler is started from under the profiler (the engine also strips the profiler's
# preload from the compiler's environment, jit.cpp)
timeout -k 10 600 python3 bench.py $ARGS > "$OUT/bench_warm.json" 2> "$OUT/warm.err" || echo "warm-up run failed"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || echo "trace pass failed"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_WAVES TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '+')
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py $ARGS > "$OUT/bench_$name.json" 2> "$OUT/pmc_$name.err" || echo "pmc pass $name failed"
done
ls -R "$OUT" | head -60
