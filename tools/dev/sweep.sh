#!/bin/bash
# dev helper (GPU box): bench.py over lanes-per-instance and batch sizes; one line per run
OUT=gpurun_out/${1:-sweep}
mkdir -p $OUT
shift
run() {   # lanes batch steps warmup
  python bench.py --steps $3 --warmup $4 --lanes $1 --no-cpu --large-batch 0 --batch $2 > $OUT/bench_l$1_b$2.json 2> $OUT/bench_l$1_b$2.err
  python - <<PY
import json
try:
    r = json.load(open("$OUT/bench_l$1_b$2.json"))
    print("lanes %2d B %6d  %.4g NR-iter*inst/s  %.2f ms/step  kernel_avg %.2f ms  flagged %d" % ($1, $2, r["value"], r["ms_per_step"], r["roofline"]["kernel_avg_ms"], r["config"]["flagged_instances"]))
except Exception as e:
    print("lanes $1 B $2 FAILED", e, open("$OUT/bench_l$1_b$2.err").read()[-800:])
PY
}
for spec in "$@"; do
  IFS=: read L B S W <<< "$spec"
  run $L $B ${S:-6} ${W:-2}
done
