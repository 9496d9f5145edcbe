#!/bin/bash
# dev helper (GPU box): what the driver runs at round end -- GPU tests, smoke, default bench
OUT=gpurun_out/${1:-validate}
mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -4 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
r = json.load(open("$OUT/bench.json"))
print("value %.4g  ms/step %.2f  lanes %s  roofline.frac %.3f  valu_frac %s  large %.4g (lanes %s)  cpu %.3g" % (
    r["value"], r["ms_per_step"], r["config"]["lanes_per_instance"], r["roofline"]["frac"],
    r["roofline_valu"]["achieved_frac"], r["large_batch"]["value"], r["large_batch"]["lanes_per_instance"], r["cpu_baseline"]["value"]))
PY
