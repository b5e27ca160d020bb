#!/bin/bash
# round-2 iteration pass: parity tests, the layout probe, short bench lines of the configs given as arguments
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
python __graft_entry__.py 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest_r02b.log
[ -x tools/probes/probe_quad ] && ./tools/probes/probe_quad | tee $OUT/probe_quad.log
for c in "$@"; do
  python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --no-final-reward 2>>$OUT/bench_err.log | tail -1 > $OUT/bench_r02b_$c.json
  python - <<PY
import json; d=json.load(open("$OUT/bench_r02b_$c.json")); print("$c", "value", round(d["value"],1), "async", round(d["value_async"],1), "ms", round(d["ms_per_step"],4), "kernel_ms", round(d["roofline"]["kernel_avg_ms"],4))
PY
done
