#!/bin/bash
# steps/s without the per-launch timing events, no profiler attached
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { env $2 python bench.py --config ${CFG:-metric} --steps 198 --warmup 20 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-28s' % ('$1'), 'steps/s %.1f  async %.1f  ms/step %.4f async %.4f' % (d['value'], d['value_async'], d['ms_per_step'], d['ms_per_step_async']))"; }
{
for r in 1 2; do
run lazy-ev "MBD_BENCH_EVENTS=all"
run lazy-noev "MBD_BENCH_EVENTS=none"
run nolazy-ev "MBD_NO_LAZY=1 MBD_BENCH_EVENTS=all"
run nolazy-noev "MBD_NO_LAZY=1 MBD_BENCH_EVENTS=none"
run nopf-noev "MBD_NO_PREFETCH=1 MBD_BENCH_EVENTS=none"
done
} | tee gpurun_out/noev.log
