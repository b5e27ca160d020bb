#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'steps/s %.1f  ms/step %.3f  rollout_kernel_ms %.3f' % (d['steps_per_sec'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))"; }
rocm-smi --showperflevel --showclocks 2>&1 | grep -E "sclk|mclk|Performance Level|fclk" | head -8
run default
( while true; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -1; sleep 0.2; done ) > gpurun_out/clk.log 2>&1 &
MON=$!
run default2
kill $MON
sort gpurun_out/clk.log | uniq -c | sort -rn | head -5
rocm-smi --setperflevel high 2>&1 | tail -2
run perflevel_high
rocm-smi --setperflevel auto 2>&1 | tail -1
run auto_again
