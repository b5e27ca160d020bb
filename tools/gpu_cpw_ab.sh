#!/bin/bash
# Candidates per wavefront + contact early-out of the planar rollouts (RolloutParams::cpw, mbd_planar.h EO), A/B: steps/s and the
# rollout kernel's time (bench.py) under MBD_CPW = 0 (filled wavefronts) / 1 / 2 / 4 / unset (the library's choice).
# usage (GPU box): tools/gpu_cpw_ab.sh [configs...]
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
CFGS=${@:-hopper512 halfcheetah1024}
for c in $CFGS; do
  for cpw in 0 1 2 4 -1; do
    MBD_CPW=$cpw python $R/bench.py --config $c --no-cpu-baseline --no-final-reward --no-extras 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print(f\"$c MBD_CPW=$cpw  {d['value']:8.1f} steps/s [{d['value_min']:.1f}, {d['value_max']:.1f}]  kernel {d['roofline']['kernel_avg_ms'] * 1e3:7.2f} us\")
"
  done
done
