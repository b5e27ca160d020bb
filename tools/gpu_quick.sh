#!/bin/bash
# quick on-box check of bench.py's lines (configs of the round) — usage: tools/gpu_quick.sh CONFIG...
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for c in "$@"; do
  timeout 600 python bench.py --config $c --steps 40 --warmup 5 2>gpurun_out/bench_err_$c.log | tail -1 > gpurun_out/bench_quick_$c.json
  python - "$c" <<'PY'
import json,sys
c=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/bench_quick_{c}.json"))
    print(c, "value %.1f"%d["value"], d["unit"][:40], "| ms %.3f"%d["ms_per_step"], "| kern %.3f ms"%d["roofline"]["kernel_avg_ms"], "| valu alg", d["valu"] and d["valu"].get("algorithmic_frac"), "| cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("kind"), "| final", d["final_reward"] and (d["final_reward"]["mean"], d["final_reward"]["std"]))
except Exception as e:
    print(c, "FAILED", e); print(open(f"gpurun_out/bench_err_{c}.log").read()[-2000:])
PY
done
