#!/bin/bash
# XCD pinning of small rollout launches (RolloutParams::xcd_pin), A/B: steps/s and kernel time (bench.py) and the rollout kernel's HBM
# bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes) with MBD_ROLL_PIN = 0 / 1.  usage (GPU box): tools/gpu_rollpin_ab.sh
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; W=/tmp/rollpin; mkdir -p $W; cd /tmp; export TMPDIR=/tmp
for c in hopper512 halfcheetah1024 metric; do
  for pin in 0 1; do
    B="python $R/bench.py --config $c --no-cpu-baseline --no-final-reward --no-extras"
    MBD_ROLL_PIN=$pin $B 2>/dev/null | tail -1 > $W/line.json
    rm -rf $W/f $W/w
    MBD_ROLL_PIN=$pin rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $W/f -o x -- $B --repeats 2 --steps 30 --warmup 5 > $W/f.log 2>&1
    MBD_ROLL_PIN=$pin rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $W/w -o x -- $B --repeats 2 --steps 30 --warmup 5 > $W/w.log 2>&1
    python - $c $pin $W <<'PY'
import json, sqlite3, sys
c, pin, W = sys.argv[1:4]
d = json.loads(open(f"{W}/line.json").read())
def q(db, name):
    return {n: a for n, a in sqlite3.connect(f"{W}/{db}/x_results.db").execute(
        f"select kernel_name, avg(value) from counters_collection where counter_name='{name}' group by kernel_name")}
f, w = q("f", "FETCH_SIZE"), q("w", "WRITE_SIZE")
for n in f:
    if "rollout" in n:
        print(f"{c:16s} MBD_ROLL_PIN={pin}  {d['value']:8.1f} steps/s [{d['value_min']:.1f}, {d['value_max']:.1f}]  kernel {d['roofline']['kernel_avg_ms'] * 1e3:7.2f} us  "
              f"{(2 * f[n] + w.get(n, 0.0)) * 1024 / 1e6:6.3f} MB per launch (B_alg {d['roofline']['algorithmic_bytes_per_launch'] / 1e6:.2f})")
PY
  done
done
