#!/bin/bash
# outputs per thread of the single-plan score + weighted-mean launch (MBD_WMEAN_V1 = 1 / 2; round 6), A/B: kernel time
# (rocprofv3 --stats) and HBM bytes per launch (separate --pmc passes, FETCH_SIZE doubled for gfx950), steps/s.
# usage (GPU box): tools/gpu_score_v1_ab.sh > gpurun_out/score_v1_ab.txt
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; W=/tmp/score_v1; mkdir -p $W; cd /tmp; export TMPDIR=/tmp
for c in humanoidrun4096 humanoidrun8192 humanoidtrack2048demo; do
  for v in 1 2; do
    B="python $R/bench.py --config $c --no-cpu-baseline --no-final-reward --no-extras --repeats 2 --steps 30 --warmup 5"
    rm -rf $W/s $W/f $W/w
    MBD_WMEAN_V1=$v python $R/bench.py --config $c --no-cpu-baseline --no-final-reward --no-extras 2>/dev/null | tail -1 > $W/line.json
    MBD_WMEAN_V1=$v rocprofv3 --kernel-trace --stats -d $W/s -o x -- $B > $W/s.log 2>&1
    MBD_WMEAN_V1=$v rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $W/f -o x -- $B > $W/f.log 2>&1
    MBD_WMEAN_V1=$v rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $W/w -o x -- $B > $W/w.log 2>&1
    python - "$c" "$v" $W <<'PY'
import json, sqlite3, sys
c, v, W = sys.argv[1:4]
d = json.loads(open(f"{W}/line.json").read())
def q(db, sql):
    return list(sqlite3.connect(f"{W}/{db}/x_results.db").execute(sql))
t = {n: a for n, a in q("s", "select name, average from top_kernels") if "score_wmean" in n}
f = {n: a for n, a in q("f", "select kernel_name, avg(value) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name")}
w = {n: a for n, a in q("w", "select kernel_name, avg(value) from counters_collection where counter_name='WRITE_SIZE' group by kernel_name")}
for n, us in t.items():
    b = (2.0 * f.get(n, 0.0) + w.get(n, 0.0)) * 1024.0
    print(f"{c:22s} MBD_WMEAN_V1={v}  {d['value']:8.1f} steps/s  {n[:40]:40s} {us:8.2f} us  {b / 1e6:8.3f} MB per launch")
PY
  done
done
