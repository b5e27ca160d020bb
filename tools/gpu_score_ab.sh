#!/bin/bash
# the score + weighted-mean launch, A/B over its layout levers (same bits whatever they say): kernel time (rocprofv3 --stats) and
# HBM bytes per launch (separate --pmc passes, FETCH_SIZE doubled for gfx950) of
#   the single-plan launch pinned to X XCDs (MBD_WMEAN_XCDS; unset: the library's choice), configs metric hopper512 halfcheetah1024 humanoidrun4096
#   the sweeps' batch launch with V outputs per thread (MBD_WMEAN_V = 1 2 4), config sweep8
# usage (GPU box): tools/gpu_score_ab.sh > gpurun_out/score_ab.txt
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; W=/tmp/score_ab; mkdir -p $W; cd /tmp; export TMPDIR=/tmp
one() {  # config, lever assignment ("" or VAR=val), label
  local c=$1 lv=$2 lab=$3
  local B="python $R/bench.py --config $c --no-cpu-baseline --no-final-reward --no-extras --repeats 2 --steps 30 --warmup 5"
  rm -rf $W/s $W/f $W/w
  env $lv rocprofv3 --kernel-trace --stats -d $W/s -o x -- $B > $W/s.log 2>&1
  env $lv rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $W/f -o x -- $B > $W/f.log 2>&1
  env $lv rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $W/w -o x -- $B > $W/w.log 2>&1
  python - "$c" "$lab" $W <<'PY'
import sqlite3, sys
c, lab, W = sys.argv[1:4]
def q(db, sql):
    return list(sqlite3.connect(f"{W}/{db}/x_results.db").execute(sql))
t = {n: a for n, a in q("s", "select name, average from top_kernels") if "score_wmean" in n or "wmean" in n}
f = {n: a for n, a in q("f", "select kernel_name, avg(value) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name")}
w = {n: a for n, a in q("w", "select kernel_name, avg(value) from counters_collection where counter_name='WRITE_SIZE' group by kernel_name")}
for n, us in t.items():
    b = (2.0 * f.get(n, 0.0) + w.get(n, 0.0)) * 1024.0
    print(f"{c:18s} {lab:22s} {n[:44]:44s} {us:8.2f} us  {b / 1e6:8.3f} MB per launch")
PY
}
for c in metric hopper512 halfcheetah1024 humanoidrun4096; do
  one $c "" "library"
  for x in 1 2 8; do one $c "MBD_WMEAN_XCDS=$x" "MBD_WMEAN_XCDS=$x"; done
done
for v in 1 2 4; do one sweep8 "MBD_WMEAN_V=$v" "MBD_WMEAN_V=$v"; done
