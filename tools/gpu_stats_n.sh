#!/bin/bash
# per-kernel durations of the step at other candidate counts (what every rank of a G-GPU run executes for the
# sampling / scoring / weighted-mean part, which is over N_total)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for N in 2048 4096 8192; do
  MBD_BENCH_N=$N rocprofv3 --kernel-trace --stats -d $OUT/stats_n$N -o r -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-final-reward > $OUT/stats_n$N.log 2>&1
done
python - <<'PY'
import sqlite3, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
for N in (2048, 4096, 8192):
    con = sqlite3.connect(f"{out}/stats_n{N}/r_results.db")
    print(N, [(r[0][:24].replace("void ", ""), round(r[2], 1)) for r in con.execute("select name,total_calls,average from top_kernels") if "mbd::" in r[0]])
PY
