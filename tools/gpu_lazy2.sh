#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/lazy_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/lazy_tests.log
tail -3 $OUT/lazy_tests.log
run() { env $3 python bench.py --config $1 --steps 198 --warmup 20 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-22s %-8s' % ('$1', '$2'), 'steps/s %.1f  async %.1f  ms/step %.4f  rollout_kernel_ms %.4f' % (d['value'], d['value_async'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))"; }
{
for round in 1 2; do
  run metric lazy "X=1"
  run metric nolazy "MBD_NO_LAZY=1"
done
for c in hopper512 halfcheetah1024; do
  for round in 1 2; do run $c lazy "X=1"; run $c nolazy "MBD_NO_LAZY=1"; done
done
} | tee $OUT/lazy_ab.log
cd /tmp && export TMPDIR=/tmp
for m in lazy nolazy; do
  E="X=1"; [ $m = nolazy ] && E="MBD_NO_LAZY=1"
  env $E rocprofv3 --kernel-trace --stats -d $OUT/prof_$m -o t -- python $R/bench.py --config metric --no-cpu-baseline --no-final-reward --steps 60 --warmup 10 > $OUT/prof_$m.log 2>&1
  python - <<P
import sqlite3,glob
db=glob.glob("$OUT/prof_$m/*.db")[0]
c=sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
v=[t for t in tabs if t.startswith('kernels')] or [t for t in tabs if 'kernel' in t]
q="select name, count(*), avg(end-start)/1000.0 from %s group by name order by 3 desc" % v[0]
print("$m")
for r in c.execute(q): print("  %-80s %5d %9.2f us" % (r[0][:80], r[1], r[2]))
P
  rm -rf $OUT/prof_$m
done 2>&1 | tee $OUT/lazy_prof.log
