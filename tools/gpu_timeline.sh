#!/bin/bash
# per-dispatch timeline (start offset, duration, gap to the previous kernel) of a few steps of the metric config
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in ${MODES:-lazy nolazy}; do
  E="X=1"; [[ $m == nolazy* ]] && E="MBD_NO_LAZY=1"; [[ $m == aux* ]] && E="MBD_NO_FUSED_NOISE=1"
  [[ $m == *-noev ]] && E="$E MBD_BENCH_EVENTS=none"; [[ $m == *-ev ]] && E="$E MBD_BENCH_EVENTS=all"
  [[ $m == nopf* ]] && E="$E MBD_NO_PREFETCH=1"
  [[ $m == nolds* ]] && E="$E MBD_LDS_RESERVE=0"
  env $E rocprofv3 --kernel-trace -d $OUT/tl_$m -o t -- python $R/bench.py --config ${CFG:-metric} --no-cpu-baseline --no-final-reward --steps 30 --warmup 5 > $OUT/tl_$m.log 2>&1
  python - <<P
import sqlite3,glob
db=glob.glob("$OUT/tl_$m/*.db")[0]
c=sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
v=[t for t in tabs if t.startswith('kernels')] or [t for t in tabs if 'kernel' in t]
rows=list(c.execute("select name, start, end from %s order by start" % v[0]))
print("$m", len(rows), "dispatches")
# async half of the run = the last third; print 14 dispatches from there (WIN=sync: from the first third, the leg with
# the per-step host read)
import os
k0=len(rows)//4 if os.environ.get("WIN")=="sync" else len(rows)-40
prev=None
gaps={}
for i,(n,s,e) in enumerate(rows):
    short=n.split('(')[0].replace('void ','').replace('mbd::','')[:22]
    if prev is not None and (i>=len(rows)-100 if os.environ.get("WIN")!="sync" else (len(rows)//6<=i<len(rows)//3)):
        gaps.setdefault((prevname,short),[]).append((s-prev)/1000.0)
    if i>=k0 and i<k0+14:
        print("  %-22s start %+10.2f us  dur %8.2f  gap %6.2f" % (short,(s-rows[k0][1])/1000.0,(e-s)/1000.0,(s-prev)/1000.0 if prev else 0))
    prev=e; prevname=short
for k,v in gaps.items(): print("  gap %-22s -> %-22s  n=%3d  avg %6.2f us" % (k[0],k[1],len(v),sum(v)/len(v)))
P
  rm -rf $OUT/tl_$m
done 2>&1 | tee $OUT/timeline.log
