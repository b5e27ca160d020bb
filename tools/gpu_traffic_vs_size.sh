#!/bin/bash
# HBM bytes per rollout launch against the batch size (PMC: 2 x FETCH_SIZE + WRITE_SIZE, separate passes), to tell what a
# small launch reads beyond its algorithmic bytes: a part per WAVEFRONT (constants gathered per lane) grows with N; a part
# per XCD (the kernel's code and the model, once per L2) does not.  usage (GPU box): tools/gpu_traffic_vs_size.sh ENV N...
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
ENV=${1:-hopper}; shift; NS=${*:-"64 512 2048 8192"}
cd /tmp && export TMPDIR=/tmp
for N in $NS; do
  for pmc in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $pmc --kernel-trace -d $OUT/tv_${N}_$pmc -o tv -- python $R/tools/gpu_planrun.py $ENV $N 12 > $OUT/tv_${N}_$pmc.log 2>&1
  done
  python - "$ENV" "$N" $OUT/tv_${N}_FETCH_SIZE/tv_results.db $OUT/tv_${N}_WRITE_SIZE/tv_results.db <<'PY'
import sqlite3, sys
env, N = sys.argv[1], int(sys.argv[2])
v = {}
for db in sys.argv[3:]:
    for k, c, n, a in sqlite3.connect(db).execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if "rollout" in k and "<" in k:
            v.setdefault(k[:60], {})[c] = (n, a)
for k, d in v.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d and d["FETCH_SIZE"][0] > 20:
        f, w = d["FETCH_SIZE"][1], d["WRITE_SIZE"][1]
        print(f"{env} N={N} {k}: launches {d['FETCH_SIZE'][0]}  fetch {2 * f:.0f} KiB (raw {f:.0f})  write {w:.0f} KiB  total {(2 * f + w) / 1024:.3f} MiB")
PY
  rm -rf $OUT/tv_${N}_FETCH_SIZE $OUT/tv_${N}_WRITE_SIZE
done
