#!/bin/bash
# EXECUTED instructions per wavefront and substep of the planar rollouts under MBD_CPW (PMC: SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAVES,
# GRBM_GUI_ACTIVE in its own pass) — the early-out makes the executed count differ from the static one (tools/count_flops.py).
# usage (GPU box): tools/gpu_cpw_pmc.sh [config:substeps_per_rollout ...]
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; W=/tmp/cpwpmc; mkdir -p $W; cd /tmp; export TMPDIR=/tmp
CFGS=${@:-hopper512:1000 halfcheetah1024:800}
for cs in $CFGS; do
  c=${cs%%:*}; sub=${cs##*:}
  for cpw in 0 -1; do
    B="python $R/bench.py --config $c --no-cpu-baseline --no-final-reward --no-extras --repeats 2 --steps 20 --warmup 2"
    rm -rf $W/a $W/b
    MBD_CPW=$cpw rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --kernel-trace -d $W/a -o x -- $B > $W/a.log 2>&1
    MBD_CPW=$cpw rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $W/b -o x -- $B > $W/b.log 2>&1
    python - $c $cpw $W $sub <<'PY'
import sqlite3, sys
c, cpw, W, sub = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
def q(db):
    con = sqlite3.connect(f"{W}/{db}/x_results.db")
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='view' or type='table'")]
    t = "counters_collection" if "counters_collection" in tabs else [x for x in tabs if "counters_collection" in x][0]
    out = {}
    for k, n, v in con.execute(f"select kernel_name, counter_name, avg(value) from {t} group by kernel_name, counter_name"):
        if "rollout" in k:
            out[n] = v
    return out
a, b = q("a"), q("b")
w = a["SQ_WAVES"]
print(f"{c:16s} MBD_CPW={cpw:>2s}  waves {w:6.0f}  VALU/wave/substep {a['SQ_INSTS_VALU'] / w / sub:7.1f}  SALU/wave/substep {a['SQ_INSTS_SALU'] / w / sub:6.1f}  "
      f"VALU-active share of wave cycles {a['SQ_ACTIVE_INST_VALU'] / a['SQ_WAVE_CYCLES']:.3f}  GRBM_GUI_ACTIVE/8 {b['GRBM_GUI_ACTIVE'] / 8:9.0f} clk")
PY
  done
done
