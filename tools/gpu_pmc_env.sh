#!/bin/bash
# PMC split of the rollout kernel of another env (MBD_AB_ENV, default hopper), reference default arguments
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
E=${MBD_AB_ENV:-hopper}
cat > /tmp/run_env.py <<PY
import sys
sys.path.insert(0, "$R/model-based-diffusion_amd")
from mbd_hip.planners.mbd_planner import Args, run_diffusion
a = Args(seed=0, env_name="$E", not_render=True, Ndiffuse=12)
run_diffusion(a)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d $OUT/pmc_env -o r -- python /tmp/run_env.py > $OUT/pmc_env.log 2>&1
python - <<PY
import sqlite3
c = sqlite3.connect("$OUT/pmc_env/r_results.db")
d = {}
for k, n, cnt, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
    if "rollout" in k: d[n] = avg
w = d["SQ_WAVES"]
print({k: round(v / w, 1) for k, v in d.items()})
print("VALU active %.1f%%, wait %.1f%%, cycles/VALU instr %.2f" % (100 * d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 4 * d["SQ_ACTIVE_INST_VALU"] / d["SQ_INSTS_VALU"]))
PY
