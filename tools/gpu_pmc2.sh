#!/bin/bash
# second-level PMC passes for the rollout kernel: where the non-VALU quarter of the wave's cycles goes
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-final-reward"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --kernel-trace -d $OUT/pmc2_a -o r -- $B > $OUT/pmc2_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU --kernel-trace -d $OUT/pmc2_b -o r -- $B > $OUT/pmc2_b.log 2>&1
ls $OUT/pmc2_a $OUT/pmc2_b
