#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace -d $OUT/pmc_a -o r -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-final-reward > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS --kernel-trace -d $OUT/pmc_b -o r -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-final-reward > $OUT/pmc_b.log 2>&1
tail -3 $OUT/pmc_b.log
