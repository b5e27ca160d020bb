#!/bin/bash
# a round's measurement pass: tests, bench line per single-GPU config, rocprofv3 kernel stats + PMC passes per config
# usage (on the GPU box): TAG=r04 tools/gpu_round.sh [tests] [bench] [prof CONFIG...]   (summaries land in gpurun_out/profiles)
TAG=${TAG:-r04}
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
python __graft_entry__.py 2>&1 | tail -1
what="$*"; [ -z "$what" ] && what="tests bench prof metric humanoidrun8192 sweep8"
if [[ " $what " == *" tests "* ]]; then
  timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_${TAG}.log
fi
if [[ " $what " == *" bench "* ]]; then
  python bench.py --steps 60 --warmup 10 2>$OUT/bench_err.log | tail -1 > $OUT/bench_${TAG}_metric.json; cat $OUT/bench_${TAG}_metric.json
  for c in hopper512 halfcheetah1024 humanoidrun4096 humanoidtrack2048demo car2d humanoidrun8192 sweep8; do
    python bench.py --config $c --steps 100 --warmup 10 2>>$OUT/bench_err.log | tail -1 > $OUT/bench_${TAG}_$c.json; cat $OUT/bench_${TAG}_$c.json
  done
fi
if [[ " $what " == *" prof "* ]]; then
  cd /tmp && export TMPDIR=/tmp
  for c in metric hopper512 halfcheetah1024 humanoidrun4096 humanoidtrack2048demo humanoidrun8192 sweep8; do
    [[ " $what " == *" $c "* ]] || continue
    B="python $R/bench.py --config $c --no-cpu-baseline --no-final-reward --no-extras --repeats 2"
    rocprofv3 --kernel-trace --stats -d $OUT/prof_${c}_stats -o ${TAG} -- $B --steps 60 --warmup 10 > $OUT/prof_${c}_stats.log 2>&1
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace -d $OUT/prof_${c}_sq -o ${TAG} -- $B --steps 20 --warmup 2 > $OUT/prof_${c}_sq.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_${c}_fetch -o ${TAG} -- $B --steps 20 --warmup 2 > $OUT/prof_${c}_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace -d $OUT/prof_${c}_write -o ${TAG} -- $B --steps 20 --warmup 2 > $OUT/prof_${c}_write.log 2>&1
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --kernel-trace -d $OUT/prof_${c}_wait -o ${TAG} -- $B --steps 20 --warmup 2 > $OUT/prof_${c}_wait.log 2>&1
    # summarise here: the databases are too big to travel back (gpurun_out is capped at 64 MiB)
    mkdir -p $OUT/profiles
    # (the PMC summary is one file keyed by config: start from the repo's, so that configs not profiled in this call stay)
    [ -f $OUT/profiles/${TAG}_pmc.json ] || cp $R/profiles/${TAG}_pmc.json $OUT/profiles/ 2>/dev/null
    python $R/tools/rocprof_summary.py --out $OUT/profiles ${TAG} --config $c $OUT/prof_${c}_stats/${TAG}_results.db $OUT/prof_${c}_sq/${TAG}_results.db $OUT/prof_${c}_fetch/${TAG}_results.db $OUT/prof_${c}_write/${TAG}_results.db $OUT/prof_${c}_wait/${TAG}_results.db | tail -1 | cut -c1-400
    rm -rf $OUT/prof_${c}_stats $OUT/prof_${c}_sq $OUT/prof_${c}_fetch $OUT/prof_${c}_write $OUT/prof_${c}_wait
  done
fi
du -sh $OUT
