#!/bin/bash
# round-1 measurement pass: tests, bench line, rocprofv3 kernel stats + PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
python __graft_entry__.py 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 60 --warmup 10 2>$OUT/bench_err.log | tail -1 > $OUT/bench_r01.json
cat $OUT/bench_r01.json
# A/B: SLP vectorizer off
if [ -f model-based-diffusion_amd/lib/libmbd_hip_noslp.so ]; then
  MBD_HIP_LIB=$R/model-based-diffusion_amd/lib/libmbd_hip_noslp.so python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 > $OUT/bench_noslp.json
  cat $OUT/bench_noslp.json
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o r01 -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-final-reward > $OUT/prof_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace -d $OUT/prof_pmc_sq -o r01 -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-final-reward > $OUT/prof_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_pmc_fetch -o r01 -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-final-reward > $OUT/prof_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace -d $OUT/prof_pmc_write -o r01 -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-final-reward > $OUT/prof_pmc_write.log 2>&1
find $OUT -name "*.csv" | head -30
du -sh $OUT
