#!/bin/bash
# round 2, third pass: full GPU suite on the sign-bit / planar-contact forms, then a within-box A/B of kernel variants
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02c_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r02c_tests.log
tail -3 gpurun_out/r02c_tests.log
V=$GRAFT_REPO_ROOT/model-based-diffusion_amd/lib/variants
run() { MBD_HIP_LIB=$3 python bench.py --config $1 --steps 198 --warmup 20 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-16s %-6s' % ('$1', '$2'), 'steps/s %.1f  ms/step %.4f  rollout_kernel_ms %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))"; }
{
for round in 1 2 3; do
  run metric head ""
  for v in old exp1 exp2; do run metric $v $V/$v.so; done
done
for round in 1 2; do
  for c in hopper512 halfcheetah1024; do run $c head ""; run $c old $V/old.so; done
done
} | tee gpurun_out/r02c_ab.log
for v in exp1 exp2; do
  MBD_HIP_LIB=$V/$v.so timeout 600 python -m pytest tests -m gpu -x -q -k "humanoid" > gpurun_out/r02c_tests_$v.log 2>&1; echo "$v rc=$?"; tail -1 gpurun_out/r02c_tests_$v.log
done
