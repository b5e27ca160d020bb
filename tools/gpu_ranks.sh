#!/bin/bash
# bench.py --gpus G as the driver's SCALE run launches it, as a DRY RUN on ONE GPU (all ranks on device 0, gloo for the
# process group: RCCL needs one device per rank): the sharded step's phases, the final rewards against one GPU's, the
# in-library exchange beside the process group's all-gather.  usage: G=8 CONFIGS="metric humanoidrun4096 ..." tools/gpu_ranks.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export MBD_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
G=${G:-8}
for c in ${CONFIGS:-metric humanoidrun4096 humanoidtrack2048demo sweep8}; do
  echo "== bench.py --config $c --gpus $G (one GPU, gloo) =="
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $G --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus $G --config $c --steps ${STEPS:-20} --warmup 5 --repeats ${REPEATS:-3} --no-cpu-baseline 2>gpurun_out/ranks_${c}_err.log | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=lambda v: round(v,3) if isinstance(v,float) else v
print('value', r(d['value']), '[', r(d['value_min']), r(d['value_max']), ']', d['unit'][:40], '| ms/step', r(d['ms_per_step']), '| N/GPU', d['config']['N_per_gpu'])
print('collective', d['config']['collective'][:70], '| rccl_world', d.get('rccl_world'), '|', d.get('dist_backend'))
print('scaling_expectation.strong:', (d.get('scaling_expectation') or {}).get('strong', '')[:110])
print('phase_ms', {k:r(v) for k,v in (d.get('phase_ms') or {}).items()})
print('other_collective', {k:r(v) for k,v in d.get('other_collective',{}).items() if k!='note'})
print('other_scaling', {k:r(v) for k,v in d.get('other_scaling',{}).items()})
print('per_rank', [r(x) for x in d.get('per_rank_plan_steps_per_sec',[])])
x=(d.get('extras') or {}).get('sweep8_replicas'); print('extras.sweep8_replicas', None if x is None else {k:r(v) for k,v in x.items() if k in ('plan_steps_per_sec','ranks_ok')})
fr=d['final_reward']; print('final_reward equals_one_gpu_bitwise', fr.get('equals_one_gpu_bitwise'), 'over', fr.get('sharded_over', fr.get('replicated_over')), 'mean', r(fr.get('mean')))
"
  tail -3 gpurun_out/ranks_${c}_err.log | cut -c1-300
done
