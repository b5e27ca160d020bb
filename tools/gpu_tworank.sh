#!/bin/bash
# bench.py --gpus 2 as the driver launches it, as a dry run on ONE GPU (both ranks on device 0, gloo for the process group):
# what the sharded step's phases and the in-library exchange cost when two processes share a device
cd "$GRAFT_REPO_ROOT" || exit 1
export MBD_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'collective', d['config']['collective'][:60])
print('phase_ms', d.get('phase_ms'))
print('other_collective', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.get('other_collective',{}).items() if k!='note'})
print('other_scaling', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.get('other_scaling',{}).items()})
print('final_reward', d['final_reward'].get('equals_one_gpu_bitwise'), d['final_reward'].get('mean'))
"
