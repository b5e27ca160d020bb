#!/bin/bash
# 2-rank dry run of bench.py on a ONE-GPU box (both ranks on device 0, gloo for the exchange): validates
# sharding, the step loop, the max-over-ranks timing and the JSON line. Not a performance number.
cd "$GRAFT_REPO_ROOT" || exit 1
MBD_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 3 2>&1 | tail -3
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-final-reward 2>&1 | tail -1 | cut -c1-300
