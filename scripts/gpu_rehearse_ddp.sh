#!/bin/bash
# N>1 GPU code path rehearsal on a 1-GPU box: 2 ranks share cuda:0 and exchange gradients through gloo
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
CALM_DIST_BACKEND=gloo CALM_LOCAL_DEVICE=0 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --batch 16 --prof-steps 1 \
  > gpurun_out/rehearse_ddp.log 2>&1 || { tail -n 40 gpurun_out/rehearse_ddp.log; exit 1; }
tail -n 1 gpurun_out/rehearse_ddp.log | cut -c1-400
