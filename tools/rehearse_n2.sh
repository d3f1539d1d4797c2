#!/bin/bash
# The N > 1 code paths on ONE GPU (rates mean nothing): two ranks over gloo sharing the card, then one rank
# with the RCCL all-gather forced.  Output: gpurun_out/rehearsal_n2.json, gpurun_out/force_sharded.json
set -e
cd "$GRAFT_REPO_ROOT"
RAGFIN_DIST_BACKEND=gloo RAGFIN_SHARE_GPU=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 50 --warmup 5 > gpurun_out/rehearsal_n2.json 2> gpurun_out/rehearsal_n2.err
echo "n2 done"; tail -c 600 gpurun_out/rehearsal_n2.json; echo
RAGFIN_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --steps 100 --no-configs > gpurun_out/force_sharded.json 2> gpurun_out/force_sharded.err
echo "force-sharded done"; tail -c 400 gpurun_out/force_sharded.json; echo
