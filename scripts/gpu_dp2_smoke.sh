#!/bin/bash
# gpurun --timeout 300 -- "bash scripts/gpu_dp2_smoke.sh"
# Two ranks of bench.py on the ONE GPU of the box (gloo group, both on cuda:0): exercises rendezvous, the gradient
# all-reduce inside TrainEngine.step, barriers and the max-over-ranks timing of the N > 1 path.  Not a measurement.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
NERFHIP_BENCH_ONE_DEVICE=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 > gpurun_out/dp2.log 2>&1; echo "rc=$?" >> gpurun_out/dp2.log
tail -3 gpurun_out/dp2.log | cut -c1-900
