#!/bin/bash
# round 3, first GPU call: Infinity-Cache yardsticks + the RCCL branch at world size 1 + 8 ranks on one device
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3_c1
O=gpurun_out/r3_c1
timeout 900 ./build/ic_yardstick all > $O/ic_yardstick.txt 2>&1
echo "ic rc=$?"
export HSA_ENABLE_IPC_MODE_LEGACY=0
PFFT_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_rccl_w1.json 2> $O/bench_rccl_w1.err
echo "rccl w1 rc=$?"; cat $O/bench_rccl_w1.json
PFFT_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 8 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_8rank_one_device.json 2> $O/bench_8rank_one_device.err
echo "8rank rc=$?"; cat $O/bench_8rank_one_device.json
tail -5 $O/bench_8rank_one_device.err
