#!/bin/bash
# `bench.py --gpus 2` exactly as the driver launches it, both ranks on ONE card (host transport: RCCL refuses two ranks per
# device), at the full per-GPU size of configs[3] (16.5 M rows per rank).  bash tools/rehearse_n2.sh
O=gpurun_out/r02z; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 SAENA_BENCH_NO_RCCL=1 SAENA_BENCH_DEVICE=0
S=$(date +%s)
timeout -k 10 1000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_n2.json 2> $O/bench_n2.err
rc=$?
echo "rc=$rc wall=$(( $(date +%s) - S ))s"
tail -c 3000 $O/bench_n2.json
tail -5 $O/bench_n2.err
exit $rc
