#!/bin/bash
# Rehearsal of the multi-rank control flow on ONE GPU (every rank on device 0, collectives through the host over gloo): NOT a measurement -
# it checks that `torchrun ... bench.py --gpus 2` completes in replica mode (with the probe child) and with the sharded update.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp LVK_BENCH_BACKEND=gloo LVK_BENCH_ONE_GPU=1
O=gpurun_out/${1:-rehearsal}; mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 40 --warmup 5 --no-cpu-baseline > $O/replicas.json 2> $O/replicas.err; echo "replicas rc $?"; tail -c 400 $O/replicas.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --backend-only --sharded --steps 40 --warmup 4 > $O/sharded_be.json 2> $O/sharded_be.err; echo "sharded backend-only rc $?"; tail -c 600 $O/sharded_be.json
