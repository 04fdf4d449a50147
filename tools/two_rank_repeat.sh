#!/bin/bash
# N runs of the two-process / one-GPU training bench (what tests/test_gpu_train.py::test_bench_train_two_ranks_on_one_device runs once) under a set of
# environment switches: how round 5 found the unordered zero-fill of a lane's BatchNorm scratch (train_ops._tzeros) — a non-finite loss in about every
# second run, never with MAF_TRAIN_LANES=0 or MAF_STEP_TAPE=0, never in a process that has the GPU to itself.
#   gpurun -- 'N=20 bash tools/two_rank_repeat.sh'            (edit the `run` lines at the bottom for an A/B)
run() { # label, env...
  local label=$1; shift
  local fails=0
  for i in $(seq 1 $N); do
    env "$@" MAF_BENCH_ONE_DEVICE=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --train --batch 2 --steps 2 --warmup 1 --no-cpu-baseline --dist-backend gloo > /tmp/o.json 2> /tmp/e.log || fails=$((fails+1))
  done
  echo "$label: $fails / $N failed"
}
N=${N:-20}
run fixed X=1
