#!/bin/bash
# Lab (GPU box): the bench on ONE box with two builds of the library, alternating (A = ${UFV_AB_BASE:-tools/scratch/libufv_r03.so}, by default the round-3 kernels; B = the in-tree build).
# The Python side is the current tree in both arms.  usage: tools/lab/ab_bench.sh [repeats]
R=$GRAFT_REPO_ROOT
N=${1:-3}
for i in $(seq $N); do
  for arm in A B; do
    if [ $arm = A ]; then export UFV_LAB=1 UFV_LIBRARY=$R/${UFV_AB_BASE:-tools/scratch/libufv_r03.so}; else unset UFV_LIBRARY; fi
    UFV_BENCH_NO_TIMER=1 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$arm', d['ms_per_step'])"
  done
done
