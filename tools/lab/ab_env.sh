#!/bin/bash
# Lab (GPU box): the bench on ONE box with and without an environment switch of the in-tree library, alternating (A = the variable set, B = unset).
# usage: tools/lab/ab_env.sh UFV_NO_FUSED_ROPE [repeats]
R=$GRAFT_REPO_ROOT
V=$1
N=${2:-3}
for i in $(seq $N); do
  for arm in A B; do
    if [ $arm = A ]; then export $V=1; else unset $V; fi
    UFV_BENCH_NO_TIMER=1 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$arm ($V ' + ('set' if '$arm' == 'A' else 'unset') + ')', d['ms_per_step'])"
  done
done
