cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for arm in A B; do
  if [ $arm = A ]; then export UFV_FP8_O_BF16=1; else unset UFV_FP8_O_BF16; fi
  UFV_BENCH_NO_TIMER=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --fp8 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$arm', d['ms_per_step'])"
done; done
