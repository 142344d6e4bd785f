cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for arm in A B; do
  if [ $arm = A ]; then unset UFV_TOWER_STREAM; else export UFV_TOWER_STREAM=bf16; fi
  UFV_BENCH_NO_TIMER=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --fp8 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$arm', d['ms_per_step'])"
done; done
unset UFV_TOWER_STREAM
python3 -m pytest tests/test_fp8_gpu.py -m gpu -q 2>&1 | tail -3
