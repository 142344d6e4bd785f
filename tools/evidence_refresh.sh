cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
bash tools/pmc_bench.sh r05 pmc > $OUT/pmc.log 2>&1
UFV_BENCH_ARGS=--fp8 bash tools/pmc_bench.sh r05 pmc_fp8 > $OUT/pmc_fp8.log 2>&1
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err
python3 bench.py --steps 10 --warmup 3 --fp8 --no-cpu-baseline > $OUT/bench_line_fp8.json 2>> $OUT/bench.err
python3 bench.py --steps 5 --warmup 2 --frames 64 --no-cpu-baseline > $OUT/bench_line_64f.json 2>> $OUT/bench.err
(time python3 -m pytest tests -m gpu -q -s 2>&1 | grep -E "PERF_FLOOR|TOWER_STREAM|passed|failed|FAILED|error|AssertionError" ) > $OUT/pytest_gpu_tail.txt 2>&1
tail -4 $OUT/pytest_gpu_tail.txt
for f in bench_line.json bench_line_fp8.json bench_line_64f.json; do tail -1 $OUT/$f | cut -c1-150; done
