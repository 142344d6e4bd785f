# After a late change to a GEMM source or the SAM2 path: the PMC passes (fingerprinted to the GEMM sources), the aux profiles and the bench lines again -- not the whole round.
cd $GRAFT_REPO_ROOT; TAG=${1:-r06}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
bash tools/pmc_bench.sh $TAG pmc > $OUT/pmc.log 2>&1
UFV_BENCH_ARGS=--fp8 bash tools/pmc_bench.sh $TAG pmc_fp8 > $OUT/pmc_fp8.log 2>&1
bash tools/profile_aux.sh $TAG > $OUT/profile_aux.log 2>&1
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 10 --warmup 3 > $OUT/bench_line.json 2> $OUT/bench.err
python3 bench.py --steps 10 --warmup 3 --fp8 --no-cpu-baseline > $OUT/bench_line_fp8.json 2>> $OUT/bench.err
python3 -m pytest tests/test_perf_floor_gpu.py tests/test_sam2_gpu.py -m gpu -q 2>&1 | tail -16 > $OUT/refresh_tests.txt
tail -c 600 $OUT/bench_line.json; tail -16 $OUT/refresh_tests.txt
