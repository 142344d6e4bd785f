R=$GRAFT_REPO_ROOT; TAG=r05; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_fp8 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --fp8 > $OUT/bench_line_fp8_profiled.json 2> $OUT/bench_fp8_profiled.err
cp $(find $OUT/bench_fp8 -name "*kernel_stats.csv" | head -1) $OUT/bench_fp8_kernel_stats.csv 2>/dev/null
rm -rf $OUT/bench_fp8
cd $R
UFV_BENCH_ARGS=--fp8 bash $R/tools/native_per_step.sh $TAG per_step_fp8 > $OUT/per_step_fp8.log 2>&1
rm -rf $OUT/per_step_fp8.d
python3 bench.py --steps 10 --warmup 3 --fp8 --no-cpu-baseline > $OUT/bench_line_fp8.json 2> $OUT/bench.err
tail -1 $OUT/bench_line_fp8.json | cut -c1-160
