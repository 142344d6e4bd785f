#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the default bench run + bench lines for --fp8 / --frames 64 (same build).
# usage: tools/profile_round.sh <tag>      (outputs under gpurun_out/<tag>/, copy what should be judged into profiles/)
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_line_profiled.json 2> $OUT/bench_profiled.err
cp $(find $OUT/bench -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/bench
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_fp8 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --fp8 > $OUT/bench_line_fp8_profiled.json 2> $OUT/bench_fp8_profiled.err
cp $(find $OUT/bench_fp8 -name "*kernel_stats.csv" | head -1) $OUT/bench_fp8_kernel_stats.csv 2>/dev/null
rm -rf $OUT/bench_fp8
cd $R
python3 bench.py --steps 10 --warmup 3 > $OUT/bench_line.json 2> $OUT/bench.err
python3 bench.py --steps 10 --warmup 3 --fp8 --no-cpu-baseline > $OUT/bench_line_fp8.json 2>> $OUT/bench.err
python3 bench.py --steps 5 --warmup 2 --frames 64 --no-cpu-baseline > $OUT/bench_line_64f.json 2>> $OUT/bench.err
# the released checkpoint's own geometry (siglip-so400m-patch14-384: 729 tokens per frame, 2704 visual tokens): a SECONDARY line + its kernel table
python3 bench.py --steps 10 --warmup 3 --img 384 > $OUT/bench_line_384.json 2>> $OUT/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench384 -- python3 $R/bench.py --steps 10 --warmup 3 --img 384 > $OUT/bench_line_384_profiled.json 2> $OUT/bench_384_profiled.err
cp $(find $OUT/bench384 -name "*kernel_stats.csv" | head -1) $OUT/bench_384_kernel_stats.csv 2>/dev/null
rm -rf $OUT/bench384
cd $R
tail -n 1 $OUT/bench_line_384.json | cut -c1-200
tail -n 1 $OUT/bench_line.json | cut -c1-300
tail -n 1 $OUT/bench_line_fp8.json | cut -c1-200
tail -n 1 $OUT/bench_line_64f.json | cut -c1-200
head -12 $OUT/bench_kernel_stats.csv | cut -c1-160
