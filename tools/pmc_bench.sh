#!/bin/bash
# HBM-side traffic and MFMA busy of the bench's kernels (run on the GPU box): separate --pmc passes of the SAME command the bench line comes
# from, as MI355X_MICROARCH.md prescribes (no trace domains beside the counters).  usage: [UFV_BENCH_ARGS=--fp8] tools/pmc_bench.sh <tag> [pmc | pmc_fp8]  -> profiles/<tag>/pmc_bench[_fp8].json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r06}
ARGS=${UFV_BENCH_ARGS:-}
SUB=${2:-pmc}
OUT=$R/gpurun_out/$TAG/$SUB
rm -rf $OUT          # (gpurun merges gpurun_out/ back file by file: counter files of an earlier pass would be averaged in)
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $ARGS > /dev/null 2> $OUT/$C.err
done
rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/BUSY -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $ARGS > /dev/null 2> $OUT/BUSY.err
# wave lifetimes in shader cycles (quad-cycles x 4): the denominator that does not depend on GRBM_GUI_ACTIVE, which reads high on launches well under 0.3 ms
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/WAVE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $ARGS > /dev/null 2> $OUT/WAVE.err
python3 $R/tools/pmc_summarize.py $TAG $SUB
