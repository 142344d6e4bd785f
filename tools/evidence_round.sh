#!/bin/bash
# Everything the round's evidence directory holds, in one GPU call (run on the GPU box; outputs under gpurun_out/<tag>/, copied into profiles/<tag>/ afterwards by
# tools/evidence_collect.py, which also runs tools/pmc_summarize.py on the merged counter files).   usage: tools/evidence_round.sh <tag>
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
bash $R/tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
bash $R/tools/native_per_step.sh $TAG per_step > $OUT/per_step.log 2>&1
UFV_BENCH_ARGS=--fp8 bash $R/tools/native_per_step.sh $TAG per_step_fp8 > $OUT/per_step_fp8.log 2>&1
rm -rf $OUT/per_step.d $OUT/per_step_fp8.d
python3 $R/tools/step_timeline.py $OUT/tl $OUT/step_timeline.json > $OUT/step_timeline.log 2>&1
rm -rf $OUT/tl
bash $R/tools/pmc_bench.sh $TAG pmc > $OUT/pmc.log 2>&1
UFV_BENCH_ARGS=--fp8 bash $R/tools/pmc_bench.sh $TAG pmc_fp8 > $OUT/pmc_fp8.log 2>&1
bash $R/tools/profile_aux.sh $TAG > $OUT/profile_aux.log 2>&1
cd $R
python3 tools/probe_blaslt.py $TAG > $OUT/gemm_vs_vendor.log 2>&1
LAB_P2_CLOCK=1 tools/lab/attn_lab_clock 14 > $OUT/attn_vit_clock.txt 2>&1
# the lab runs LABNOTES (round 6) quotes: raw outputs, copied to profiles/<tag>/lab/ by tools/evidence_collect.py
mkdir -p $OUT/lab
python3 tools/lab/gateup_half_time.py 2399 2799 4703 2304 1536 > $OUT/lab/gateup_half_time.txt 2>&1
python3 tools/lab/attn_729_time.py > $OUT/lab/attn_729_time.txt 2>&1
python3 tools/lab/seg_grad_norms.py > $OUT/lab/seg_grad_norms.txt 2>&1
tools/lab/grid_barrier_probe > $OUT/lab/grid_barrier_probe.txt 2>&1
UFV_PARITY_REPORT_BENCH=$OUT/parity_table_bench_full.json UFV_PARITY_FULL=mirror python3 -m pytest tests/test_bench_workload_gpu.py -m gpu -q -s > $OUT/parity_full.log 2>&1
UFV_PARITY_REPORT_BENCH=$OUT/parity_table_bench.json python3 -m pytest tests/test_bench_workload_gpu.py -m gpu -q -s > $OUT/parity_bench.log 2>&1
UFV_PARITY_REPORT=$OUT/parity_table.json python3 -m pytest tests/test_parity_bf16_gpu.py -m gpu -q -s > $OUT/parity_bf16.log 2>&1
(time python3 -m pytest tests -m gpu -q 2>&1 | grep -E "^PERF_FLOOR|^PARITY|^CONFIG|^GEOM384|^ROUNDING|TOWER_STREAM|passed|failed|FAILED|error|AssertionError|measured by the tests" ) > $OUT/pytest_gpu_tail.txt 2>&1
tail -3 $OUT/pytest_gpu_tail.txt
