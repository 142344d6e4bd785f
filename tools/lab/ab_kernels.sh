#!/bin/bash
# Lab (GPU box): per-kernel mean durations of the bench under rocprofv3 with two builds of the library (A = $UFV_AB_BASE, B = in-tree), GEMM and attention rows side by side.
# usage: UFV_AB_BASE=tools/scratch/libX.so tools/lab/ab_kernels.sh <tag>
R=$GRAFT_REPO_ROOT
TAG=${1:-ab}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for arm in A B; do
  if [ $arm = A ]; then export UFV_LAB=1 UFV_LIBRARY=$R/${UFV_AB_BASE:-tools/scratch/libufv_r03.so}; else unset UFV_LIBRARY; fi
  UFV_BENCH_NO_TIMER=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$arm -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/line_$arm.json 2> $OUT/err_$arm.txt
  cp $(find $OUT/prof_$arm -name "*kernel_stats.csv" | head -1) $OUT/stats_$arm.csv
  rm -rf $OUT/prof_$arm
done
python3 - "$OUT" <<'PY'
import csv, sys
out = sys.argv[1]
def load(a):
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f"{out}/stats_{a}.csv"))}
A, B = load("A"), load("B")
tot = 0.0
for n in sorted(A, key=lambda k: -A[k][0] * A[k][1]):
    if n in B and ("gemm_nt" in n or "attn_fwd" in n):
        ca, ua = A[n]; cb, ub = B[n]
        d = (ub - ua) * cb / 13.0 / 1e3
        tot += d
        print(f"{n[:118]:118s} calls {ca:5d}  A {ua:8.1f} us  B {ub:8.1f} us  {100 * (ub / ua - 1):+5.1f} %  {d:+.3f} ms/step")
print(f"sum over these rows: {tot:+.3f} ms per step")
PY
