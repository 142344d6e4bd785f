#!/bin/bash
# elapsed shader cycles (GRBM_GUI_ACTIVE / 8) and MFMA-busy cycles of the bench's GEMM kernels for the two ping-pong schedules, same box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-clock}
mkdir -p $OUT
for MODE in two four; do
  if [ $MODE = four ]; then export UFV_GEMM_4PHASE=1; else unset UFV_GEMM_4PHASE; fi
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/$MODE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/$MODE.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${MODE}_t -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>> $OUT/$MODE.err
done
python3 - <<PY
import csv, glob, collections
for mode in ("two", "four"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % mode, recursive=True):
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"]
            if "gemm_nt_256<false, true" in n: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur = []
    for f in glob.glob("$OUT/%s_t/**/*kernel_stats.csv" % mode, recursive=True):
        for row in csv.DictReader(open(f)):
            if "gemm_nt_256<false, true" in row["Name"]: dur.append(float(row["AverageNs"]) / 1e3)
    cyc = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8
    busy = sum(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024
    print("%s phases: gate/up %.1f us (kernel trace), %.0f elapsed cycles -> %.2f GHz, MFMA busy %.1f %%" % (mode, dur[0], cyc, cyc / dur[0] / 1e3, 100 * busy / cyc))
PY
