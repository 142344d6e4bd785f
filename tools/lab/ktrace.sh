#!/bin/bash
# Lab (GPU box): per-kernel durations of a python lab script from rocprofv3's kernel trace (not from host-side event timing, which sees launch pitch on
# kernels of a few microseconds).   usage: tools/lab/ktrace.sh <name> <script.py> [args...]   -> prints the top rows of the stats table
R=$GRAFT_REPO_ROOT
NAME=$1; shift
OUT=$R/gpurun_out/lab/kt_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/$@ > $OUT/stdout.txt 2>&1
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f}")
PY
rm -rf $OUT
