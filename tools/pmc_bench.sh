#!/bin/bash
# HBM-side traffic and MFMA busy of the bench's kernels (run on the GPU box): separate --pmc passes of the SAME command the bench line comes
# from, as MI355X_MICROARCH.md prescribes (no trace domains beside the counters).  usage: tools/pmc_bench.sh <tag>  -> gpurun_out/<tag>/pmc_bench.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
OUT=$R/gpurun_out/$TAG/pmc
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/$C.err
done
rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/BUSY -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/BUSY.err
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for sub in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "BUSY"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"]
            if "gemm_nt" in n or "attn_fwd" in n:
                key = n.split("(")[0].replace("void (anonymous namespace)::", "")
                acc[(key, row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in acc.items():
            res[k][c + "_mean"] = sum(v) / len(v); res[k]["launches_" + c] = len(v)
json.dump(res, open("$R/gpurun_out/$TAG/pmc_bench.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(res.items()):
    print(k[:90], {c: round(v, 1) for c, v in d.items() if c.endswith("_mean")})
PY
