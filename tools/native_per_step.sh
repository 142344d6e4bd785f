#!/bin/bash
# Launches per TIMED STEP of every kernel of bench.py, by differencing two rocprofv3 --stats runs (4 and 14 timed steps, same warm-up): what the model
# build and the warm-up launch drops out.  Prints every at::native / rocclr row that still has a per-step count and writes gpurun_out/<tag>/per_step.json.
# usage (GPU box): [UFV_BENCH_ARGS=--fp8] tools/native_per_step.sh <tag> [out-name]    (out-name: per_step | per_step_fp8 ...)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r06}
NAME=${2:-per_step}
ARGS=${UFV_BENCH_ARGS:-}
OUT=$R/gpurun_out/$TAG/$NAME.d
mkdir -p $OUT
for K in 4 14; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k$K -- python3 $R/bench.py --steps $K --warmup 2 --no-cpu-baseline $ARGS > $OUT/k$K.json 2> $OUT/k$K.err
done
python3 - "$OUT" "$R/gpurun_out/$TAG/$NAME.json" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
def load(k):
    f = glob.glob(f"{out}/k{k}/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load(4), load(14)
rows = []
for n in sorted(set(a) | set(b)):
    ca, ta = a.get(n, (0, 0.0)); cb, tb = b.get(n, (0, 0.0))
    per, us = (cb - ca) / 10.0, (tb - ta) / 10.0 / 1e3
    if per > 0:
        rows.append({"kernel": n[:160], "launches_per_step": per, "us_per_step": round(us, 2)})
rows.sort(key=lambda r: -r["us_per_step"])
json.dump(rows, open(sys.argv[2], "w"), indent=1)
nat = [r for r in rows if "at::native" in r["kernel"] or "rocclr" in r["kernel"]]
print("kernels per timed step:", sum(r["launches_per_step"] for r in rows), " us:", round(sum(r["us_per_step"] for r in rows), 1))
print("at::native / rocclr rows with a per-step count:", len(nat))
for r in nat:
    print("  ", r)
PY
