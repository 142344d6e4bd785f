#!/usr/bin/env python3
"""One timed step of bench.py, launch by launch (GPU box): `rocprofv3 --kernel-trace` of a short run, then the dispatches between the last two `patchify*`
launches (= one whole step: tower, connector, splice, prefill) in start order with their durations and the idle gap in front of each.
usage: tools/step_timeline.py <trace-dir> <out.json> [bench args...]     (run as: python3 tools/step_timeline.py gpurun_out/r05/tl gpurun_out/r05/step_timeline.json)
The trace itself is taken by a child process (rocprofv3 ... -- python3 bench.py); this script never touches the GPU."""
import csv
import glob
import json
import os
import re
import subprocess
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)(I|E)", name)
    if m:
        return m.group(1)
    return name.split("(")[0][:90]


def main():
    d, out = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
    extra = sys.argv[3:]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.check_call(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(root, "bench.py"),
                           "--steps", "3", "--warmup", "2", "--no-cpu-baseline"] + extra, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "patchify" in r[2]]
    a, b = starts[-2], starts[-1]
    step = rows[a:b]
    tl, prev_end = [], None
    for s, e, n in step:
        tl.append({"kernel": short(n), "us": round((e - s) / 1e3, 2), "gap_us": round((s - prev_end) / 1e3, 2) if prev_end is not None else 0.0})
        prev_end = e
    agg = {}
    for t in tl:
        g = agg.setdefault(t["kernel"], {"launches": 0, "us": 0.0, "min_us": 1e9, "max_us": 0.0})
        g["launches"] += 1; g["us"] += t["us"]; g["min_us"] = min(g["min_us"], t["us"]); g["max_us"] = max(g["max_us"], t["us"])
    summary = {"launches": len(tl), "kernel_us": round(sum(t["us"] for t in tl), 1), "gap_us": round(sum(t["gap_us"] for t in tl), 1),
               "wall_us": round((step[-1][1] - step[0][0]) / 1e3, 1)}
    json.dump({"summary": summary, "by_kernel": {k: {kk: round(vv, 2) for kk, vv in v.items()} for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["us"])},
               "timeline": tl}, open(out, "w"), indent=0)
    print(summary)


if __name__ == "__main__":
    main()
