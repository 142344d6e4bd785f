#!/usr/bin/env python3
"""Copy what the judge reads from gpurun_out/<tag>/ (scratch, merged back by gpurun) into profiles/<tag>/ (tracked).  usage: tools/evidence_collect.py r05
Only summaries travel: bench lines, kernel-stats CSVs, per-step censuses, timing texts, parity tables, the GPU-test tail.  The PMC summaries
(profiles/<tag>/pmc_bench*.json) are written by tools/pmc_summarize.py on the GPU box's counter files and re-made here when those files are present."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = [
    "bench_line.json", "bench_line_fp8.json", "bench_line_64f.json", "bench_line_profiled.json", "bench_line_fp8_profiled.json",
    "bench_line_384.json", "bench_line_384_profiled.json", "bench_384_kernel_stats.csv",
    "bench_kernel_stats.csv", "bench_fp8_kernel_stats.csv", "per_step.json", "per_step_fp8.json", "step_timeline.json",
    "decode_timings.txt", "decode_kernel_stats.csv", "decode_fp8_timings.txt", "decode_fp8_kernel_stats.csv",
    "sam2_timings.txt", "sam2_kernel_stats.csv", "train_timings.txt", "train_kernel_stats.csv",
    "gemm_vs_vendor.json", "attn_vit_clock.txt", "pytest_gpu_tail.txt",
    "parity_table_bench.json", "parity_table_bench_full.json", "parity_table.json",
]


def write_perf_floors(dst):
    """profiles/<tag>/perf_floors.json from the two things it must agree with: the FLOOR dicts of the tests (recorded figure, margin) and the PERF_FLOOR lines of the
    committed GPU-test tail (what the run measured).  Written by this script only -- round 5's hand-made file drifted from both."""
    import json
    import re
    tail = os.path.join(dst, "pytest_gpu_tail.txt")
    if not os.path.isfile(tail):
        print("absent pytest_gpu_tail.txt: perf_floors.json not written")
        return
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    src = open(os.path.join(ROOT, "tests", "test_perf_floor_gpu.py")).read()
    floors = {k: float(v) for k, v in re.findall(r'"([a-z0-9_]+)":\s*([0-9.]+)', src[src.index("FLOOR_MS = {"):src.index("MARGIN =")])}
    margin = float(re.search(r"MARGIN = ([0-9.]+)", src).group(1))
    cfg = open(os.path.join(ROOT, "tests", "test_configs_gpu.py")).read()
    floors["config4_train_step_ms"] = float(re.search(r"CONFIG4_STEP_FLOOR_MS = ([0-9.]+)", cfg).group(1))
    rows = {}
    for line in open(tail):
        m = re.match(r"PERF_FLOOR (\w+): measured ([0-9.]+) (ms|us), recorded ([0-9.]+) (?:ms|us), limit ([0-9.]+)", line)
        if m:
            rows[m.group(1)] = dict(measured=float(m.group(2)), unit=m.group(3), recorded=float(m.group(4)), limit=float(m.group(5)))
    out = dict(source="profiles/%s/pytest_gpu_tail.txt (the run) + tests/test_perf_floor_gpu.py FLOOR_MS / tests/test_configs_gpu.py (the recorded figures); written by tools/evidence_collect.py" % os.path.basename(dst),
               margin=margin, floors={})
    for k, rec in sorted(floors.items()):
        r = rows.get(k)
        row = dict(recorded=rec, limit=round(rec * margin, 4), measured=r["measured"] if r else None, unit=r["unit"] if r else ("us" if k.endswith("_us") else "ms"),
                   within_limit=(r is None or r["measured"] <= rec * margin))
        if r and abs(r["recorded"] - rec) > 1e-6:           # the test's figure was moved AFTER the committed run (to the slow end that run showed): say so
            row["recorded_when_the_tail_was_taken"] = r["recorded"]
        out["floors"][k] = row
    with open(os.path.join(dst, "perf_floors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote perf_floors.json:", sum(1 for v in out["floors"].values() if v["measured"] is not None), "of", len(out["floors"]), "floors measured in the tail")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    for name in KEEP:
        p = os.path.join(src, name)
        if os.path.isfile(p) and os.path.getsize(p) > 0:
            if name.startswith("bench_line"):                 # keep the JSON line only (stdout may carry a warning line before it)
                with open(p) as f:
                    lines = [l for l in f.read().splitlines() if l.startswith("{")]
                if not lines:
                    print("no JSON line in", name); continue
                with open(os.path.join(dst, name), "w") as f:
                    f.write(lines[-1] + "\n")
            else:
                shutil.copyfile(p, os.path.join(dst, name))
            print("copied", name)
        else:
            print("absent", name)
    lab = os.path.join(src, "lab")
    if os.path.isdir(lab):
        os.makedirs(os.path.join(dst, "lab"), exist_ok=True)
        for name in sorted(os.listdir(lab)):
            if name.endswith(".txt") and os.path.getsize(os.path.join(lab, name)) > 0:
                with open(os.path.join(lab, name)) as f:
                    text = "".join(l for l in f if "amdgpu.ids" not in l)
                with open(os.path.join(dst, "lab", name), "w") as f:
                    f.write(text)
                print("copied lab/" + name)
    write_perf_floors(dst)
    for sub in ("pmc", "pmc_fp8"):
        if os.path.isdir(os.path.join(src, sub)):
            # gpurun merges the box's gpurun_out/ into the local one file by file: the counter files of EARLIER passes (another tree) stay beside the newest ones.
            # Keep the newest pass of every counter directory only -- a summary that averages two trees is no evidence for either.
            import glob
            for d in glob.glob(os.path.join(src, sub, "*", "*")):
                files = sorted(glob.glob(os.path.join(d, "*_counter_collection.csv")), key=os.path.getmtime)
                for old in files[:-1]:
                    for f in glob.glob(old.replace("_counter_collection.csv", "_*")):
                        os.remove(f)
            subprocess.call([sys.executable, os.path.join(ROOT, "tools", "pmc_summarize.py"), tag, sub], env=dict(os.environ, GRAFT_REPO_ROOT=ROOT))


if __name__ == "__main__":
    main()
