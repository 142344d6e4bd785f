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
    "bench_kernel_stats.csv", "bench_fp8_kernel_stats.csv", "per_step.json", "per_step_fp8.json", "step_timeline.json",
    "decode_timings.txt", "decode_kernel_stats.csv", "decode_fp8_timings.txt", "decode_fp8_kernel_stats.csv",
    "sam2_timings.txt", "sam2_kernel_stats.csv", "train_timings.txt", "train_kernel_stats.csv",
    "gemm_vs_vendor.json", "attn_vit_clock.txt", "pytest_gpu_tail.txt",
    "parity_table_bench.json", "parity_table_bench_full.json", "parity_table.json", "perf_floors.json",
]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
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
    for sub in ("pmc", "pmc_fp8"):
        if os.path.isdir(os.path.join(src, sub)):
            subprocess.call([sys.executable, os.path.join(ROOT, "tools", "pmc_summarize.py"), tag, sub], env=dict(os.environ, GRAFT_REPO_ROOT=ROOT))


if __name__ == "__main__":
    main()
