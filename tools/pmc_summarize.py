#!/usr/bin/env python3
"""gpurun_out/<tag>/pmc/*/**/counter_collection.csv (tools/pmc_bench.sh) -> profiles/<tag>/pmc_bench.json.
Per kernel: means of FETCH_SIZE / WRITE_SIZE (KB) / SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE and the launch duration, plus
the three summary entries bench.py and DESIGN.md quote (gate/up GEMM, ViT attention, LLM attention).  Corrections as MI355X_MICROARCH.md
prescribes: FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950 (doubled here), WRITE_SIZE is exact, both sit on the L2's
memory side (Infinity-Cache hits included); GRBM_GUI_ACTIVE sums the 8 XCDs (/ 8 = elapsed shader cycles), SQ_VALU_MFMA_BUSY_CYCLES sums the
1024 SIMDs (/ 1024 = busy cycles of one matrix pipe)."""
import collections, csv, glob, json, os, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
sub = sys.argv[2] if len(sys.argv) > 2 else "pmc"          # "pmc" (bf16 bench) | "pmc_fp8" (bench.py --fp8): the sub-directory tools/pmc_bench.sh wrote
fp8 = sub.endswith("fp8")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag, sub)


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)          # argument list
    return n.strip()


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "BUSY", "WAVE"):
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"]
            if "gemm_nt" in n or "attn_fwd" in n or "layernorm_k" in n or "rmsnorm_k" in n:
                k = short(n)
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row["Counter_Name"] in ("GRBM_GUI_ACTIVE",):
                    acc[k]["duration_ns"].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
means = {}
for k, d in sorted(acc.items()):
    means[k] = {c + "_mean": sum(v) / len(v) for c, v in d.items()}
    means[k]["launches"] = len(d.get("GRBM_GUI_ACTIVE", d.get("SQ_VALU_MFMA_BUSY_CYCLES", [])))
    if "GRBM_GUI_ACTIVE_mean" in means[k] and "SQ_VALU_MFMA_BUSY_CYCLES_mean" in means[k]:
        means[k]["elapsed_cycles"] = means[k]["GRBM_GUI_ACTIVE_mean"] / 8.0
        means[k]["mfma_busy_fraction"] = (means[k]["SQ_VALU_MFMA_BUSY_CYCLES_mean"] / 1024.0) / means[k]["elapsed_cycles"]
    if "SQ_WAVE_CYCLES_mean" in means[k] and "SQ_WAVES_mean" in means[k] and "SQ_VALU_MFMA_BUSY_CYCLES_mean" in means[k] and means[k]["SQ_WAVES_mean"] > 0:
        # mean lifetime of a wave in shader cycles (SQ_WAVE_CYCLES counts quad-cycles), and the matrix pipe's busy cycles per SIMD over the time its
        # waves were resident: waves / 1024 SIMDs waves share a pipe one after (or beside) the other, so resident time per SIMD = lifetime x waves / 1024
        # when they run one at a time (attn_fwd_vit72_p2: 1024 waves, one per SIMD) -- an upper bound on residency otherwise
        life = 4.0 * means[k]["SQ_WAVE_CYCLES_mean"] / means[k]["SQ_WAVES_mean"]
        means[k]["wave_lifetime_cycles"] = life
        means[k]["mfma_busy_per_simd_cycles"] = means[k]["SQ_VALU_MFMA_BUSY_CYCLES_mean"] / 1024.0
        means[k]["mfma_busy_fraction_of_wave_lifetime"] = means[k]["mfma_busy_per_simd_cycles"] / life if means[k]["SQ_WAVES_mean"] <= 1024.5 else None


def summary(match, label, alg_bytes=None):
    ks = [k for k in means if match(k)]
    if not ks:
        return None
    k = max(ks, key=lambda x: means[x].get("launches", 0))
    m = means[k]
    out = {"kernel": label, "kernel_symbol": k, "launches": m.get("launches"), "elapsed_cycles": round(m.get("elapsed_cycles", 0)),
           "mfma_busy_fraction": round(m.get("mfma_busy_fraction", 0), 4), "duration_us_profiled": round(m.get("duration_ns_mean", 0) / 1e3, 1)}
    if m.get("mfma_busy_fraction_of_wave_lifetime") is not None:
        out["mfma_busy_fraction_of_wave_lifetime"] = round(m["mfma_busy_fraction_of_wave_lifetime"], 4)
        out["wave_lifetime_cycles"] = round(m["wave_lifetime_cycles"])
    if "FETCH_SIZE_mean" in m and "WRITE_SIZE_mean" in m:
        out["fetch_bytes"] = int(m["FETCH_SIZE_mean"] * 1024 * 2)
        out["write_bytes"] = int(m["WRITE_SIZE_mean"] * 1024)
        out["hbm_bytes_per_launch"] = out["fetch_bytes"] + out["write_bytes"]
    if alg_bytes:
        out["algorithmic_bytes_per_launch"] = alg_bytes
    return out


sys.path.insert(0, root)
import bench as _bench          # noqa: E402  (only for the source fingerprint; nothing of it runs)
res = {"gemm_sources_sha256": _bench.gemm_sources_sha256(),
       "source": f"tools/pmc_bench.sh {tag}: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES | SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, one pass each, no trace "
                 "domains, on `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline" + (" --fp8" if fp8 else "") + "`; summarised by tools/pmc_summarize.py",
       "correction": "gfx950: FETCH_SIZE (KB) reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE (KB) exact; counters sit on the "
                     "L2's memory side: Infinity-Cache hits included.  GRBM_GUI_ACTIVE sums the 8 XCDs -> / 8 = elapsed shader cycles (reads high on launches shorter than "
                     "~0.3 ms); SQ_VALU_MFMA_BUSY_CYCLES sums the 1024 SIMDs",
       "per_kernel_means": means}
M, N, K = 2399, 37888, 3584
# (--fp8: e4m3 operands, 1 byte per element; the SwiGLU output is e4m3 + block scales in the fused MX chain)
res["gate_up"] = summary(lambda k: k.startswith("gemm_nt_256<false, true"), f"gemm_nt_256<{'e4m3' if fp8 else 'bf16'}, swiglu, 256x256, two-phase> M={M} N={N} K={K}",
                         alg_bytes=(M * K + N * K + M * (N // 2)) * (1 if fp8 else 2))
res["vit_attention"] = summary(lambda k: "attn_fwd_vit72" in k, "ViT attention B=32 H=16 S=576 hd=72", alg_bytes=32 * 576 * 1152 * 2 * 4)
res["llm_attention"] = summary(lambda k: "attn_fwd_c128" in k or "attn_fwd_mfma<128" in k, "causal attention S=2399 28/4 heads hd=128 (attn_fwd_c128)")
# the in-kernel clock of the ViT attention kernel (diagnostic build, tools/lab/attn_lab.hip LAB_P2_CLOCK; MI355X_MICROARCH.md DVFS item 6): settles which denominator
# the MFMA-busy figure gets -- GRBM_GUI_ACTIVE / 8 reads high on a 74 us dispatch (it implies 2.3-2.5 GHz), the stamps give the clock the kernel really ran at
clock_file = os.path.join(root, "gpurun_out", tag, "attn_vit_clock.txt")
if res.get("vit_attention") and os.path.exists(clock_file):
    for line in open(clock_file):
        if line.startswith("{"):
            ck = json.loads(line)
            res["vit_attention"]["in_kernel_clock"] = dict(
                ck, source="tools/lab/attn_lab_clock 14 (LAB_P2_CLOCK=1): >= 2 s of back-to-back launches on random operands, then s_memtime / s_memrealtime stamped once "
                           "around every wave's pass loop of the last launch; MFMA-busy cycles per SIMD = 81 periods x 22 v_mfma_f32_32x32x16_bf16 x 32 cycles = what "
                           "SQ_VALU_MFMA_BUSY_CYCLES / 1024 counts")
            res["vit_attention"]["mfma_busy_fraction_measured_clock"] = ck["mfma_busy_over_wall_x_clock"]
out = os.path.join(root, "profiles", tag)
os.makedirs(out, exist_ok=True)
json.dump(res, open(os.path.join(out, "pmc_bench_fp8.json" if fp8 else "pmc_bench.json"), "w"), indent=1, sort_keys=True)
for k in ("gate_up", "vit_attention", "llm_attention"):
    print(k, json.dumps(res[k]))
