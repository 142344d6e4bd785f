#!/bin/bash
# PMC passes for the attention lab (run on the GPU box): usage tools/lab/pmc_attn.sh <kernel id> <tag> [B H S hd causal Hkv] [lab binary]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; TAG=$2; ARGS="$3"; BIN=${4:-attn_lab}
mkdir -p $R/gpurun_out/pmc_$TAG
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_$TAG/a -- $R/tools/lab/$BIN $K $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_$TAG/b -- $R/tools/lab/$BIN $K $ARGS > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("a", "b"):
    for f in glob.glob("$R/gpurun_out/pmc_$TAG/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "attn" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()):
            v = sorted(v); print("%-28s median %.4g  (n=%d)" % (k, v[len(v)//2], len(v)))
PY
