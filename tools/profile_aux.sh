#!/bin/bash
# Evidence for the paths outside the headline metric (GPU box): rocprofv3 kernel-trace stats + the tools' own timing lines for greedy decode, the SAM2-L
# segmentation path and the decoder training step, all at full dimensions.  usage: tools/profile_aux.sh <tag>  -> gpurun_out/<tag>/{decode,sam2,train}_*
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for name in decode decode_fp8 sam2 train; do
  case $name in
    decode) SCRIPT="$R/tools/bench_decode.py";;
    decode_fp8) SCRIPT="$R/tools/bench_decode.py --fp8";;
    sam2)   SCRIPT="$R/tools/bench_sam2.py";;
    train)  SCRIPT="$R/tools/bench_train.py --steps 3 --warmup 1";;
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- python3 $SCRIPT > $OUT/${name}_timings.txt 2> $OUT/${name}.err
  cp $(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/prof_$name
  echo "== $name"; tail -8 $OUT/${name}_timings.txt
done
# the reference's per-device training shape (scripts/train/train_1121v1.sh: batch 2 x 32 frames, --gradient_checkpointing True): two micro-batches of S = 2399 per step
python3 $R/tools/bench_train.py --steps 3 --warmup 1 --micro-batches 2 >> $OUT/train_timings.txt 2>> $OUT/train.err
python3 $R/tools/bench_train.py --steps 3 --warmup 1 --micro-batches 2 --gradient-checkpointing >> $OUT/train_timings.txt 2>> $OUT/train.err
tail -2 $OUT/train_timings.txt
