#!/bin/bash
# Lab: builds tools/lab/p2_<name> (attn_lab with a variant of the generated ViT attention body) for each "name:ENV..." argument, plain and with --stamps.
# usage: tools/lab/build_p2_variants.sh "base:" "ostore:UFV_P2_DROP=ostore" "p18:UFV_P2_OPT=p18" ...      (runs the builds in parallel)
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/tools/scratch
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  for st in "" "--stamps"; do
    tag=$name${st:+_st}
    ( env $envs UFV_P2_OUT=$R/tools/scratch/p2_$tag.inc python3 $R/tools/gen_attn_p2.py $st > /dev/null &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -DUFV_VIT_P2_ASM_FILE="\"$R/tools/scratch/p2_$tag.inc\"" $R/tools/lab/attn_lab.hip -o $R/tools/lab/p2_$tag 2> $R/tools/scratch/p2_$tag.err && echo built p2_$tag || echo FAILED p2_$tag ) &
    while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 2; done
  done
done
wait
