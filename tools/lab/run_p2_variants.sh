#!/bin/bash
# Lab (GPU box): time every tools/lab/p2_<name> on the ViT shape (kernel id 14), bit-compare with kernel 11, and print the per-period stamps of the _st builds.
R=$GRAFT_REPO_ROOT
for name in "$@"; do
  echo "=== $name"
  for rep in 1 2; do LAB_CMP=11 $R/tools/lab/p2_$name 14 2>/dev/null | grep -E "^kernel|different BITS"; done
  if [ -x $R/tools/lab/p2_${name}_st ]; then LAB_P2_STAMPS=1 $R/tools/lab/p2_${name}_st 14 2>/dev/null | grep -A5 "^stamps"; fi
done
