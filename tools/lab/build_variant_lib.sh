#!/bin/bash
# Lab: builds tools/scratch/libufv_<name>.so = the in-tree library with extra compiler flags on the GEMM translation units (e.g. '-DUFV_RESID_SC=" sc0 sc1"'),
# for a same-box A/B with tools/lab/ab_bench.sh (UFV_AB_BASE=tools/scratch/libufv_<name>.so).   usage: tools/lab/build_variant_lib.sh <name> <flags...>
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
D=$R/tools/scratch/var_$name
mkdir -p $D
C=$R/ufvideo_amd/csrc
FL="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -I$R/include"
for f in gemm256 gemm256_b gemm256_q gemm256_s gemm256_r gemm256_m gemm256_m2 gemm; do
  ( /opt/rocm/bin/hipcc $FL "$@" -c $C/$f.hip -o $D/$f.o 2> $D/$f.err || echo FAILED $f ) &
done
wait
OBJS=""
for o in $C/*.o; do b=$(basename $o); if [ -f $D/$b ]; then OBJS="$OBJS $D/$b"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/scratch/libufv_$name.so $OBJS && echo built tools/scratch/libufv_$name.so
