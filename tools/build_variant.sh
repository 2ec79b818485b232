#!/bin/bash
# Build a variant of libsqfa_hip.so for A/B timing on the GPU box:
#   tools/build_variant.sh NAME "EXTRA_FLAGS" tag:T:MR:G:CPL:TJ:WAVES [more configs...]
# recompiles only the listed pair-kernel configurations with EXTRA_FLAGS (the configuration may
# differ from configs.hpp only in TJ/WAVES-independent macros: geometry changes need a full
# build) and links them with the remaining objects of the regular build into
# variants/build/NAME.so (git-ignored: *.so, build/).  Time with tools/time_variants*.py.
set -e
cd "$(dirname "$0")/../sqfa_amd/csrc"
name=$1; extra=$2; shift 2
out=../../variants/build; mkdir -p $out/obj_$name
objs=""
skip=""
for cfg in "$@"; do
  IFS=: read tag T MR G CPL TJ WAVES <<< "$cfg"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -fno-gpu-rdc $extra \
    -DSQFA_TAG=$tag -DSQFA_T=$T -DSQFA_MR=$MR -DSQFA_G=$G -DSQFA_CPL=$CPL -DSQFA_TJ=$TJ -DSQFA_WAVES=$WAVES \
    -c pair_inst.hip -o $out/obj_$name/pair_${tag}_${MR}.o &
  objs="$objs $out/obj_$name/pair_${tag}_${MR}.o"
  skip="$skip pair_${tag}_${MR}.o"
done
wait
for o in build/*.o; do
  b=$(basename $o)
  case " $skip " in *" $b "*) ;; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $objs
echo "built $out/$name.so"
