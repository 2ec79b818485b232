#!/bin/bash
# Build a variant of libsqfa_hip.so for A/B timing on the GPU box:
#   tools/build_variant.sh NAME "EXTRA_FLAGS" tag:T:MR:G:CPL:TJ:WAVES [more configs...]
# recompiles the listed pair-kernel configurations with EXTRA_FLAGS and -- because the geometry table of
# sqfa_api.hip must agree with them -- sqfa_api.o with the same rows overridden (configs.hpp: one macro per
# row), then links them with the remaining objects of the regular build into variants/build/NAME.so
# (git-ignored).  Select a variant at run time with SQFA_HIP_LIBRARY=variants/build/NAME.so (sqfa_amd/_lib.py);
# the installed library is never replaced.  Time with tools/time_variants_any.py.
set -e
cd "$(dirname "$0")/../sqfa_amd/csrc"
name=$1; extra=$2; shift 2
out=../../variants/build; mkdir -p $out/obj_$name
FLAGS="-O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -fno-gpu-rdc"
objs=""
skip=" sqfa_api.o "
rows=()
for cfg in "$@"; do
  IFS=: read tag T MR G CPL TJ WAVES <<< "$cfg"
  TAG=$(echo $tag | tr a-z A-Z)
  rows+=("-DSQFA_ROW_${TAG}_${MR}(X)=X($T,$MR,$G,$CPL,$TJ,$WAVES)")
done
for cfg in "$@"; do
  IFS=: read tag T MR G CPL TJ WAVES <<< "$cfg"
  /opt/rocm/bin/hipcc $FLAGS $extra "${rows[@]}" \
    -DSQFA_TAG=$tag -DSQFA_T=$T -DSQFA_MR=$MR -DSQFA_G=$G -DSQFA_CPL=$CPL -DSQFA_TJ=$TJ -DSQFA_WAVES=$WAVES \
    -c pair_inst.hip -o $out/obj_$name/pair_${tag}_${MR}.o &
  objs="$objs $out/obj_$name/pair_${tag}_${MR}.o"
  skip="$skip pair_${tag}_${MR}.o "
done
/opt/rocm/bin/hipcc $FLAGS "${rows[@]}" -c sqfa_api.hip -o $out/obj_$name/sqfa_api.o &
objs="$objs $out/obj_$name/sqfa_api.o"
wait
for o in build/*.o; do
  b=$(basename $o)
  case "$skip" in *" $b "*) ;; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $objs
echo "built $out/$name.so"
