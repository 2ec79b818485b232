#!/bin/bash
# Build a variant of libsqfa_hip.so in which the listed sizes run on the 2-D lane layout (pair_kernel_2d.hpp):
#   tools/build_variant_2d.sh NAME "EXTRA_FLAGS" tag:T:MR:GC:CPL:TJ:WAVES:RS [more...]
# -> variants/build/NAME.so (select with SQFA_HIP_LIBRARY).  The whole-column objects of the regular build stay linked.
set -e
cd "$(dirname "$0")/../sqfa_amd/csrc"
name=$1; extra=$2; shift 2
out=../../variants/build; mkdir -p $out/obj_$name
FLAGS="-O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -fno-gpu-rdc"
rows32=""; rows64=""
for cfg in "$@"; do
  IFS=: read tag T MR GC CPL TJ WAVES RS <<< "$cfg"
  row="X($T,$MR,$GC,$CPL,$TJ,$WAVES,$RS)"
  if [ "$tag" = "f32" ]; then rows32="$rows32 $row"; else rows64="$rows64 $row"; fi
done
defs=("-DSQFA_CONFIGS2D_F32(X)=$rows32" "-DSQFA_CONFIGS2D_F64(X)=$rows64")
objs=""
for cfg in "$@"; do
  IFS=: read tag T MR GC CPL TJ WAVES RS <<< "$cfg"
  /opt/rocm/bin/hipcc $FLAGS $extra "${defs[@]}" \
    -DSQFA_TAG=$tag -DSQFA_T=$T -DSQFA_MR=$MR -DSQFA_G=$GC -DSQFA_CPL=$CPL -DSQFA_TJ=$TJ -DSQFA_WAVES=$WAVES -DSQFA_RS=$RS \
    -c pair_inst.hip -o $out/obj_$name/pair2d_${tag}_${MR}.o &
  objs="$objs $out/obj_$name/pair2d_${tag}_${MR}.o"
done
/opt/rocm/bin/hipcc $FLAGS "${defs[@]}" -c sqfa_api.hip -o $out/obj_$name/sqfa_api.o &
objs="$objs $out/obj_$name/sqfa_api.o"
wait
for o in build/*.o; do
  b=$(basename $o)
  # the regular build's copy of an object that this variant rebuilt (sqfa_api.o, the listed 2-D rows) is left out
  [ "$b" = "sqfa_api.o" ] || [ -f "$out/obj_$name/$b" ] || objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $objs
echo "built $out/$name.so"
