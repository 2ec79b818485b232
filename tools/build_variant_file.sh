#!/bin/bash
# Build a variant of libsqfa_hip.so in which ONE translation unit other than a pair instantiation is compiled with
# extra flags:   tools/build_variant_file.sh NAME project_kernel "-DSQFA_PROJ_DEPTH=2"
# -> variants/build/NAME.so (select it with SQFA_HIP_LIBRARY=...; the installed library is never replaced).
set -e
cd "$(dirname "$0")/../sqfa_amd/csrc"
name=$1; unit=$2; extra=$3
out=../../variants/build; mkdir -p $out/obj_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -fno-gpu-rdc $extra -c $unit.hip -o $out/obj_$name/$unit.o
objs="$out/obj_$name/$unit.o"
for o in build/*.o; do
  [ "$(basename $o)" = "$unit.o" ] || objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $objs
echo "built $out/$name.so"
