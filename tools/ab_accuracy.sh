#!/bin/bash
# Developer aid: tools/grad_accuracy.py for several pre-built library variants:  tools/ab_accuracy.sh "SPECS" lib1.so lib2.so ...
specs=$1; shift
for lib in "$@"; do
  echo "== $(basename $lib .so)"
  SQFA_HIP_LIBRARY=$(realpath $lib) python3 tools/grad_accuracy.py $specs 2>&1 | grep -v amdgpu.ids
done
