#!/bin/bash
# Round 4 A/B: SQFA_DPP_S_MASK bit 7 (shipped) against 0 on the rows not covered by tools/ab_f32_mirror.sh
V=variants/build
O=gpurun_out/r4/mask.txt
mkdir -p gpurun_out/r4
: > $O
run() { echo "== $1" | tee -a $O; shift; python tools/ab_pairs.py "$@" 2>&1 | tee -a $O; }
run "f64 m=24 (2-D): shipped (bit 7) | mask 0" 1000:24:smsqfa:f64 - $V/r4e_2d_m0.so -
run "f64 m=32 (2-D)" 600:32:smsqfa:f64 - $V/r4e_2d_m0.so
run "f64 m=33 (2-D)" 600:32:sqfa:f64 - $V/r4e_2d_m0.so
run "f32 m=40 (2-D)" 1000:40:smsqfa - $V/r4e_2d_m0.so
