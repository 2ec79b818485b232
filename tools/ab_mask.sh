#!/bin/bash
# Round 4 A/B: SQFA_DPP_S_MASK bits 8 and 15 (partners lane ^ 8 / ^ 15 of the 16- and 32-lane groups through DPP row_ror:8 / row_mirror)
# on top of the shipped bit 7
V=variants/build
O=gpurun_out/r4/mask2.txt
mkdir -p gpurun_out/r4
: > $O
run() { echo "== $1" | tee -a $O; shift; python tools/ab_pairs.py "$@" 2>&1 | tee -a $O; }
run "f64 m=32 (2-D, 16 column lanes): shipped (bit 7) | bits 7, 8, 15" 600:32:smsqfa:f64 - $V/r4f_2d_m3.so -
run "f64 m=33 (2-D)" 600:32:sqfa:f64 - $V/r4f_2d_m3.so
run "f64 m=48 (2-D)" 300:48:smsqfa:f64 - $V/r4f_2d_m3.so
run "f32 m=48 (16 lanes)" 600:48:smsqfa - $V/r4f_wc_m3.so
run "f32 m=64 (32 lanes)" 300:64:smsqfa - $V/r4f_wc_m3.so
run "f32 m=16 (the factor pass K0b runs on 16 lanes: evaluation time)" 1000:16:smsqfa - $V/r4f_wc_m3.so
run "f32 m=32 (K0b on 32 lanes)" 1000:32:smsqfa - $V/r4f_wc_m3.so
