#!/bin/bash
# Round 4 A/B: share of the partner rows that float64 4-lane groups move through ds_swizzle (SQFA_SWZ_ROWS_OF_8; the float64
# default was "all 8 of 8": two 32-bit crossbar operations per element) -- profiles/r4_pairs_fewer_lanes.txt, float64 part.
V=variants/build
O=gpurun_out/r4/f64_swizzle.txt
mkdir -p gpurun_out/r4
: > $O
run() { echo "== $1" | tee -a $O; shift; python tools/ab_pairs.py "$@" 2>&1 | tee -a $O; }
run "f64 m=16: shipped 8x2 | 4x4 swizzle 4/8, 2/8, 0/8 | 8x2 swizzle 2/8" 1000:16:smsqfa:f64 - $V/r4b_f64_16_4x4_s4.so $V/r4b_f64_16_4x4_s2.so $V/r4b_f64_16_4x4_s0.so $V/r4b_f64_16_8x2_s2.so -
run "f64 m=12: shipped 4x3 (8/8) | 4/8 | 0/8" 1000:12:smsqfa:f64 - $V/r4b_f64_12_4x3_s4.so $V/r4b_f64_12_4x3_s0.so
run "f64 m=17: shipped 8x3 | 4x5 two waves, swizzle 4/8 | 0/8" 1000:16:sqfa:f64 - $V/r4b_f64_17_4x5_s4.so $V/r4b_f64_17_4x5_s0.so
