#!/bin/bash
# Round 4 A/B: re-tune of the DPP / ds_swizzle split of the float32 rows (SQFA_SWZ_ROWS_OF_8: shipped 4 of 8 for 4-lane groups, 1 of 8 for 8-lane groups)
V=variants/build
O=gpurun_out/r4/f32_swizzle.txt
mkdir -p gpurun_out/r4
: > $O
run() { echo "== $1" | tee -a $O; shift; python tools/ab_pairs.py "$@" 2>&1 | tee -a $O; }
run "f32 m=16: shipped (4/8) | 0/8 | 2/8 | 6/8" 1000:16:smsqfa - $V/r4c_f32_16_s0.so $V/r4c_f32_16_s2.so $V/r4c_f32_16_s6.so -
run "f32 m=17: shipped (4/8) | 0/8 | 2/8 | 6/8" 1000:16:sqfa - $V/r4c_f32_17_s0.so $V/r4c_f32_17_s2.so $V/r4c_f32_17_s6.so
run "f32 m=32: shipped (1/8) | 0/8 | 2/8" 1000:32:smsqfa - $V/r4c_f32_32_s0.so $V/r4c_f32_32_s2.so
