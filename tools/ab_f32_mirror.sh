#!/bin/bash
# Round 4 A/B: partner lane ^ 7 of the 8-lane groups through DPP row_half_mirror instead of ds_swizzle (SQFA_DPP_S_MASK bit 7)
V=variants/build
O=gpurun_out/r4/f32_mirror.txt
mkdir -p gpurun_out/r4
: > $O
run() { echo "== $1" | tee -a $O; shift; python tools/ab_pairs.py "$@" 2>&1 | tee -a $O; }
run "f32 m=32: shipped | partner 7 by DPP" 1000:32:smsqfa - $V/r4d_f32_32_m7.so -
run "f32 m=33: shipped | partner 7 by DPP" 1000:32:sqfa - $V/r4d_f32_33_m7.so
run "f32 m=24: shipped | partner 7 by DPP" 1000:24:smsqfa - $V/r4d_f32_24_m7.so
