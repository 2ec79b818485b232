#!/bin/bash
# Round 4 A/B: fewer lanes per pair.  Variants built by tools/build_variant.sh into variants/build.
set -e
V=variants/build
O=gpurun_out/r4/fewer_lanes.txt
mkdir -p gpurun_out/r4
: > $O
run() { echo "== $1" | tee -a $O; shift; python tools/ab_pairs.py "$@" 2>&1 | tee -a $O; }
run "f32 m=16 (c3): shipped 4x4 vs 2x8 unpaired / paired" 1000:16:smsqfa - $V/r4_f32_16_2x8.so $V/r4_f32_16_2x8p.so -
run "f32 m=17 (c3-SQFA): shipped 4x5 vs 2x9 unpaired / paired" 1000:16:sqfa - $V/r4_f32_17_2x9.so $V/r4_f32_17_2x9p.so -
run "f32 m=12: shipped 4x3 vs 2x6 unpaired / paired" 1000:12:smsqfa - $V/r4_f32_12_2x6.so $V/r4_f32_12_2x6p.so
run "f32 m=24: shipped 8x3 vs 4x6" 1000:24:smsqfa - $V/r4_f32_24_4x6.so
run "f64 m=16: shipped 8x2 vs 4x4" 1000:16:smsqfa:f64 - $V/r4_f64_16_4x4.so
run "f64 m=17: shipped 8x3 vs 4x5 (two waves forced) unpaired / paired" 1000:16:sqfa:f64 - $V/r4_f64_17_4x5.so $V/r4_f64_17_4x5p.so
run "f64 m=12: shipped 4x3 vs 2x6" 1000:12:smsqfa:f64 - $V/r4_f64_12_2x6.so
run "f64 m=8: shipped 2x4 vs 1x8" 1000:8:smsqfa:f64 - $V/r4_f64_8_1x8.so
