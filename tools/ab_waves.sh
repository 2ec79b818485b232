#!/bin/bash
# Round 4 A/B: waves per workgroup / tile width of the 4-lane float32 rows (shipped: 16 x 8 tiles, 4 waves = two B classes per wave)
V=variants/build
O=gpurun_out/r4/waves.txt
mkdir -p gpurun_out/r4
: > $O
run() { echo "== $1" | tee -a $O; shift; python tools/ab_pairs.py "$@" 2>&1 | tee -a $O; }
run "f32 m=16: shipped (TJ 8, 4 waves) | 8 waves | 2 waves | TJ 16, 8 waves" 1000:16:smsqfa - variants/build/r4g_f32_16_w8.so variants/build/r4g_f32_16_w2.so variants/build/r4g_f32_16_t16.so -
run "f32 m=17: shipped | 8 waves" 1000:16:sqfa - variants/build/r4g_f32_17_w8.so
