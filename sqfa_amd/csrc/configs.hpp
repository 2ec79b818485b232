// configs.hpp -- the (dtype, padded size) -> lane-group geometry table.
// X(T, MR, G, CPL, TJ, WAVES): matrices up to MR x MR are handled by G lanes per pair with
// CPL column slots per lane; a workgroup of WAVES wave64 owns a tile of (64/G) x TJ classes.
// A problem of size m runs on the smallest MR >= m (identity padding is exact: padded
// generalized eigenvalues are 1).  Keep in sync with the CONFIGS list in the Makefile.
// Every row is its own macro so that a development build can override one geometry from the command line
// (tools/build_variant.sh: -D'SQFA_ROW_F32_32(X)=X(float,32,16,2,4,1)').
#pragma once

#ifndef SQFA_ROW_F32_4
#define SQFA_ROW_F32_4(X) X(float, 4, 1, 4, 8, 4)
#endif
#ifndef SQFA_ROW_F32_8
#define SQFA_ROW_F32_8(X) X(float, 8, 1, 8, 8, 4)
#endif
#ifndef SQFA_ROW_F32_12
#define SQFA_ROW_F32_12(X) X(float, 12, 4, 3, 8, 4)
#endif
#ifndef SQFA_ROW_F32_16
#define SQFA_ROW_F32_16(X) X(float, 16, 4, 4, 8, 4)
#endif
#ifndef SQFA_ROW_F32_17
#define SQFA_ROW_F32_17(X) X(float, 17, 4, 5, 8, 4)
#endif
#ifndef SQFA_ROW_F32_20
#define SQFA_ROW_F32_20(X) X(float, 20, 4, 5, 8, 4)
#endif
#ifndef SQFA_ROW_F32_24
#define SQFA_ROW_F32_24(X) X(float, 24, 8, 3, 8, 4)
#endif
#ifndef SQFA_ROW_F32_32
// round 4: two waves per workgroup (with the slot-exchange sweeps: 5.70 -> 5.51 ms at C=1000; four: 5.67, 8-wide tiles: 5.67-5.83)
#define SQFA_ROW_F32_32(X) X(float, 32, 8, 4, 4, 2)
#endif
#ifndef SQFA_ROW_F32_33
#define SQFA_ROW_F32_33(X) X(float, 33, 8, 5, 4, 2)  // two waves: 8.29 -> 8.21 ms (four: 8.67)
#endif
#ifndef SQFA_ROW_F32_48
#define SQFA_ROW_F32_48(X) X(float, 48, 16, 3, 4, 2)  // round 4: two waves per workgroup, 2.615 -> 2.545 ms at C=300
#endif
#ifndef SQFA_ROW_F32_64
#define SQFA_ROW_F32_64(X) X(float, 64, 32, 2, 4, 2)
#endif

// Small-launch rows (round 3): the same padded sizes with MORE lanes per pair.  A launch with few pairs does not fill the chip
// -- c2 (C=100: 4 950 pairs) is 78 waves of the one-lane-per-pair m=8 row on 1 024 SIMDs -- so its duration is one wave's
// latency, and fewer columns per lane means fewer dependent instructions per sweep.  Measured (pair kernel, us):
//   C=100:  m=8 27.6 -> 14.9 (4 lanes x 2 slots; 2 x 4: 20.6)   m=12 35.9 -> 27.8 (16 x 1; 8 x 2: 30.2)
//           m=16 60.8 -> 35.3 (16 x 1; 8 x 2: 39.3)              m=17 91.0 -> 67.1 (8 x 3; 16 x 2: 71.9)
//   C=300:  m=8 30.3 -> 23.9, but m=16 113 -> 147, m=17 176 -> 217;   C=1000: m=8 154 -> 176
// so a problem runs on these rows only while its pair count (per shard) stays below small_launch_max_pairs(MR).
// (m=20 and m=32 were added from the same kind of measurement, see small_launch_max_pairs.)
#ifndef SQFA_CONFIGS_F32_SMALL
#define SQFA_CONFIGS_F32_SMALL(X) \
  X(float, 8, 4, 2, 8, 4)   \
  X(float, 12, 16, 1, 8, 4) \
  X(float, 16, 16, 1, 8, 4) \
  X(float, 17, 8, 3, 8, 4)  \
  X(float, 20, 8, 3, 8, 4)  \
  X(float, 32, 32, 1, 4, 1)
#endif
#ifndef SQFA_CONFIGS_F64_SMALL
#define SQFA_CONFIGS_F64_SMALL(X) \
  X(double, 8, 2, 4, 8, 4)   \
  X(double, 16, 8, 2, 8, 4)  \
  X(double, 17, 8, 3, 8, 4)
#endif
// crossovers measured with tools/time_small_launch.py (profiles/r3_small_launch.txt): m <= 8 the small row still wins at 180 k
// pairs (64.5 vs 73.4 us) and loses at 500 k (176 vs 154); m=9...12 at ~15 k pairs; m=16 / 17 at ~20 k
// m=20 (8 lanes x 3 slots) 118 -> 86 us at C=50, 121 -> 96 at C=100, 124 -> 110 at C=130, 156 -> 162 at C=200; m=32 (32 lanes x 1
// slot) 235 -> 115, 249 -> 151, 293 -> 270, 489 -> 539; other candidates lost or gained < 15 % (m=4 4 x 1, m=24 16 x 2, m=33 16 x 3 / 32 x 2): no row
constexpr long small_launch_max_pairs(int MR) { return MR <= 8 ? 250000 : (MR <= 12 ? 14000 : (MR <= 17 ? 20000 : 12000)); }
// float64 crossovers (round 4, tools/time_small_launch.py f64, profiles/r4_small_launch_f64.txt; regular / small-launch row, us):
//   m=8  (1 x 8 / 2 x 4):  C=100 39.5 / 30.5, C=200 45.1 / 34.3, C=300 48.5 / 52.0, C=450 106 / 89, C=600 140 / 128, C=800 210 / 215
//   m=16 (4 x 4 / 8 x 2):  C=100 101 / 74, C=200 160 / 139, C=300 236 / 265, C=450 510 / 478, C=600 824 / 822, C=800 1314 / 1454
//   m=17 (4 x 5 / 8 x 3):  C=100 133 / 103, C=150 141 / 159, C=200 257 / 225, C=300 400 / 415, C=450 830 / 794, C=600 1393 / 1428
// (wave-count quantisation makes the curves cross more than once: the thresholds sit where the small row stops winning clearly)
#ifndef SQFA_SMALL_MAX_PAIRS_F64_8
#define SQFA_SMALL_MAX_PAIRS_F64_8 250000
#endif
#ifndef SQFA_SMALL_MAX_PAIRS_F64_16
#define SQFA_SMALL_MAX_PAIRS_F64_16 30000
#endif
constexpr long small_launch_max_pairs_f64(int MR) { return MR <= 8 ? SQFA_SMALL_MAX_PAIRS_F64_8 : SQFA_SMALL_MAX_PAIRS_F64_16; }

#define SQFA_CONFIGS_F32(X) \
  SQFA_ROW_F32_4(X) \
  SQFA_ROW_F32_8(X) \
  SQFA_ROW_F32_12(X) \
  SQFA_ROW_F32_16(X) \
  SQFA_ROW_F32_17(X) \
  SQFA_ROW_F32_20(X) \
  SQFA_ROW_F32_24(X) \
  SQFA_ROW_F32_32(X) \
  SQFA_ROW_F32_33(X) \
  SQFA_ROW_F32_48(X) \
  SQFA_ROW_F32_64(X)

#ifndef SQFA_ROW_F64_4
#define SQFA_ROW_F64_4(X) X(double, 4, 1, 4, 8, 4)
#endif
// round 4 (profiles/r4_pairs_fewer_lanes.txt): one lane per pair -- no cross-lane traffic at all, 240 VGPRs, two waves per
// SIMD -- 0.322 -> 0.268 ms at C=1000; the previous 2 lanes x 4 slots stay as the small-launch row below
#ifndef SQFA_ROW_F64_8
#define SQFA_ROW_F64_8(X) X(double, 8, 1, 8, 8, 4)
#endif
#ifndef SQFA_ROW_F64_12
#define SQFA_ROW_F64_12(X) X(double, 12, 4, 3, 8, 4)
#endif
// round 4: 4-lane groups with every partner row through DPP (no ds_swizzle: see swizzled_rows_of_8) -- m=16 2.13 -> 1.92 ms,
// m=17 3.67 -> 3.53 ms at C=1000; the 8-lane rows of round 3 stay as small-launch rows
#ifndef SQFA_ROW_F64_16
#define SQFA_ROW_F64_16(X) X(double, 16, 4, 4, 8, 4)
#endif
#ifndef SQFA_ROW_F64_17
#define SQFA_ROW_F64_17(X) X(double, 17, 4, 5, 8, 4)
#endif
#ifndef SQFA_ROW_F64_20
#define SQFA_ROW_F64_20(X) X(double, 20, 8, 3, 8, 4)
#endif
#ifndef SQFA_ROW_F64_64
#define SQFA_ROW_F64_64(X) X(double, 64, 64, 1, 4, 2)
#endif

#define SQFA_CONFIGS_F64(X) \
  SQFA_ROW_F64_4(X) \
  SQFA_ROW_F64_8(X) \
  SQFA_ROW_F64_12(X) \
  SQFA_ROW_F64_16(X) \
  SQFA_ROW_F64_17(X) \
  SQFA_ROW_F64_20(X) \
  SQFA_ROW_F64_64(X)

// 2-D lane layouts (pair_kernel_2d.hpp): X(T, MR, GC, CPL, TJ, WAVES, RS) -- GC column lanes x 2 row lanes per pair, CPL
// column slots x ceil(MR/2) rows per lane, row partner lane ^ RS.  Keep in sync with CONFIGS2D in the Makefile.
// Used where the whole-column layout is down to ONE wave per SIMD (measured round 3, C=1000, ms per evaluation,
// whole columns -> 2-D; profiles/r3_pairs_2d.txt):
//   float32 m=40: 32.8 -> 19.4            float64 m=24: 13.1 -> 9.55   m=32: 36.5 -> 24.9   m=33: 52.1 -> 39.3
//                                         float64 m=48 (C=300): 30.6 -> 10.9   [m=64 (C=300): 48.7 -> 34.9, not shipped: its
//                                         unrolled 32-lane tournament takes 3.4 minutes to compile]
// and NOT where the whole-column layout already holds two or more waves per SIMD -- everything per rotation is executed
// by both row lanes (+20-27 % instructions), which more waves do not pay back:
//   float32 m=24: 3.64 -> 4.63, m=32: 8.55-8.82 -> 9.80-10.2 (tiles 4 x 8: 10.5; compiled for three waves: 9.99),
//   m=33: 12.1 -> 13.4;  float64 m=17: 3.94 -> 5.84, m=20: 6.08 -> 7.76.
#ifndef SQFA_CONFIGS2D_F32
#define SQFA_CONFIGS2D_F32(X) \
  X(float, 40, 8, 5, 4, 1, 16)
#endif
#ifndef SQFA_CONFIGS2D_F64
#define SQFA_CONFIGS2D_F64(X) \
  X(double, 24, 8, 3, 4, 1, 16)  \
  X(double, 32, 16, 2, 8, 1, 32) \
  X(double, 33, 16, 3, 8, 1, 32) \
  X(double, 48, 16, 3, 4, 1, 32)
#endif
