// pair_kernel.hpp -- the pairwise affine-invariant distance tile kernel for gfx950.
//
// One workgroup (WAVES wave64) owns a TI x TJ tile of (A class, B class) pairs.  Inside a
// wave, G adjacent lanes co-operate on one pair ("lane group"); a wave therefore works
// on 64/G pairs at once, all sharing the same B class j (its L_j^-1 is staged once per
// wave-round in LDS and read as broadcasts), with TI = 64/G different A classes i.
//
// Per pair (A = S_i, B = S_j = L_j L_j^T):
//   1. X = L_j^-1 F_i                  (F_i = L_i, or L_i V_i with orthogonal columns after the class factor
//                                        pass; M = X X^T = L_j^-1 A L_j^-T either way)
//   2. one-sided (Hestenes) Jacobi on the COLUMNS of X, held in registers: each lane owns
//      CPL column slots of all MR rows; rotations between columns of one lane are local.
//      Since round 4 the columns TRAVEL between the lanes of a group (exchange_slots: slot s is
//      exchanged with the same slot of lane ^ mask, masks from GF(2^g) arithmetic) so that every pair
//      of columns from different slots is local once per sweep; same-slot pairs become local through
//      an LDS transposition of the (lane, slot) blocks (transpose_slots: 4 x 4 groups) or keep the
//      two-owner steps of the original XOR tournament (lane^s, slot^t: cross_step2; also the whole
//      sweep of the few geometries the exchange scheme does not cover).  Cross-lane moves go
//      through DPP (1.63 VALU issue slots per move on gfx950) or ds_swizzle (the LDS crossbar, off
//      the VALU), in a ratio tuned per group size.  Columns carry a squared scale (x = sqrt(D) x^),
//      so a rotation costs one fma per element and its parameters one v_rsq + one v_rcp
//      (rot_local for one-owner rotations, rot_scaled for two owners).
//      On exit X J = Y with orthogonal columns y_k = sigma_k v_k: lambda_k = |y_k|^2 are
//      the generalized eigenvalues of (A,B), v_k the eigenvectors of M.
//   3. d2 = scale * sum log(lambda)^2, D = sqrt(d2+eps) | d2
//   4. backward (closed form, SURVEY.md 3.4 / oracle/closed_form.py): with
//      u~_k = L_j^-T y_k:  dL/dA += sum_k (g_k/lambda_k) u~ u~^T,  dL/dB -= sum_k g_k u~ u~^T,
//      g_k = w * dD/dd2 * scale * 2 log(lambda_k)/lambda_k.
//      Lower triangles only, reduced with transposing tree reductions: A-side sums over the
//      lane group go to a per-wave private LDS accumulator (deterministic), B-side sums
//      over the whole wave (all its pairs share j) go straight to the slab in HBM.  Tiles
//      flush the A side to the slab; finalize_kernel reduces slabs per class in a fixed order.
//
// Cost model behind the choices (tools/ubench/*.hip, MI355X): plain VALU op = 1 issue slot
// (~2.5 cycles/wave-instr at >= 2 waves/SIMD, 5.4 for a lone wave), DPP op = 1.63 slots,
// v_rsq/v_rcp/v_sqrt = 3.2 slots, v_pk_fma_f32 = 1.7 slots for 2 flops, v_permlane16/32_swap 3.2,
// ds_swizzle ~2.3 cycles per CU (LDS pipe, off the VALU), ds_bpermute ~6; LDS float atomics: unusable.
//
// Replaces: src/sqfa/linalg.py:19-70,144-162, src/sqfa/distances.py:46-89,177-237,
// src/sqfa/_optim.py:16-30,88-96 of the reference and the autograd backward of that chain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sqfa {

struct PairParams {
  const void* LT;     // [nA][MR*MR]  LT[c][k] = L_A[k][c]  (columns of L contiguous), identity padded
  const void* Linv;   // inverse Cholesky factor of each B class, identity padded: [nB][MR*MR] row-major, or
                      // [nB][MR(MR+1)/2] packed lower triangle (row r at r(r+1)/2) for MR >= 32 (PairCfg::PACK_LINV)
  const void* W;      // optional (nA,nB) pair weights, or nullptr
  const void* EW;     // optional (nA,nB,m) per-eigenvalue weights in eig_out's (unsorted column) order, or nullptr:
                      // the gradient is then that of  sum_ijk EW_ijk lambda_k(A_i,B_j)  (backward of generalized_eigenvalues)
  void* slab_grad;    // [tiles of this shard][TI+tj][TRI]  lower triangles: TI A-side rows, then tj B-side rows per tile;
                      // tiles are numbered like the compact launch grid (row-major over the owned tiles)
  void* slab_loss;    // [tiles of this shard]
  int* slab_flag;     // [tiles of this shard][2]  {NaN count, inf count}
  int* row_start;     // [nbi + 1] number of owned tiles before block-row bi (written by the Cholesky prologue, read by K2)
  void* dist_out;     // (nA,nB) or nullptr
  void* eig_out;      // (nA,nB,m) or nullptr
  unsigned long long* sweep_counter;  // optional debug counter {sum of sweeps, wave rounds}
  int nA, nB, m;
  int self_mode, sqrt_mode, want_grad;
  int shard_index, shard_count;
  int nbi, nbj;
  int tj;  // B classes per tile in this launch (<= the configuration's TJ, a multiple of its wave count)
  int factor_mode;  // class factor pass: 0 by pair count (PairCfg::FACTOR_MIN_PAIRS), 1 always, -1 never
  const double* mean_linv;  // Lbar^-1 of the mean A class (MR x MR doubles, identity padded) for the mean-metric factor pass, or nullptr
  double scale, eps, uniform_weight;
  float scale_f, eps_f, uniform_weight_f;  // the same three, pre-rounded for the float32 kernels (stay in SGPRs)
};

template <typename T> __device__ __forceinline__ T param_scale(const PairParams& p) {
  if constexpr (sizeof(T) == 4) return p.scale_f; else return p.scale;
}
template <typename T> __device__ __forceinline__ T param_eps(const PairParams& p) {
  if constexpr (sizeof(T) == 4) return p.eps_f; else return p.eps;
}
template <typename T> __device__ __forceinline__ T param_uniform_weight(const PairParams& p) {
  if constexpr (sizeof(T) == 4) return p.uniform_weight_f; else return p.uniform_weight;
}

// ---------------------------------------------------------------------------------------
// Which tiles exist.  The launch grid is COMPACT: workgroup w of shard r is the w-th tile, in
// row-major order, among the tiles (bi, bj) that (a) hold at least one pair of the job (self
// mode: some i > j) and (b) belong to the shard, (bi + bj) % shard_count == shard_index.
// Launching the full nbi x nbj rectangle and returning early from foreign tiles would be
// simpler, but workgroups are dealt to the 8 XCDs round-robin by linear id and the stripe
// pattern (bi + bj) % N is periodic in exactly that id: with N = 2, 4, 8 shards the tiles of a
// shard all landed on 4, 2, 2 of the 8 XCDs and a shard ran no faster than the whole job.
__host__ __device__ inline int tiles_in_row(int bi, int nbj, int TI, int TJ, int self_mode) {
  if (!self_mode) return nbj;
  const int q = bi * TI + TI - 2;  // tiles with j0 <= i0 + TI - 2 hold a pair i > j
  if (q < 0) return 0;
  const int len = q / TJ + 1;
  return len < nbj ? len : nbj;
}
// tiles of block-row bi owned by the shard: bj = first, first + N, ... (count returned)
__host__ __device__ inline int shard_tiles_in_row(int bi, int len, int shard_index, int shard_count, int* first) {
  const int o = ((shard_index - bi) % shard_count + shard_count) % shard_count;
  *first = o;
  return len > o ? (len - 1 - o) / shard_count + 1 : 0;
}

__host__ __device__ constexpr int ilog2(int v);
__host__ __device__ constexpr int pow2ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
__host__ __device__ constexpr int tri_index(int r, int c) { return r * (r + 1) / 2 + c; }  // r >= c

// ---------------------------------------------------------------------------------------
// scalar helpers
template <typename T> struct Real;
template <> struct Real<float> {
  static constexpr float kEps = 1.1920929e-07f;
#ifndef SQFA_EARLY2_F32
#define SQFA_EARLY2_F32 3.0e-6f  // 1e-7 until round 3; measured against the float64 kernels: 1e-7 ... 3e-6 identical to three digits, 1e-5 first visible
#endif
  // a sweep in which every cos^2 between columns stays below this is the last one.  3e-6 where the probes against the float64
  // kernels show no change at all (column lengths >= 12); the short columns of m <= 8 keep 1e-6: at C=10, m=5 the gradient
  // error moved from 6.0e-7 to 9.1e-7 with 3e-6 (nothing at 1e-6), and those kernels are not where the time goes
#ifndef SQFA_EARLY2_F32_SHORT
#define SQFA_EARLY2_F32_SHORT 1.0e-6f
#endif
  template <int MR> static constexpr float early2() { return MR >= 12 ? SQFA_EARLY2_F32 : SQFA_EARLY2_F32_SHORT; }
#ifndef SQFA_RENORM_LOG2
#define SQFA_RENORM_LOG2 24
#endif
  static constexpr float kScaleHi = (float)(1ull << SQFA_RENORM_LOG2), kScaleLo = 1.0f / kScaleHi;  // see the sweep loop
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static __device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
  static __device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ float abs_(float x) { return __builtin_fabsf(x); }
  static __device__ __forceinline__ float copysign_(float a, float b) { return __builtin_copysignf(a, b); }
  static __device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
  static __device__ __forceinline__ float log_(float x) { return logf(x); }
  static __device__ __forceinline__ bool finite(float x) { return __builtin_isfinite(x); }
};
template <> struct Real<double> {
  static constexpr double kEps = 2.220446049250313e-16;
#ifndef SQFA_EARLY2_F64
#define SQFA_EARLY2_F64 1.0e-15
#endif
  template <int MR> static constexpr double early2() { return SQFA_EARLY2_F64; }
  static constexpr double kScaleHi = (double)(1ull << SQFA_RENORM_LOG2), kScaleLo = 1.0 / kScaleHi;
  // v_rcp_f64 seed + one third-order step: 1/x = y (1 + e + e^2 + O(e^3)), e = 1 - x y
  static __device__ __forceinline__ double rcp(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, __builtin_fma(e, e, e), y);
  }
  // v_rsq_f64 seed (>= 24 good bits) + one third-order correction step: with e = 1 - x y^2,
  // x^-1/2 = y (1 - e)^-1/2 = y (1 + e/2 + 3e^2/8 + O(e^3)) -- full double precision in six
  // instructions instead of the ~45 of the IEEE sqrt + divide sequences.  Only called with
  // x > 0 whenever the result is used (rot_scaled selects it away otherwise).
  static __device__ __forceinline__ double rsq(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
  }
  static __device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
  static __device__ __forceinline__ double abs_(double x) { return __builtin_fabs(x); }
  static __device__ __forceinline__ double copysign_(double a, double b) { return __builtin_copysign(a, b); }
  static __device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
  static __device__ __forceinline__ double log_(double x) { return log(x); }
  static __device__ __forceinline__ bool finite(double x) { return __builtin_isfinite(x); }
};

// ---------------------------------------------------------------------------------------
// cross-lane movement.  DPP controls: quad_perm [1,0,3,2]=0xB1 (xor 1), [2,3,0,1]=0x4E (xor 2),
// [3,2,1,0]=0x1B (xor 3), row_half_mirror=0x141, row_mirror=0x140.
template <int CTRL> __device__ __forceinline__ int dpp_mov_i(int v) {
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, dpp_mov_i<CTRL>(__builtin_bit_cast(int, v)));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(dpp_mov_i<CTRL>(hi), dpp_mov_i<CTRL>(lo));
}

template <int S> __device__ __forceinline__ float swizzle_xor(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (S << 10) | 0x1F));
}
template <int S> __device__ __forceinline__ double swizzle_xor(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_ds_swizzle(hi, (S << 10) | 0x1F), __builtin_amdgcn_ds_swizzle(lo, (S << 10) | 0x1F));
}

// value held by lane (lane ^ S).  S > 0: compile-time partner -- DPP quad_perm for S <= 3,
// ds_swizzle (LDS crossbar, bit-mask mode) for 4 <= S < 32; S == 0: runtime partner `s`
// through ds_bpermute (3x slower than ds_swizzle, tools/ubench/swizzle_rate.hip).
// bit S set: partner lane^S (S = 7, 8, 15) through a DPP row move instead of the crossbar.  Round 4: bit 7 (row_half_mirror) on --
// the 8-lane groups reach partners 4..7 through ds_swizzle only, and taking one of the four off the crossbar measured m=24
// 2.937 -> 2.890 ms, m=32 7.66-7.69 -> 7.53, m=33 10.34 -> 10.19 (C=1000, profiles/r4_pairs_fewer_lanes.txt, last section)
// Bits 8 and 15 (row_ror:8, row_mirror: the 16- and 32-lane groups) on as well: float64 m=32 / 33 / 48 (2-D rows) -1.6 / -1.1 /
// -2.7 %, float32 m=48 / 64 -2.4 / -1.0 %, nothing moves at m <= 33 float32 (same file, "bits 7, 8, 15").
#ifndef SQFA_DPP_S_MASK
#define SQFA_DPP_S_MASK 0x8180
#endif
template <int S, typename T> __device__ __forceinline__ T lane_xor(T v, int s) {
  if constexpr (S == 1) return dpp_mov<0xB1>(v);
  else if constexpr (S == 2) return dpp_mov<0x4E>(v);
  else if constexpr (S == 3) return dpp_mov<0x1B>(v);
  else if constexpr (S == 7 && ((SQFA_DPP_S_MASK >> 7) & 1)) return dpp_mov<0x141>(v);   // row_half_mirror: l -> l ^ 7
  else if constexpr (S == 8 && ((SQFA_DPP_S_MASK >> 8) & 1)) return dpp_mov<0x128>(v);   // row_ror:8: l -> l ^ 8
  else if constexpr (S == 15 && ((SQFA_DPP_S_MASK >> 15) & 1)) return dpp_mov<0x140>(v);  // row_mirror: l -> l ^ 15
  else if constexpr (S == 0) return __shfl_xor(v, s, 64);
  else if constexpr (S < 32) return swizzle_xor<S>(v);
  else return __shfl_xor(v, S, 64);
}

// Same, for the r-th element of a column: DPP moves cost two VALU issue slots on gfx950
// (tools/ubench/valu_rate.hip), ds_swizzle runs on the otherwise idle LDS crossbar (~2.3
// cycles per wave-op per CU).  Of every 8 rows, SWZ go through the crossbar so that both
// pipes share the cross-lane traffic.  Measured optimum (tools/time_variants_any.py): all
// rows for 4-lane groups in round 1 (half of them once the steps were paired, see below); for 8-lane groups, whose partners 4..7 can only be reached through
// the crossbar anyway, 1 of 8 in float32 and none in float64 (two 32-bit moves per element).
#ifndef SQFA_SWZ_ROWS_OF_8
#define SQFA_SWZ_ROWS_OF_8 -1  // -1: by group size and element type
#endif
template <typename T, int G, int MR, bool EXCHANGE = false> constexpr int swizzled_rows_of_8() {
  if (SQFA_SWZ_ROWS_OF_8 >= 0) return SQFA_SWZ_ROWS_OF_8;
  // slot-exchange sweeps (round 4), float32 4-lane groups: ALL rows through the crossbar again -- half as many cross-lane moves
  // as the tournament issued, and every DPP move is 1.63 issue slots of a kernel that is bound by them (C=1000: m=16
  // 0.712 -> 0.692 ms with 8 of 8, 0.701 with 6, 0.732 with 2, 0.790 with none; m=17 1.142 -> 1.134; float64 keeps none:
  // m=16 1.573 / 1.589 / 1.617 / 1.637 ms with 0 / 2 / 6 / 8; 8-lane groups +-0.5 %: unchanged)
  if (EXCHANGE && G <= 4 && sizeof(T) == 4) return 8;
  // 4-lane groups: with two steps in flight and the fetch bursts at raised priority (float32, MR >= 16) the
  // crossbar is the tighter pipe again and half of the rows go back to DPP (round 2, alternating runs on one
  // box: m=16 1.098 -> 1.066 ms, m=17 1.647 -> 1.588); the unpaired m=12 keeps all rows on the crossbar
  // (0.574 vs 0.586 ms)
  // float64 4-lane groups: NO rows through the crossbar (round 4; a 64-bit element is two ds_swizzle operations and the
  // crossbar became the tighter pipe: m=16 4 x 4 2.09 ms with all rows swizzled, 1.98 with half, 1.92 with none; m=12
  // 0.959 -> 0.886; m=17 4 x 5 3.79 -> 3.53 -- profiles/r4_pairs_fewer_lanes.txt)
  if (G <= 4) return sizeof(T) == 8 ? 0 : (MR >= 16 ? 4 : 8);
  return sizeof(T) == 4 ? 1 : 0;
}
template <int S, int SWZ, typename T> __device__ __forceinline__ T lane_xor_row(T v, int s, int r) {
  if constexpr (S >= 1 && S <= 3 && SWZ > 0) {
    if (r % 8 < SWZ) return swizzle_xor<S>(v);
  }
  return lane_xor<S>(v, s);
}

// gfx950 row swaps: v_permlane{16,32}_swap exchange the odd H-lane rows of the first operand
// with the even H-lane rows of the second.  For a value `a` wanted by the lanes with bit H
// clear and `b` wanted by the lanes with bit H set, (a', b') = swap(a, b) gives
//   a' + b' = a[lane] + a[lane ^ H]  where bit H is clear,  b[lane] + b[lane ^ H]  where it is set
// -- one exchange step of a transposing reduction in two instructions, no selects.
template <int H> __device__ __forceinline__ int row_swap_sum_i(int a, int b, int& other) {
  static_assert(H == 16 || H == 32, "row swaps exist for 16 and 32 lanes");
  if constexpr (H == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    other = r[1];
    return r[0];
  } else {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    other = r[1];
    return r[0];
  }
}
template <int H> __device__ __forceinline__ float row_swap_sum(float a, float b) {
  int o;
  const int k = row_swap_sum_i<H>(__builtin_bit_cast(int, a), __builtin_bit_cast(int, b), o);
  return __builtin_bit_cast(float, k) + __builtin_bit_cast(float, o);
}
template <int H> __device__ __forceinline__ double row_swap_sum(double a, double b) {
  int ohi, olo;
  const int khi = row_swap_sum_i<H>(__double2hiint(a), __double2hiint(b), ohi);
  const int klo = row_swap_sum_i<H>(__double2loint(a), __double2loint(b), olo);
  return __hiloint2double(khi, klo) + __hiloint2double(ohi, olo);
}

// sum over the G lanes of a lane group (every lane gets the total)
template <int G, typename T> __device__ __forceinline__ T group_sum(T v) {
  if constexpr (G >= 2) v += dpp_mov<0xB1>(v);
  if constexpr (G >= 4) v += dpp_mov<0x4E>(v);
  if constexpr (G >= 8) v += dpp_mov<0x141>(v);   // quads are uniform now: half mirror == xor 4
  if constexpr (G >= 16) v += dpp_mov<0x140>(v);  // row mirror == xor 8
  if constexpr (G >= 32) v = row_swap_sum<16>(v, v);
  if constexpr (G >= 64) v = row_swap_sum<32>(v, v);
  return v;
}
// sum over all 64 lanes, fixed association order (deterministic)
template <typename T> __device__ __forceinline__ T wave_sum(T v) { return group_sum<64>(v); }
// tell the compiler that a value is the same in every lane (moves it to SGPRs)
__device__ __forceinline__ float wave_uniform(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ double wave_uniform(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                          __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// ---------------------------------------------------------------------------------------
// Scaled ("fast") Jacobi rotations in squared quantities.
//
// Column slot c holds x^ with  x_true = sqrt(D[c]) x^ ; nrm[c] = |x_true|^2 is tracked
// separately.  For my column x (norm^2 no, scale^2 Dx) against a partner column y (nr, Dy)
// with scaled inner product gh = <x^, y^>  (gam = gh sqrt(Dx Dy),  g2 = gam^2 = gh^2 Dx Dy):
//   dh = (nr - no)/2,  h = sqrt(dh^2 + g2),  u = cos^2 th = (1 + |dh|/h)/2,
//   tan th = k gam  with  k = sign(dh) / (2 h u)
// and the rotation  x' = cos (x - tan y),  y' = cos (y + tan x)  becomes, per element, ONE fma:
//   x^' = x^ - (k gh Dy) y^,   y^' = y^ + (k gh Dx) x^,   Dx' = u Dx,  Dy' = u Dy,
//   no' = no - k g2,  nr' = nr + k g2
// -- no square root of a scale is ever needed inside the sweeps (one v_rsq and one v_rcp per
// rotation), and both owners of a cross-lane rotation work from the same Dx, Dy, gh, so the
// rotation they jointly apply to the true columns is orthogonal to rounding.  D changes by
// factors in [1/2, 2], a few hundred times at most: no range problem.
//
// The two owners must choose opposite signs; sign(dh) does that by itself except when the
// norms are EXACTLY equal (dh = +0 on both sides): `tie` (+1 on one owner, -1 on the other)
// breaks that tie antisymmetrically.  Outputs are the identity (u = ru = 1, k = 0) when the
// columns are already orthogonal to working precision.
// `MR` (the rows a column has in the calling layout) only selects the stop threshold: Real<T>::early2<MR>()
template <typename T, int MR>
__device__ __forceinline__ void rot_scaled(T no, T nr, T gh, T Dx, T Dy, T tol2, T tie, T& u, T& ru, T& k, T& g2,
                                           bool& big) {
  using R = Real<T>;
#ifdef SQFA_ABL_NO_PARAMS  // development: timing ablation, wrong results
  u = T(1); ru = T(1); k = gh * T(1e-3); g2 = gh * Dx; return;
#endif
  const T ab = no * nr;
  g2 = gh * gh * (Dx * Dy);
  const bool rot = g2 > tol2 * ab;
  big = big || (g2 > R::template early2<MR>() * ab);
  const T dh = T(0.5) * (nr - no);
  const T rh = R::rsq(R::fma_(dh, dh, g2));
  const T uu = R::fma_(T(0.5) * R::abs_(dh), rh, T(0.5));
  const T ruu = R::rcp(uu);
  const T sgn = dh == T(0) ? tie : dh;
  u = rot ? uu : T(1);
  ru = rot ? ruu : T(1);
  k = rot ? R::copysign_(T(0.5) * rh * ruu, sgn) : T(0);
}

// The same rotation for two columns of ONE lane (local_step2), written in d = nr - no and g4 = 4 g2 = (2 gh)^2 Dx Dy so that
// no halving is left in the chain:  h4 = sqrt(d^2 + g4) = 2 h,  w = 1 + |d| / h4 = 2 u,  k/2 = 1 / (h4 w).  Returns u,
// kgh = k gh (the caller's a1 = -kgh Dy, a2 = kgh Dx) and kg4 = (k/2) g4 = 2 k g2 (norm updates -+ kg4 / 2, folded into an
// fma).  One owner: no tie to break (d = +0 for equal norms: the sign bit a lone owner would pick anyway).  There is no
// "rotate at all?" select either: |d| enters as |d| + RotFloor (1e-18 of a norm: invisible), which turns the one case the select
// guarded -- d = 0 and gh = 0 exactly: zero columns, identity-padded columns, two copies of one class -- into |d| / h4 = 1, i.e.
// u = 1 with kgh = kg4 = (large but finite) x 0 = 0: the identity.  Rotations below the rounding noise are harmless and cost what
// a skipped one costs.  24 instructions instead of the 30 of rot_scaled + its caller; at 4 x 4 slots every rotation of a sweep
// is of this kind.
template <typename T> struct RotFloor;
template <> struct RotFloor<float> { static constexpr float v = 1.0e-18f; };     // squared: 1e-36, a normal float
template <> struct RotFloor<double> { static constexpr double v = 1.0e-150; };
template <typename T, int MR>
__device__ __forceinline__ void rot_local(T no, T nr, T gh, T Dx, T Dy, T& u, T& kgh, T& kg4, bool& big) {
  using R = Real<T>;
#ifdef SQFA_ABL_NO_PARAMS  // development: timing ablation, wrong results
  u = T(1); kgh = gh * T(1e-3); kg4 = gh * Dx; return;
#endif
  const T e = gh + gh;
  const T g4 = e * e * (Dx * Dy);
  big = big || (g4 > (T(4) * R::template early2<MR>()) * (no * nr));
  const T d = nr - no;
  const T ad = R::abs_(d) + RotFloor<T>::v;
  const T rh4 = R::rsq(R::fma_(ad, ad, g4));
  const T w = R::fma_(ad, rh4, T(1));
  const T k = R::copysign_(rh4 * R::rcp(w), d);
  u = T(0.5) * w;
  kgh = k * e;
  kg4 = k * g4;
}

// Inner product of two register columns with NA independent partial sums (combined pairwise at the end): one
// accumulator makes the MR fmas one dependency chain, which two waves per SIMD (MR >= 24) do not hide.
#ifndef SQFA_DOT_ACCS
#define SQFA_DOT_ACCS 0  // 0: by size
#endif
template <typename T, int MR> constexpr int dot_accs() {
  if (SQFA_DOT_ACCS > 0) return SQFA_DOT_ACCS;
  return 1;
}
template <typename T, int MR>
__device__ __forceinline__ T dot_cols(const T (&a)[MR], const T (&b)[MR]) {
  using R = Real<T>;
  constexpr int NA = dot_accs<T, MR>();
  T acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = T(0);
#pragma unroll
  for (int r = 0; r < MR; ++r) acc[r % NA] = R::fma_(a[r], b[r], acc[r % NA]);
  if constexpr (NA == 4) return (acc[0] + acc[1]) + (acc[2] + acc[3]);
  else if constexpr (NA == 2) return acc[0] + acc[1];
  else {
    T t = acc[0];
#pragma unroll
    for (int k = 1; k < NA; ++k) t += acc[k];
    return t;
  }
}

// 2-D lane layouts (pair_kernel_2d.hpp): the rows of a column are split over the two lanes (lane, lane ^ RS), RS = 16 or
// 32; an inner product is then the sum of the two lanes' partial sums (one row swap + one add, identical in both lanes:
// a + b and b + a are the same float).  RS = 0: whole columns per lane, nothing to add.
template <int RS, typename T> __device__ __forceinline__ T row_total(T v) {
  if constexpr (RS == 0) return v;
  else return row_swap_sum<RS>(v, v);
}

// Sizes m = G (CPL-1) + 1 (SQFA's K+1: 17 = 4*4+1, 33 = 8*4+1) leave ONE real column in the last slot of
// one lane of the group.  Carried through the tournament it doubles the rounds of every partner
// (pow2ceil(CPL) = 8 instead of 4) for steps in which a single lane pair of the group does useful work.
// With SQFA_Z_VISITS the tournament runs over the first CPL-1 slots only and the lone column ("z") travels
// instead: it visits the lanes of its group one after the other along a Gray-code path of xor moves and
// meets the CPL-1 columns of the lane it is visiting as LOCAL rotations (no partner fetches, no second
// update); the empty last slots of the other lanes hold zero columns, for which rot_scaled returns the
// identity.  Every pair of columns still meets exactly once per sweep.
#ifndef SQFA_Z_VISITS
#define SQFA_Z_VISITS 1
#endif
#ifndef SQFA_Z_PRIO
#define SQFA_Z_PRIO -1  // wave priority while the lone column moves; -1: 3 for float32 (m=33 8.91 -> 8.75 ms, m=17 1.094 -> 1.090), 0 for float64 (2.958 -> 2.973 with 3)
#endif
template <int G, int MR, int CPL> constexpr bool z_visits_cfg() {
  return SQFA_Z_VISITS && G > 1 && MR == G * (CPL - 1) + 1;
}
// slots that take part in the tournaments (LONE_LAST: all but the last)
template <int CPL, bool LONE_LAST> constexpr int tournament_slots() { return (LONE_LAST && SQFA_Z_VISITS) ? CPL - 1 : CPL; }

// one tournament round against the lane group member (lane ^ s): every column slot c of
// mine meets slot (c ^ t) of the partner, t = 0..pow2ceil(CPL)-1.
//
// For t != 0 slots c and cp = c^t are handled together: pair a = (my c, partner's cp) and
// pair b = (my cp, partner's c).  The partner evaluates the same code, so pair b is ITS
// pair a: its inner product and rotation parameters are fetched with a few cross-lane moves
// instead of being recomputed.  Slot c is rotated first (old partner values); slot cp is
// rotated afterwards against the partner's ALREADY ROTATED slot c, with the algebraically
// equivalent form  x' = x/cos - tan y_new, i.e.  x^' = x^ - (k gh Dy_new) y^_new,  Dx' = Dx/u,
// so only one MR-long temporary is live at a time.
template <typename T, int MR, int CPL, int S, int SWZ, bool LONE_LAST, int RS = 0>
__device__ __forceinline__ void cross_round(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], int s, T tol2, bool& big) {
  using R = Real<T>;
  constexpr int CE = tournament_slots<CPL, LONE_LAST>();
  constexpr int TP2 = pow2ceil(CE);
  const int lane_id = (int)(threadIdx.x & 63);
  const T tie = ((lane_id ^ s) > lane_id) ? T(1) : T(-1);
#pragma unroll
  for (int t = 0; t < TP2; ++t) {
#pragma unroll
    for (int c = 0; c < CE; ++c) {
      const int cp = c ^ t;
      if (cp < c || cp >= CE) continue;  // resolved at compile time after unrolling
      // LONE_LAST: the last slot holds a real column in ONE lane of the group only (MR = G(CPL-1)+1,
      // e.g. 17 = 4*4+1), so no two lanes ever have a last-slot pair to rotate
      if (LONE_LAST && c == CPL - 1 && cp == CPL - 1) continue;
      T rv[MR];
#pragma unroll
      for (int r = 0; r < MR; ++r) rv[r] = lane_xor_row<S, SWZ>(x[cp][r], s, r);  // partner's slot cp
      const T gh = row_total<RS>(dot_cols<T, MR>(x[c], rv));
      const T nr1 = lane_xor<S>(nrm[cp], s);
      const T Dp = lane_xor<S>(D[cp], s);
      T u1, ru1, k1, g21;
      rot_scaled<T, MR>(nrm[c], nr1, gh, D[c], Dp, tol2, tie, u1, ru1, k1, g21, big);
      const T kgh = k1 * gh, kg2 = k1 * g21;
      {
        const T a = -(kgh * Dp);
#pragma unroll
        for (int r = 0; r < MR; ++r) x[c][r] = R::fma_(a, rv[r], x[c][r]);
      }
      D[c] *= u1;
      nrm[c] -= kg2;
      if (cp != c) {
        // my slot cp meets the partner's slot c: the partner has just evaluated exactly that
        // rotation from its side (as ITS slot-c rotation); its k is mine with the sign flipped.
        const T kgh2 = lane_xor<S>(kgh, s);
        const T kg22 = lane_xor<S>(kg2, s);
        const T ru2 = lane_xor<S>(ru1, s);
        const T Dpn = lane_xor<S>(D[c], s);  // partner's slot c, already rescaled
#pragma unroll
        for (int r = 0; r < MR; ++r) rv[r] = lane_xor_row<S, SWZ>(x[c][r], s, r);  // partner's slot c, rotated
        const T b = kgh2 * Dpn;
#pragma unroll
        for (int r = 0; r < MR; ++r) x[cp][r] = R::fma_(b, rv[r], x[cp][r]);
        D[cp] *= ru2;
        nrm[cp] += kg22;
      }
    }
  }
}

// The same round with the steps of one t processed TWO AT A TIME (they touch disjoint slots): both
// partner columns are requested first (2 x MR cross-lane fetches back to back), then both inner
// products / rotations are evaluated, then both second fetches, then both second updates.  In the
// one-step-at-a-time form the compiler paces every inner product by the LDS crossbar (fetch, wait,
// fma, five deep) and a wave spends a quarter of its time in s_waitcnt; with two steps in flight the
// second step's fetch latency hides behind the first step's arithmetic.  Same rotations, same order
// of the floating-point operations inside every step: results are bitwise identical.
#ifndef SQFA_FETCH_PRIO
#define SQFA_FETCH_PRIO 3
#endif
#ifndef SQFA_PARAM_PRIO
#define SQFA_PARAM_PRIO 1
#endif
#ifndef SQFA_PARAM_PRIO_ALL
#define SQFA_PARAM_PRIO_ALL -1  // -1: by size
#endif
// wave priority 1 in the phases around the sweeps -- bit 0: X formation, bit 1: back-transform, bit 2: rank-one sums; -1: float32
// rank-one sums (m=16 0.688 -> 0.677 ms, m=17 -0.6 %, m=24 -0.3 %), all three from 32 rows on (m=32 5.96 -> 5.87; bit 2 alone -0.2 %)
#ifndef SQFA_BWD_PRIO
#define SQFA_BWD_PRIO -1
#endif
template <typename T, int MR, int CPL, int S, int SWZ, bool LONE_LAST, int C0, int C1, int T_, int RS = 0>
__device__ __forceinline__ void cross_step2(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], int s, T tol2, T tie, bool& big) {
  using R = Real<T>;
  constexpr int NS = (C1 >= 0) ? 2 : 1;
  constexpr int cs[2] = {C0, C1 >= 0 ? C1 : C0};
  constexpr int cps[2] = {C0 ^ T_, (C1 >= 0 ? C1 : C0) ^ T_};
  T rv[2][MR];
  // the fetch bursts are issued at raised priority: the sooner a wave's 2 x MR crossbar requests are in
  // flight, the more of their latency its SIMD neighbours' arithmetic covers (measured, both bursts at
  // priority 3: m=16 -2.5 %, m=17 -1 %, m=32 -3.5 %, m=33 -4 %)
  if (SQFA_FETCH_PRIO) __builtin_amdgcn_s_setprio(SQFA_FETCH_PRIO);
#pragma unroll
  for (int q = 0; q < NS; ++q) {
#pragma unroll
    for (int r = 0; r < MR; ++r) rv[q][r] = lane_xor_row<S, SWZ>(x[cps[q]][r], s, r);  // partner's slot cp
  }
  if (SQFA_FETCH_PRIO) __builtin_amdgcn_s_setprio(0);
  T nr1[2], Dp[2];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    nr1[q] = lane_xor<S>(nrm[cps[q]], s);
    Dp[q] = lane_xor<S>(D[cps[q]], s);
  }
  __builtin_amdgcn_sched_barrier(0);
  T kgh[2], kg2[2], ru1[2];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const int c = cs[q];
    // m=16 (four waves per SIMD): the inner product + parameter chain, the latency-critical part of a
    // step, also runs at raised priority (-1.4 %; +1 % at m=17, no change at m=32: off there)
    // (round 4, slot-exchange sweeps: every float32 row below 32 rows -- m=17 -0.7 %, m=24 -0.8 %; m=32 +0.4 %)
    constexpr bool PARAM_PRIO = SQFA_PARAM_PRIO != 0 && sizeof(T) == 4 && (SQFA_PARAM_PRIO_ALL > 0 || (SQFA_PARAM_PRIO_ALL < 0 && MR < 32) || (MR == 16 && CPL == 4));
    if (PARAM_PRIO) __builtin_amdgcn_s_setprio(SQFA_PARAM_PRIO);
    const T gh = row_total<RS>(dot_cols<T, MR>(x[c], rv[q]));
    T u1, k1, g21;
    rot_scaled<T, MR>(nrm[c], nr1[q], gh, D[c], Dp[q], tol2, tie, u1, ru1[q], k1, g21, big);
    kgh[q] = k1 * gh;
    kg2[q] = k1 * g21;
    if (PARAM_PRIO) __builtin_amdgcn_s_setprio(0);
    const T a = -(kgh[q] * Dp[q]);
#pragma unroll
    for (int r = 0; r < MR; ++r) x[c][r] = R::fma_(a, rv[q][r], x[c][r]);
    D[c] *= u1;
    nrm[c] -= kg2[q];
  }
  if constexpr (T_ != 0) {
    __builtin_amdgcn_sched_barrier(0);
    // my slot cp meets the partner's slot c: the partner has just evaluated exactly that rotation
    // from its side (as ITS slot-c rotation); its k is mine with the sign flipped.
    T kgh2[2], kg22[2], ru2[2], Dpn[2];
    if (SQFA_FETCH_PRIO) __builtin_amdgcn_s_setprio(SQFA_FETCH_PRIO);
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      kgh2[q] = lane_xor<S>(kgh[q], s);
      kg22[q] = lane_xor<S>(kg2[q], s);
      ru2[q] = lane_xor<S>(ru1[q], s);
      Dpn[q] = lane_xor<S>(D[cs[q]], s);  // partner's slot c, already rescaled
#pragma unroll
      for (int r = 0; r < MR; ++r) rv[q][r] = lane_xor_row<S, SWZ>(x[cs[q]][r], s, r);  // partner's slot c, rotated
    }
    if (SQFA_FETCH_PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const int cp = cps[q];
      const T b = kgh2[q] * Dpn[q];
#pragma unroll
      for (int r = 0; r < MR; ++r) x[cp][r] = R::fma_(b, rv[q][r], x[cp][r]);
      D[cp] *= ru2[q];
      nrm[cp] += kg22[q];
    }
  }
}

// steps of round t: the slots c < (c ^ t) < CPL (t != 0) or all c (t == 0), two at a time
template <typename T, int MR, int CPL, int S, int SWZ, bool LONE_LAST, int T_, int CSTART, int RS = 0>
__device__ __forceinline__ void cross_t_steps(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], int s, T tol2, T tie, bool& big) {
  // find the next two valid first slots from CSTART on (compile time)
  constexpr auto valid = [](int c) constexpr {
    constexpr int CE = tournament_slots<CPL, LONE_LAST>();
    const int cp = c ^ T_;
    if (c >= CE || cp < c || cp >= CE) return false;
    if (LONE_LAST && c == CPL - 1 && cp == CPL - 1) return false;
    return true;
  };
  constexpr int first = [&]() constexpr { int c = CSTART; while (c < CPL && !valid(c)) ++c; return c; }();
  if constexpr (first < CPL) {
    constexpr int second = [&]() constexpr { int c = first + 1; while (c < CPL && !valid(c)) ++c; return c; }();
    if constexpr (second < CPL) {
      cross_step2<T, MR, CPL, S, SWZ, LONE_LAST, first, second, T_, RS>(x, nrm, D, s, tol2, tie, big);
      cross_t_steps<T, MR, CPL, S, SWZ, LONE_LAST, T_, second + 1, RS>(x, nrm, D, s, tol2, tie, big);
    } else {
      cross_step2<T, MR, CPL, S, SWZ, LONE_LAST, first, -1, T_, RS>(x, nrm, D, s, tol2, tie, big);
    }
  }
}

template <typename T, int MR, int CPL, int S, int SWZ, bool LONE_LAST, int T_ = 0, int RS = 0>
__device__ __forceinline__ void cross_round_paired(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], int s, T tol2, bool& big) {
  constexpr int TP2 = pow2ceil(tournament_slots<CPL, LONE_LAST>());
  if constexpr (T_ < TP2) {
    const int lane_id = (int)(threadIdx.x & 63);
    const T tie = ((lane_id ^ s) > lane_id) ? T(1) : T(-1);
    cross_t_steps<T, MR, CPL, S, SWZ, LONE_LAST, T_, 0, RS>(x, nrm, D, s, tol2, tie, big);
    cross_round_paired<T, MR, CPL, S, SWZ, LONE_LAST, T_ + 1, RS>(x, nrm, D, s, tol2, big);
  }
}

// Two steps in flight (cross_step2) where it was measured to pay (C=1000, MI355X): float32 4-lane
// groups from MR = 16 on (m=16 -1.8 %, m=17 -4.5 %; m=12 +1.5 %: off) and float32 8-lane groups
// (m=24 -3.5 %, m=32 -6 %, m=33 -7.6 %: two waves per SIMD hide less latency by themselves).
#ifndef SQFA_PAIRED_STEPS
#define SQFA_PAIRED_STEPS 1
#endif
#ifndef SQFA_PAIRED_F64
#define SQFA_PAIRED_F64 1  // float64: m=12 -3.5 %, m=17 -3 %, m=8 / m=16 unchanged
#endif
template <typename T, int G, int MR> constexpr bool paired_steps() {
  if (!SQFA_PAIRED_STEPS) return false;
  if (sizeof(T) == 8) return SQFA_PAIRED_F64 != 0;
#ifndef SQFA_PAIRED_G16
#define SQFA_PAIRED_G16 0
#endif
#ifndef SQFA_PAIRED_G2
#define SQFA_PAIRED_G2 0  // round 4 A/B rows (2 lanes x 8 / 9 slots): profiles/r4_pairs_fewer_lanes.txt
#endif
  return (G == 4 && MR >= 16) || G == 8 || (SQFA_PAIRED_G16 && G == 16) || (SQFA_PAIRED_G2 && G == 2);
}

// LONE: -1 = derived from the sizes (whole columns per lane: MR is the matrix size); 0 / 1 = given (2-D layouts: MR is the
// number of rows a lane holds)
template <typename T, int MR, int G, int CPL, int S, int LONE = -1, int RS = 0>
__device__ __forceinline__ void cross_rounds_static(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], T tol2, bool& big) {
  if constexpr (S < G) {
    constexpr bool LONE_LAST = LONE < 0 ? (MR == G * (CPL - 1) + 1) : (LONE != 0);
    if constexpr (paired_steps<T, G, MR>())
      cross_round_paired<T, MR, CPL, S, swizzled_rows_of_8<T, G, MR>(), LONE_LAST, 0, RS>(x, nrm, D, S, tol2, big);
    else
      cross_round<T, MR, CPL, S, swizzled_rows_of_8<T, G, MR>(), LONE_LAST, RS>(x, nrm, D, S, tol2, big);
    cross_rounds_static<T, MR, G, CPL, S + 1, LONE, RS>(x, nrm, D, tol2, big);
  }
}

// Rotations between the columns of one lane, in tournament order: round t pairs slot c with c ^ t, and
// the (up to two) pairs of a round that are handled together touch disjoint slots, so their inner
// products and parameter chains are independent instruction streams (the plain c1 < c2 double loop
// is one long dependency chain through slot 0: fine with four waves per SIMD, exposed with two).
#ifndef SQFA_LOCAL_PARAM_PRIO
#define SQFA_LOCAL_PARAM_PRIO -1  // -1: by element type
#endif
#ifndef SQFA_LOCAL_TOURNAMENT
#define SQFA_LOCAL_TOURNAMENT 1
#endif
template <typename T, int MR, int CPL, int A0, int B0, int A1, int B1, int RS = 0>
__device__ __forceinline__ void local_step2(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], T tol2, bool& big) {
  using R = Real<T>;
  constexpr int NS = A1 >= 0 ? 2 : 1;
  constexpr int ca[2] = {A0, A1 >= 0 ? A1 : A0}, cb[2] = {B0, A1 >= 0 ? B1 : B0};
  T gh[2] = {T(0), T(0)};
#ifdef SQFA_ABL_NO_DOT  // development: timing ablation, wrong results
#pragma unroll
  for (int q = 0; q < NS; ++q) gh[q] = x[ca[q]][0] * x[cb[q]][1];
#else
  if constexpr (dot_accs<T, MR>() > 1) {
#pragma unroll
    for (int q = 0; q < NS; ++q) gh[q] = dot_cols<T, MR>(x[ca[q]], x[cb[q]]);
  } else {
#pragma unroll
    for (int r = 0; r < MR; ++r) {
#pragma unroll
      for (int q = 0; q < NS; ++q) gh[q] = R::fma_(x[ca[q]][r], x[cb[q]][r], gh[q]);
    }
  }
#endif
  if constexpr (RS != 0) {
#pragma unroll
    for (int q = 0; q < NS; ++q) gh[q] = row_total<RS>(gh[q]);
  }
  T a1[2], a2[2];
  // the parameter chains (serial, two transcendentals each) run at raised wave priority: float32 m=16 0.706 -> 0.682 ms, m=17
  // 1.070 -> 1.024, m=32 6.14 -> 5.98; float64 m=16 1.584 -> 1.594: off there.  (Priority 3: the same; inner products included:
  // slightly worse.)
  constexpr int LP = SQFA_LOCAL_PARAM_PRIO >= 0 ? SQFA_LOCAL_PARAM_PRIO : (sizeof(T) == 4 ? 1 : 0);
  if (LP) __builtin_amdgcn_s_setprio(LP);
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    T u, kgh, kg4;
    rot_local<T, MR>(nrm[ca[q]], nrm[cb[q]], gh[q], D[ca[q]], D[cb[q]], u, kgh, kg4, big);
    a1[q] = -(kgh * D[cb[q]]);
    a2[q] = kgh * D[ca[q]];
    D[ca[q]] *= u;
    D[cb[q]] *= u;
    nrm[ca[q]] = R::fma_(T(-0.5), kg4, nrm[ca[q]]);
    nrm[cb[q]] = R::fma_(T(0.5), kg4, nrm[cb[q]]);
  }
  if (LP) __builtin_amdgcn_s_setprio(0);
#ifdef SQFA_ABL_NO_UPDATE  // development: timing ablation, wrong results (two rows keep the parameters alive)
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#else
#pragma unroll
  for (int r = 0; r < MR; ++r) {
#endif
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const T xp = x[ca[q]][r];
      x[ca[q]][r] = R::fma_(a1[q], x[cb[q]][r], xp);
      x[cb[q]][r] = R::fma_(a2[q], xp, x[cb[q]][r]);
    }
  }
}
template <typename T, int MR, int CPL, int T_, int CSTART>
__device__ __forceinline__ void local_t_steps(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], T tol2, bool& big) {
  constexpr auto valid = [](int c) constexpr { return c < CPL && (c ^ T_) > c && (c ^ T_) < CPL; };
  constexpr int first = [&]() constexpr { int c = CSTART; while (c < CPL && !valid(c)) ++c; return c; }();
  if constexpr (first < CPL) {
    constexpr int second = [&]() constexpr { int c = first + 1; while (c < CPL && !valid(c)) ++c; return c; }();
    if constexpr (second < CPL) {
      local_step2<T, MR, CPL, first, first ^ T_, second, second ^ T_>(x, nrm, D, tol2, big);
      local_t_steps<T, MR, CPL, T_, second + 1>(x, nrm, D, tol2, big);
    } else {
      local_step2<T, MR, CPL, first, first ^ T_, -1, -1>(x, nrm, D, tol2, big);
    }
  }
}
template <typename T, int MR, int CPL, int C, int RS = 0>
__device__ __forceinline__ void z_visit_steps(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], T tol2, bool& big) {
  if constexpr (C < CPL - 1) {
    local_step2<T, MR, CPL, C, CPL - 1, -1, -1, RS>(x, nrm, D, tol2, big);
    z_visit_steps<T, MR, CPL, C + 1, RS>(x, nrm, D, tol2, big);
  }
}
template <typename T, int MR, int CPL, int T_>
__device__ __forceinline__ void local_rounds(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], T tol2, bool& big) {
  if constexpr (T_ < pow2ceil(CPL)) {
    local_t_steps<T, MR, CPL, T_, 0>(x, nrm, D, tol2, big);
    local_rounds<T, MR, CPL, T_ + 1>(x, nrm, D, tol2, big);
  }
}

// The travelling lone column (see SQFA_Z_VISITS above): visit V rotates the last slot against the other
// slots of the lane, then every lane takes its partner's last slot.  The moves follow the reflected Gray
// code (xor 1, 2, 1, 4, 1, 2, 1, ...), the last one (xor G/2) brings the column home.
template <typename T, int MR, int G, int CPL, int SWZ, int V, int RS = 0>
__device__ __forceinline__ void z_visits(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], T tol2, bool& big) {
  if constexpr (V < G) {
    constexpr int Z = CPL - 1;
    z_visit_steps<T, MR, CPL, 0, RS>(x, nrm, D, tol2, big);
    constexpr int NXT = V + 1;
    constexpr int BIT = (NXT == G) ? G / 2 : (NXT & -NXT);
    constexpr int ZP = SQFA_Z_PRIO >= 0 ? SQFA_Z_PRIO : (sizeof(T) == 4 ? 3 : 0);
    if (ZP) __builtin_amdgcn_s_setprio(ZP);
#pragma unroll
    for (int r = 0; r < MR; ++r) x[Z][r] = lane_xor_row<BIT, SWZ>(x[Z][r], BIT, r);
    nrm[Z] = lane_xor<BIT>(nrm[Z], BIT);
    D[Z] = lane_xor<BIT>(D[Z], BIT);
    if (ZP) __builtin_amdgcn_s_setprio(0);
    z_visits<T, MR, G, CPL, SWZ, V + 1, RS>(x, nrm, D, tol2, big);
  }
}

// ---------------------------------------------------------------------------------------
// Slot-exchange ordering of a sweep (round 4).  In the tournament above the columns never move: a rotation between
// columns of different lanes is evaluated by BOTH owners, each fetching the partner's column (MR cross-lane moves per
// rotated column, 2 MR per rotation) -- at 4 lanes x 4 slots 96 of the 120 rotations of a sweep are of that kind and the
// moves are a quarter of all vector instructions.  Here the COLUMNS travel instead: slot s of every lane is exchanged
// with the same slot of lane ^ mask (one in-place cross-lane move per element, nothing else), after which the
// rotations between DIFFERENT slots are local to a lane -- no fetch, no second owner, no parameters to pass on.
//
// Which exchanges: write the lane index l and a phase counter k as elements of GF(2^g), G = 2^g lanes per group, and
// give slot s the multiplier mu_s = s (as a field element; slot 0 never moves).  In phase k the column dealt to
// (l, s) sits in lane l + mu_s k.  Columns (l, s), (l', s') with s != s' share a lane iff (mu_s + mu_s') k = l + l',
// and since mu_s + mu_s' != 0 that happens in exactly ONE of the G phases: every such pair is local exactly once per
// sweep.  Going from phase k to k + d moves slot s by the lane mask mu_s d; the phases are visited along the cyclic
// reflected Gray code (d = 1, 2, 1, 4, ..., closing with G/2), so the G-th exchange brings every column home and a
// sweep is G x [exchange the slots 1..CE-1, rotate all CE (CE-1) / 2 slot pairs locally].  Columns of the SAME slot
// are always in different lanes: those G (G-1) / 2 pairs per slot keep the two-owner form (the t = 0 steps of the
// tournament, one fetch and one update per owner).  Per lane and sweep at 4 x 4: 24 MR moves instead of 48 MR, the
// same 96 MR fmas; at 8 x 4: 52 MR instead of 112 MR.  Needs 2 <= CE <= G slots (CE: the slots of the tournament).
#ifndef SQFA_SLOT_EXCHANGE
#define SQFA_SLOT_EXCHANGE 1
#endif
// The bursts of cross-lane moves (slot exchanges, the transposition's LDS passes) are issued at raised wave priority, like the
// fetch bursts of the two-owner steps: the sooner a wave's moves are in flight, the more of their latency its SIMD neighbours'
// arithmetic covers.  C=1000, pair kernel ms without -> with priority 3: m=16 0.708 -> 0.704, m=17 1.101 -> 1.081, m=24 2.528 ->
// 2.375, m=32 6.26 -> 6.03, m=33 9.16 -> 8.88, m=48 (C=300) 3.135 -> 2.855; m=12 and float64 unchanged (priority 1: the same).
#ifndef SQFA_EXCH_PRIO
#define SQFA_EXCH_PRIO 3
#endif

#ifndef SQFA_PHASE_BARRIER
#define SQFA_PHASE_BARRIER 0
#endif
#ifndef SQFA_SLOT_EXCHANGE_MAX_G
#define SQFA_SLOT_EXCHANGE_MAX_G 16
#endif
__host__ __device__ constexpr int gf2_mul(int a, int b, int g) {
  const int poly = g == 1 ? 0x3 : (g == 2 ? 0x7 : (g == 3 ? 0xB : (g == 4 ? 0x13 : 0x25)));  // x+1, x^2+x+1, x^3+x+1, x^4+x+1, x^5+x^2+1
  int p = 0;
  for (int i = 0; i < g; ++i)
    if ((b >> i) & 1) p ^= a << i;
  for (int i = 2 * g - 2; i >= g; --i)
    if ((p >> i) & 1) p ^= poly << (i - g);
  return p;
}
template <int G, int CE> constexpr bool slot_exchange_ok() {  // CE: the slots that take part in the tournament
  return SQFA_SLOT_EXCHANGE && G >= 2 && G <= SQFA_SLOT_EXCHANGE_MAX_G && CE >= 2 && CE <= G;
}
template <int G, int MR, int CPL> constexpr bool slot_exchange_cfg() {
  return slot_exchange_ok<G, (z_visits_cfg<G, MR, CPL>() ? CPL - 1 : CPL)>();
}
// Compile-time proof of the schedule for one group shape: walking the G phases, every pair of columns from DIFFERENT slots
// shares a lane in exactly one phase, columns of the same slot never do, and the last exchange brings every column home.
template <int G, int CE> constexpr bool slot_exchange_schedule_ok() {
  constexpr int g = ilog2(G);
  int lane_of[G * CE] = {};  // current lane of the column dealt to (l, s): index l * CE + s
  for (int l = 0; l < G; ++l)
    for (int s = 0; s < CE; ++s) lane_of[l * CE + s] = l;
  int met[G * CE][G * CE] = {};
  for (int ph = 1; ph <= G; ++ph) {
    const int delta = ph == G ? G / 2 : (ph & -ph);
    for (int l = 0; l < G; ++l)
      for (int s = 1; s < CE; ++s) lane_of[l * CE + s] ^= gf2_mul(s, delta, g);
    for (int a = 0; a < G * CE; ++a)
      for (int b = a + 1; b < G * CE; ++b)
        if (lane_of[a] == lane_of[b]) ++met[a][b];
  }
  for (int a = 0; a < G * CE; ++a) {
    if (lane_of[a] != a / CE) return false;  // not home again
    for (int b = a + 1; b < G * CE; ++b)
      if (met[a][b] != ((a % CE) != (b % CE) ? 1 : 0)) return false;
  }
  return true;
}
// exchange slots S_..CE-1 for the phase step DELTA
template <typename T, int MR, int G, int CPL, int CE, int SWZ, int DELTA, int S_ = 1>
__device__ __forceinline__ void exchange_slots(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL]) {
  if constexpr (S_ < CE) {
    constexpr int M = gf2_mul(S_, DELTA, ilog2(G));
    static_assert(M > 0 && M < G, "exchange partner outside the lane group");
#pragma unroll
    for (int r = 0; r < MR; ++r) x[S_][r] = lane_xor_row<M, SWZ>(x[S_][r], M, r);
    nrm[S_] = lane_xor<M>(nrm[S_], M);
    D[S_] = lane_xor<M>(D[S_], M);
    exchange_slots<T, MR, G, CPL, CE, SWZ, DELTA, S_ + 1>(x, nrm, D);
  }
}
// all pairs among the slots 0..CE-1 of a lane, tournament order, two disjoint pairs at a time
template <typename T, int MR, int CPL, int CE, int T_, int CSTART, int RS>
__device__ __forceinline__ void slot_t_steps(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], T tol2, bool& big) {
  constexpr auto valid = [](int c) constexpr { return c < CE && (c ^ T_) > c && (c ^ T_) < CE; };
  constexpr int first = [&]() constexpr { int c = CSTART; while (c < CE && !valid(c)) ++c; return c; }();
  if constexpr (first < CE) {
    constexpr int second = [&]() constexpr { int c = first + 1; while (c < CE && !valid(c)) ++c; return c; }();
    if constexpr (second < CE) {
      local_step2<T, MR, CPL, first, first ^ T_, second, second ^ T_, RS>(x, nrm, D, tol2, big);
      slot_t_steps<T, MR, CPL, CE, T_, second + 1, RS>(x, nrm, D, tol2, big);
    } else {
      local_step2<T, MR, CPL, first, first ^ T_, -1, -1, RS>(x, nrm, D, tol2, big);
    }
  }
}
template <typename T, int MR, int CPL, int CE, int T_, int RS>
__device__ __forceinline__ void slot_rounds(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], T tol2, bool& big) {
  if constexpr (T_ < pow2ceil(CE)) {
    slot_t_steps<T, MR, CPL, CE, T_, 0, RS>(x, nrm, D, tol2, big);
    slot_rounds<T, MR, CPL, CE, T_ + 1, RS>(x, nrm, D, tol2, big);
  }
}
template <typename T, int MR, int G, int CPL, int CE, int SWZ, int RS, int PH = 1>
__device__ __forceinline__ void exchange_phases(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], T tol2, bool& big) {
  if constexpr (PH <= G) {
    constexpr int DELTA = PH == G ? G / 2 : (PH & -PH);
#ifndef SQFA_ABL_NO_EXCHANGE
#if SQFA_EXCH_PRIO
    __builtin_amdgcn_s_setprio(SQFA_EXCH_PRIO);
#endif
    exchange_slots<T, MR, G, CPL, CE, SWZ, DELTA>(x, nrm, D);
#if SQFA_EXCH_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#endif
#if SQFA_PHASE_BARRIER
    __builtin_amdgcn_sched_barrier(0);
#endif
    slot_rounds<T, MR, CPL, CE, 1, RS>(x, nrm, D, tol2, big);
    exchange_phases<T, MR, G, CPL, CE, SWZ, RS, PH + 1>(x, nrm, D, tol2, big);
  }
}
// the same-slot pairs: the t = 0 steps of the tournament against every lane of the group
template <typename T, int MR, int CPL, int S, int SWZ, bool LONE_LAST, int RS, int C>
__device__ __forceinline__ void same_slot_single(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], T tol2, T tie, bool& big) {
  if constexpr (C < tournament_slots<CPL, LONE_LAST>()) {
    cross_step2<T, MR, CPL, S, SWZ, LONE_LAST, C, -1, 0, RS>(x, nrm, D, S, tol2, tie, big);
    same_slot_single<T, MR, CPL, S, SWZ, LONE_LAST, RS, C + 1>(x, nrm, D, tol2, tie, big);
  }
}
template <typename T, int MR, int G, int CPL, int SWZ, bool LONE_LAST, int RS, int S = 1>
__device__ __forceinline__ void same_slot_rounds(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], T tol2, bool& big) {
  if constexpr (S < G) {
    const int lane_id = (int)(threadIdx.x & 63);
    const T tie = ((lane_id ^ S) > lane_id) ? T(1) : T(-1);
    if constexpr (paired_steps<T, G, MR>()) {
      cross_t_steps<T, MR, CPL, S, SWZ, LONE_LAST, 0, 0, RS>(x, nrm, D, S, tol2, tie, big);
    } else {
      same_slot_single<T, MR, CPL, S, SWZ, LONE_LAST, RS, 0>(x, nrm, D, tol2, tie, big);
    }
    same_slot_rounds<T, MR, G, CPL, SWZ, LONE_LAST, RS, S + 1>(x, nrm, D, tol2, big);
  }
}
// Square groups (4 lanes x 4 tournament slots: m = 16, 17): the same-slot pairs become local as well once the 4 x 4 block
// (lane, slot) of every row is TRANSPOSED -- column (l, s) moves to lane s, slot l.  That is a lane-dependent choice of
// registers, which costs selects on the VALU; through LDS it costs no vector instruction at all: per row, four 4-byte
// writes at the transposed positions and one 16-byte read (the wave's own 1 KB scratch: the L_j^-1 staging area, idle
// during the sweeps and reloaded for the back-transform; LDS operations of a wave execute in order, so consecutive rows
// reuse the buffer without waiting).  A sweep is then G x [exchange, rotate locally] + transpose + rotate locally:
// every rotation is local and evaluated once (30 per lane at 4 x 4 instead of 24 + 12 two-owner half steps).  The
// layout alternates between the two orientations from sweep to sweep; an odd sweep count is undone by one more
// transposition after the loop, so everything outside the sweeps sees the dealt positions.
// Measured (C=1000, ms, two-owner same-slot steps -> transposition; profiles/r4_pairs_slot_exchange.txt): float32 m=16
// 0.771 -> 0.753, float64 m=16 1.737 -> 1.678; but m=17 (three / two waves per SIMD and the lone column's serial visits on top)
// 1.161 -> 1.210 and 3.08 -> 3.22 although the sweep is 11 % shorter in instructions: the wave sits out the LDS round trip
// and nobody covers for it.  SQFA_SLOT_TRANSPOSE: 1 = full groups only (m = 16), 2 = also with a lone column
#ifndef SQFA_SLOT_TRANSPOSE
#define SQFA_SLOT_TRANSPOSE 1
#endif
// 8 lanes x 4 slots (m = 32): the same transposition inside every quad of lanes makes the same-slot pairs whose lanes differ
// in the two low bits local; what is left -- lanes that differ in bit 2, same slot before, ANY slot pair after -- is one full
// round of the two-owner tournament against lane ^ 4 (10 steps instead of the 16 of the 28 same-slot half steps it replaces).
#ifndef SQFA_SLOT_TRANSPOSE_G8
#define SQFA_SLOT_TRANSPOSE_G8 1
#endif
// (m=32 6.58 -> 6.29 ms, m=33 with its lone column 9.49 -> 9.14 ms at C=1000; the 4-lane group with a lone column, m=17,
// loses as said above)
template <int G, int MR, int CPL> constexpr bool slot_transpose_cfg() {
  constexpr int CE = z_visits_cfg<G, MR, CPL>() ? CPL - 1 : CPL;
  return SQFA_SLOT_TRANSPOSE && slot_exchange_cfg<G, MR, CPL>() && CE == 4 &&
         ((G == 4 && (CE == CPL || SQFA_SLOT_TRANSPOSE >= 2)) || (G == 8 && SQFA_SLOT_TRANSPOSE_G8));
}
// development switch: transpose back at the end of every sweep (same pair order in every sweep, twice the LDS passes)
#ifndef SQFA_SLOT_TRANSPOSE_TWICE
#define SQFA_SLOT_TRANSPOSE_TWICE 0
#endif
constexpr int kTransposePitch = 72;
constexpr int kTransposeElems = 3 * kTransposePitch + 64;  // elements of wave-private LDS `buf` must provide
template <typename T, int MR, int CPL>
__device__ __forceinline__ void transpose_slots(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], T* buf) {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));  // re-derived per call: not worth a register across the sweeps
  const int lane = t & 63;
  // slot s of the wave is written as one row of 64 lanes (row pitch kTransposePitch), lane (q, j) then reads the four lanes of
  // its quad from row j: 16 aligned bytes.  Pitch 72 = 8 mod 32 banks: the writes of a row are consecutive, the 16-byte reads of
  // eight neighbouring lanes fall into eight different 4-bank groups -- no conflicts on either side (pitch 64 serialises the
  // reads four-fold, the (lane, slot)-major layout the writes)
  T* wr = buf + lane;
  const T* rd = buf + (lane & ~3) + kTransposePitch * (lane & 3);
  auto pass = [&](T& v0, T& v1, T& v2, T& v3) {
    wr[0] = v0;
    wr[kTransposePitch] = v1;
    wr[2 * kTransposePitch] = v2;
    wr[3 * kTransposePitch] = v3;
    if constexpr (sizeof(T) == 4) {
      const float4 q = *reinterpret_cast<const float4*>(rd);
      v0 = q.x; v1 = q.y; v2 = q.z; v3 = q.w;
    } else {
      const double2 q0 = *reinterpret_cast<const double2*>(rd), q1 = *reinterpret_cast<const double2*>(rd + 2);
      v0 = q0.x; v1 = q0.y; v2 = q1.x; v3 = q1.y;
    }
  };
#pragma unroll
  for (int r = 0; r < MR; ++r) pass(x[0][r], x[1][r], x[2][r], x[3][r]);
  pass(nrm[0], nrm[1], nrm[2], nrm[3]);
  pass(D[0], D[1], D[2], D[3]);
}

// one sweep over the tournament slots (the lone column of the m = G (CPL-1) + 1 sizes still travels by z_visits)
template <typename T, int MR, int G, int CPL, bool LONE_LAST, bool TRANSPOSE = false, int RS = 0>
__device__ __forceinline__ void exchange_sweep(T (&x)[CPL][MR], T (&nrm)[CPL], T (&D)[CPL], T tol2, bool& big, T* buf = nullptr) {
  constexpr int CE = tournament_slots<CPL, LONE_LAST>();
  constexpr int SWZ = swizzled_rows_of_8<T, G, MR, true>();
  static_assert(slot_exchange_schedule_ok<G, CE>(), "slot-exchange schedule does not cover every pair exactly once");
  exchange_phases<T, MR, G, CPL, CE, SWZ, RS>(x, nrm, D, tol2, big);
  if constexpr (TRANSPOSE) {
#ifndef SQFA_ABL_NO_EXCHANGE
#if SQFA_EXCH_PRIO
    __builtin_amdgcn_s_setprio(SQFA_EXCH_PRIO);
#endif
    transpose_slots<T, MR, CPL>(x, nrm, D, buf);
#if SQFA_EXCH_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#endif
#if SQFA_PHASE_BARRIER
    __builtin_amdgcn_sched_barrier(0);
#endif
    slot_rounds<T, MR, CPL, CE, 1, RS>(x, nrm, D, tol2, big);
    if constexpr (G == 8) {  // lanes that differ in bit 2: every slot pair, two owners
      if constexpr (paired_steps<T, G, MR>()) cross_round_paired<T, MR, CPL, 4, SWZ, LONE_LAST, 0, RS>(x, nrm, D, 4, tol2, big);
      else cross_round<T, MR, CPL, 4, SWZ, LONE_LAST, RS>(x, nrm, D, 4, tol2, big);
    }
    if constexpr (SQFA_SLOT_TRANSPOSE_TWICE != 0) transpose_slots<T, MR, CPL>(x, nrm, D, buf);
  } else {
    same_slot_rounds<T, MR, G, CPL, SWZ, LONE_LAST, RS>(x, nrm, D, tol2, big);
  }
}

// ---------------------------------------------------------------------------------------
// Transposing tree reduction.  Every lane holds a value for each of 2^LEVEL consecutive
// indices [BASE, BASE + 2^LEVEL); the sum over the 2^LEVEL lanes of an aligned lane block is
// wanted for each index.  A butterfly would cost LEVEL cross-lane adds PER INDEX and leave
// the result replicated; here, at distance d = 2^(LEVEL-1) a lane keeps the half of the
// indices selected by its own bit and hands the other half to its partner, so the work
// halves per level (~2 cross-lane adds per index in total) and lane l of the block ends with
// the finished sum of index BASE + l -- ready for a conflict-free LDS update / coalesced store.
__device__ __forceinline__ float bpermute(int byte_addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ double bpermute(int byte_addr, double v) {
  return __hiloint2double(__builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v)),
                          __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v)));
}
template <int D, typename T> __device__ __forceinline__ T xor_fetch(T v, int lane) {
  if constexpr (D == 1) return dpp_mov<0xB1>(v);
  else if constexpr (D == 2) return dpp_mov<0x4E>(v);
  else if constexpr (D < 32) return swizzle_xor<D>(v);
  else return bpermute((lane ^ D) << 2, v);  // the caller's lane id (not a hoisted mbcnt)
}

__host__ __device__ constexpr int tri_row(int idx) {
  int r = 0;
  while ((r + 1) * (r + 2) / 2 <= idx) ++r;
  return r;
}

// lower-triangle entry IDX of  sum_c coef[c] * x[c] x[c]^T  over this lane's column slots
template <typename T, int MR, int CPL>
struct OuterProduct {
  const T (&x)[CPL][MR];
  const T (&coef)[CPL];
  template <int IDX> __device__ __forceinline__ T get() const {
    if constexpr (IDX >= MR * (MR + 1) / 2) {
      return T(0);
    } else {
      constexpr int r = tri_row(IDX), cc = IDX - r * (r + 1) / 2;
      T acc = T(0);
#pragma unroll
      for (int c = 0; c < CPL; ++c) acc = Real<T>::fma_(coef[c] * x[c][r], x[c][cc], acc);
      return acc;
    }
  }
};

#ifndef SQFA_BACK_SCHED_BARRIER
#define SQFA_BACK_SCHED_BARRIER -1  // -1: by size, 0: never, 1: always
#endif
#ifndef SQFA_SWAP_MIN
#define SQFA_SWAP_MIN 16
#endif
template <int LEVEL, int BASE, typename T, typename P>
__device__ __forceinline__ T tree_reduce(const P& prod, int lane) {
  if constexpr (LEVEL == 0) {
    return prod.template get<BASE>();
  } else {
    constexpr int H = 1 << (LEVEL - 1);
    const T a = tree_reduce<LEVEL - 1, BASE, T>(prod, lane);
    const T b = tree_reduce<LEVEL - 1, BASE + H, T>(prod, lane);
    if constexpr (H >= SQFA_SWAP_MIN) {
      return row_swap_sum<H>(a, b);
    } else {
      const bool upper = (lane & H) != 0;
      const T keep = upper ? b : a;
      const T send = upper ? a : b;
      return keep + xor_fetch<H>(send, lane);
    }
  }
}

#ifndef SQFA_TREE_SCHED_BARRIER
#define SQFA_TREE_SCHED_BARRIER 1
#endif
template <int LEVEL, int I, int N, int TRI, typename T, typename P, typename F>
__device__ __forceinline__ void tree_reduce_blocks(const P& prod, int lane, F&& sink) {
  if constexpr (I < N) {
    constexpr int W = 1 << LEVEL;
    const T v = tree_reduce<LEVEL, I * W, T>(prod, lane);
    const int idx = I * W + (lane & (W - 1));
    if constexpr ((I + 1) * W <= TRI) {
      sink(idx, v);
    } else {
      if (idx < TRI) sink(idx, v);
    }
#if SQFA_TREE_SCHED_BARRIER
    __builtin_amdgcn_sched_barrier(0);  // keep the blocks apart: interleaving them stretches live ranges into spills
#endif
    tree_reduce_blocks<LEVEL, I + 1, N, TRI, T>(prod, lane, sink);
  }
}

__host__ __device__ constexpr int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// ---------------------------------------------------------------------------------------
template <typename T, int MR_, int G_, int CPL_, int TJ_, int WAVES_>
struct PairCfg {
  using type = T;
  static constexpr int MR = MR_;     // padded matrix size (rows, and real columns)
  static constexpr int G = G_;       // lanes per pair
  static constexpr int CPL = CPL_;   // column slots per lane (G*CPL >= MR)
  static constexpr int TJ = TJ_;     // B classes per tile
  static constexpr int WAVES = WAVES_;  // waves per workgroup
  static constexpr int THREADS = 64 * WAVES_;
  static constexpr int PPW = 64 / G; // pairs per wave
  static constexpr int TI = PPW;     // A classes per tile
  static constexpr int TRI = MR * (MR + 1) / 2;
  static constexpr int TRIP = TRI | 1;  // odd LDS stride between matrices
#ifndef SQFA_MAX_SWEEPS
#define SQFA_MAX_SWEEPS 30
#endif
  static constexpr int MAX_SWEEPS = SQFA_MAX_SWEEPS;
  // lane groups up to this size unroll the tournament rounds (compile-time partners: DPP / ds_swizzle);
  // larger groups loop over the partner at run time (ds_bpermute)
  // (round 2, C=1000: unrolling the 15 / 31 rounds of the 16- / 32-lane groups as well -- float32 m=48 67.1 -> 43.7 ms,
  // m=64 178.7 -> 123.6; float64 m=24 29.8 -> 13.1, m=32 43.3 -> 36.5 (with one wave per SIMD), m=33 58.3 -> 51.6;
  // float64 32-lane groups (m=48) spill out of 512 VGPRs when unrolled, 313 -> 1256 ms: run-time loop kept there)
#ifndef SQFA_STATIC_G
#define SQFA_STATIC_G 0  // 0: by element type
#endif
  static constexpr int STATIC_G = SQFA_STATIC_G > 0 ? SQFA_STATIC_G : (sizeof(T) == 4 ? 32 : 16);
  // register budget: waves per SIMD the kernel is compiled for (256-thread blocks)
  static constexpr int XREGS = CPL * MR * (int)(sizeof(T) / 4);
#ifndef SQFA_F64_SMALL_WAVES
#define SQFA_F64_SMALL_WAVES 3  // 168 VGPRs: measured 8 % faster than 2 waves at m=16 (LDS allows 3 workgroups per CU)
#endif
  // float64 16-lane groups with their 15 rounds unrolled: m=32 (128 registers of state) needs the whole file
  // (1261 spilled VGPRs at two waves per SIMD), m=24 (96) fits two waves
#ifndef SQFA_F64_TWO_WAVE_XREGS
#define SQFA_F64_TWO_WAVE_XREGS 180  // 140 until round 4: m=17 as 4 x 5 (170 registers of state) runs at two waves per SIMD
#endif
  static constexpr int F64_TWO_WAVE_XREGS = G_ >= 16 ? 100 : SQFA_F64_TWO_WAVE_XREGS;
  static constexpr int MIN_WAVES = sizeof(T) == 8 ? (XREGS <= 64 ? SQFA_F64_SMALL_WAVES : (XREGS <= F64_TWO_WAVE_XREGS ? 2 : 1))
                                                  : (XREGS <= 64 ? 4 : (XREGS <= 100 ? 3 : (XREGS <= 170 ? 2 : 1)));
  // L_j^-1 staged in LDS as a packed lower triangle (one-wave workgroups, m >= 32: 8 instead of 7
  // workgroups per CU) or as a full MR x MR block (smaller sizes: no occupancy to gain, and the
  // regular row pitch keeps the back-transform's LDS reads vectorised and out of the spill range)
  // K0b (class_factor_kernel below): the A-side factor handed to the pair kernel is not the Cholesky factor L_i but
  // L_i V_i with the columns made orthogonal by a few Jacobi sweeps of their own -- still a factor of Sigma_i, so the
  // pencil is unchanged, but the pair sweeps start from columns that are orthogonal in the plain inner product and
  // converge a sweep earlier ("class factors").  0 sweeps: the pass is off and L_i stays triangular.
#ifndef SQFA_FACTOR_SWEEPS
#define SQFA_FACTOR_SWEEPS -1  // -1: by size
#endif
#ifndef SQFA_FACTOR_MIN_M
#define SQFA_FACTOR_MIN_M 12
#endif
  static constexpr int FACTOR_SWEEPS = SQFA_FACTOR_SWEEPS >= 0 ? SQFA_FACTOR_SWEEPS : (MR_ >= SQFA_FACTOR_MIN_M ? 2 : 0);
  static constexpr bool DENSE_FACTOR = FACTOR_SWEEPS > 0;
  // the pass orthogonalises the columns in the metric of the mean class (class_factor_mean_kernel) for the sizes whose stacked
  // columns (2 MR rows of double state per slot) fit the register file of a lone wave
#ifndef SQFA_FACTOR_MEAN
#define SQFA_FACTOR_MEAN 1
#endif
  // (m = 20 / 24 / 33: two stacked double slots per lane spill -- the pass took 0.3-0.5 ms there; not offered)
  static constexpr bool MEAN_METRIC = SQFA_FACTOR_MEAN && DENSE_FACTOR && (MR_ <= 17 || MR_ == 32);
  // The pass is a handful of lone waves: its duration is one wave's latency whatever the class count (float32: 10 us at
  // m <= 16, 20 at 17, 32-37 at 20-24, 45 / 76 at 32 / 33, 0.28 / 0.40 ms at 48 / 64), while what it saves is a share of the
  // pair kernel (m=16 8 %, 17 6 %, 20-24 3 %, 32-64 9-10 %; m=12 2 %).  Launches with fewer pairs (per shard) than this
  // skip it -- c5 (C=100, m=17: 4 950 pairs, a 0.08 ms kernel) lost 13 % with it, one shard of eight of c3 would lose 3 %.
  // (float64 pairs cost ~2.5x as much, the pass the same: thresholds / 2.5.)
  static constexpr long FACTOR_MIN_PAIRS_F32 = MR_ <= 12 ? 600000 : (MR_ <= 16 ? 100000 : (MR_ <= 17 ? 160000 : (MR_ <= 24 ? 250000 :
                                               (MR_ <= 33 ? 45000 : (MR_ <= 48 ? 40000 : 25000)))));
  static constexpr long FACTOR_MIN_PAIRS = sizeof(T) == 4 ? FACTOR_MIN_PAIRS_F32 : FACTOR_MIN_PAIRS_F32 * 2 / 5;
  static constexpr bool PACK_LINV = MR_ >= 32;
  static constexpr int LINV_ELEMS = PACK_LINV ? MR_ * (MR_ + 1) / 2 : MR_ * MR_;
  // per-wave pitch of that staging area: it doubles as the scratch of transpose_slots
  static constexpr int LI_PITCH = (slot_transpose_cfg<G_, MR_, CPL_>() && LINV_ELEMS < kTransposeElems) ? kTransposeElems : LINV_ELEMS;
  static_assert(G * CPL >= MR, "not enough column slots");
  static_assert(TJ % WAVES == 0, "TJ must be a multiple of the wave count");
  static_assert((TJ & (TJ - 1)) == 0, "TJ must be a power of two (run-time halving, shift-based tile search)");
};

// EIG_BWD selects the backward of the eigenvalues themselves (per-eigenvalue weights EWt) at COMPILE time:
// as a run-time branch it kept six more values alive across the backward phase of the hot instantiation
// (7 spilled VGPRs instead of 1 at m=16, +20 MB of scratch traffic per launch).
template <typename Cfg, bool EIG_BWD>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MIN_WAVES) void pair_tile_kernel(
    const PairParams p, const typename Cfg::type* __restrict__ LT,
    const typename Cfg::type* __restrict__ LinvAll, const typename Cfg::type* __restrict__ Wt,
    const typename Cfg::type* __restrict__ EWt) {
  using T = typename Cfg::type;
  using R = Real<T>;
  constexpr int MR = Cfg::MR, G = Cfg::G, CPL = Cfg::CPL, TI = Cfg::TI;
  constexpr int WAVES = Cfg::WAVES, TRI = Cfg::TRI, TRIP = Cfg::TRIP, NT = Cfg::THREADS;

  __shared__ T s_ga[WAVES * TI * TRIP];  // per-wave private A-side accumulators (lower triangles)
  __shared__ T s_li[WAVES * Cfg::LI_PITCH];  // L_j^-1 of the B class each wave is working on (layout: Cfg::PACK_LINV)
  __shared__ T s_red[WAVES];
  __shared__ int s_redi[2 * WAVES];

  const int tj = p.tj;  // run-time tile width (the host narrows tiles when a launch would not fill the chip)
  // my tile: the blockIdx.x-th tile of this shard (compact grid, see tiles_in_row); scalar search
  int bi = 0, bj = 0;
  {
    int w = blockIdx.x;
    if (p.shard_count == 1) {
      // single shard: every tile of a row is mine; tile widths are powers of two (no divisions)
      const int sh = ilog2(tj);
      for (; bi < p.nbi; ++bi) {
        const int qq = bi * TI + TI - 2;
        int len = p.nbj;
        if (p.self_mode) {
          const int l2 = qq < 0 ? 0 : (qq >> sh) + 1;
          len = l2 < len ? l2 : len;
        }
        if (w < len) {
          bj = w;
          break;
        }
        w -= len;
      }
    } else {
      for (; bi < p.nbi; ++bi) {
        int first;
        const int cnt = shard_tiles_in_row(bi, tiles_in_row(bi, p.nbj, TI, tj, p.self_mode), p.shard_index,
                                           p.shard_count, &first);
        if (w < cnt) {
          bj = first + w * p.shard_count;
          break;
        }
        w -= cnt;
      }
    }
    if (bi >= p.nbi) return;  // cannot happen for a grid sized by launch_pair_tiles
  }
  const int i0 = bi * TI, j0 = bj * tj;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave id in an SGPR
  const int tile = blockIdx.x;  // compact tile number = slab slot

  if (p.want_grad) {
    for (int k = tid; k < WAVES * TI * TRIP; k += NT) s_ga[k] = T(0);
  }
  __syncthreads();

  const T tol2 = R::kEps * R::kEps * T(MR);
  const T scale = param_scale<T>(p), eps = param_eps<T>(p);
  // (float64: all three at m=16 only -- 1.572 -> 1.544 ms; m=17 +0.5 %, m=12 unchanged: off there)
  constexpr int BWD_PRIO = SQFA_BWD_PRIO >= 0 ? SQFA_BWD_PRIO : (sizeof(T) == 4 ? (MR >= 32 ? 7 : 4) : (MR == 16 ? 7 : 0));

  // per-wave partial results, wave-uniform so that they live in SGPRs across the sweep loop
  T loss_acc = T(0);
  int n_nan = 0, n_inf = 0;

  // The sweep loop needs nearly the whole register budget.  Everything that depends on the
  // lane id is therefore re-derived from an opaque copy of threadIdx.x before and after it
  // (a few integer ops) instead of being kept live across it in registers that would spill.
  auto opaque_lane = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));
    return t & 63;
  };

  for (int jj = wave; jj < tj; jj += WAVES) {
    const int j = j0 + jj;  // wave-uniform
    int lane = opaque_lane();
    int g = lane % G;
    int i = i0 + lane / G;
    bool valid = (i < p.nA) && (j < p.nB) && (!p.self_mode || i > j);
    if (!__any(valid)) {
      if (p.want_grad) {  // nothing to add for this B class, but the slab entry must be defined
        T* gbz = static_cast<T*>(p.slab_grad) + ((size_t)tile * (TI + tj) + TI + jj) * TRI;
        for (int k = lane; k < TRI; k += 64) gbz[k] = T(0);
      }
      continue;
    }
    const T* lt = LT + (size_t)(i < p.nA ? i : p.nA - 1) * (MR * MR);
    const int jc = __builtin_amdgcn_readfirstlane(j < p.nB ? j : p.nB - 1);
    constexpr int LE = Cfg::LINV_ELEMS;
    auto li_at = [](int r, int k) constexpr { return Cfg::PACK_LINV ? tri_index(r, k) : r * Cfg::MR + k; };
    T* li = s_li + wave * Cfg::LI_PITCH;
    {
      const T* __restrict__ src = LinvAll + (size_t)jc * LE;
      for (int k = lane; k < LE; k += 64) li[k] = src[k];
    }

    // ---- 1. X = L_j^-1 L_i, my CPL columns ------------------------------------------
    // columns of L_i are loaded straight into x, then multiplied in place by the lower
    // triangular L_j^-1 (rows in descending order never read an overwritten entry)
    T x[CPL][MR];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int col = c * G + g;  // columns dealt round-robin: slot c of lane g is column c*G + g
      const T* src = lt + (size_t)(col < MR ? col : 0) * MR;
      const bool real_col = col < MR;
#pragma unroll
      for (int k = 0; k < MR; ++k) x[c][k] = real_col ? src[k] : T(0);
    }
    if (BWD_PRIO & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int r = MR - 1; r >= 0; --r) {
      T acc[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) acc[c] = T(0);
#pragma unroll
      for (int k = 0; k <= r; ++k) {
        const T l = li[li_at(r, k)];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          // slot c holds column c*G + g >= c*G of the lower triangular L_i: its entries k < c*G are zero in
          // every lane, so those terms are skipped at compile time (m=32 -1.6 %, m=17 -0.6 %).  Not for the
          // float32 m=16 instantiation: no time to gain there (1.076 vs 1.078 ms) and the shorter live ranges
          // make the register allocator spill 7 VGPRs instead of 1 (+30 MB of scratch traffic per launch)
          constexpr bool SKIP_ZEROS = !Cfg::DENSE_FACTOR && !(sizeof(T) == 4 && MR == 16);
          if (!SKIP_ZEROS || k >= c * G) acc[c] = R::fma_(l, x[c][k], acc[c]);
        }
      }
#pragma unroll
      for (int c = 0; c < CPL; ++c) x[c][r] = acc[c];
    }

    if (BWD_PRIO & 1) __builtin_amdgcn_s_setprio(0);
    // ---- 2. one-sided Jacobi ---------------------------------------------------------
    T nrm[CPL], D[CPL];  // true squared norms; squared column scales, x_true = sqrt(D) x (see rot_scaled)
#pragma unroll
    for (int c = 0; c < CPL; ++c) D[c] = T(1);
    int sweeps = 0;
    bool more = true;
    while (more && sweeps < Cfg::MAX_SWEEPS) {
      // D shrinks by cos^2 >= 1/2 per rotation: harmless for the handful of large rotations of
      // a converging run, but a slow pathological pencil could walk it (and 1/x^) out of the
      // float32 range over many sweeps.  Fold the scales back into the columns whenever one
      // leaves [2^-SQFA_RENORM_LOG2, 2^SQFA_RENORM_LOG2] (wave-uniform, practically never).
      {
        bool far = false;
#pragma unroll
        for (int c = 0; c < CPL; ++c) far = far || !(D[c] > R::kScaleLo && D[c] < R::kScaleHi);
        if (__any(far)) {
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            const T dc = R::sqrt_(D[c]);
#pragma unroll
            for (int r = 0; r < MR; ++r) x[c][r] *= dc;
            D[c] = T(1);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        T a = T(0);
#pragma unroll
        for (int r = 0; r < MR; ++r) a = R::fma_(x[c][r], x[c][r], a);
        nrm[c] = a * D[c];
      }
      bool big = false;
      if constexpr (slot_exchange_cfg<G, MR, CPL>()) {
        // travelling columns, see exchange_slots / transpose_slots
        exchange_sweep<T, MR, G, CPL, (MR == G * (CPL - 1) + 1), slot_transpose_cfg<G, MR, CPL>()>(x, nrm, D, tol2, big, li);
      } else {
      // pairs inside my own lane
      if constexpr (SQFA_LOCAL_TOURNAMENT && G == 1) {  // measured: m=8 (one lane per pair) -4 %; no change for G >= 4
        local_rounds<T, MR, CPL, 1>(x, nrm, D, tol2, big);
      } else {
      constexpr int CE = z_visits_cfg<G, MR, CPL>() ? CPL - 1 : CPL;  // the lone column meets them in z_visits
#pragma unroll
      for (int c1 = 0; c1 < CE; ++c1) {
#pragma unroll
        for (int c2 = c1 + 1; c2 < CE; ++c2) {
          const T gh = dot_cols<T, MR>(x[c1], x[c2]);
          T u, ru, k, g2;
          rot_scaled<T, MR>(nrm[c1], nrm[c2], gh, D[c1], D[c2], tol2, T(1), u, ru, k, g2, big);
          const T kgh = k * gh, kg2 = k * g2;
          const T a1 = -(kgh * D[c2]), a2 = kgh * D[c1];
#pragma unroll
          for (int r = 0; r < MR; ++r) {
            const T xp = x[c1][r];
            x[c1][r] = R::fma_(a1, x[c2][r], xp);
            x[c2][r] = R::fma_(a2, xp, x[c2][r]);
          }
          D[c1] *= u;
          D[c2] *= u;
          nrm[c1] -= kg2;
          nrm[c2] += kg2;
        }
      }
      }
      // pairs across the lanes of my group
      if constexpr (G > 1 && G <= Cfg::STATIC_G) {
        cross_rounds_static<T, MR, G, CPL, 1>(x, nrm, D, tol2, big);
      } else if constexpr (G > Cfg::STATIC_G) {
#pragma unroll 1
        for (int s = 1; s < G; ++s) cross_round<T, MR, CPL, 0, 0, (MR == G * (CPL - 1) + 1)>(x, nrm, D, s, tol2, big);
      }
      }
      if constexpr (z_visits_cfg<G, MR, CPL>())
        z_visits<T, MR, G, CPL, swizzled_rows_of_8<T, G, MR, slot_exchange_cfg<G, MR, CPL>()>(), 0>(x, nrm, D, tol2, big);
      more = __any(big);
      ++sweeps;
#ifdef SQFA_ABL_FIXED_SWEEPS  // development: timing ablations run a fixed number of sweeps
      more = sweeps < SQFA_ABL_FIXED_SWEEPS;
#endif
    }
    if constexpr (slot_transpose_cfg<G, MR, CPL>()) {
      if (!SQFA_SLOT_TRANSPOSE_TWICE && (sweeps & 1)) transpose_slots<T, MR, CPL>(x, nrm, D, li);  // back to the dealt positions (wave-uniform branch)
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const T dc = R::sqrt_(D[c]);
#pragma unroll
      for (int r = 0; r < MR; ++r) x[c][r] *= dc;  // back to the true columns
    }
    lane = opaque_lane();
    if constexpr (slot_transpose_cfg<G, MR, CPL>()) {  // the sweeps used the L_j^-1 staging area as scratch: stage it again
      const T* __restrict__ src = LinvAll + (size_t)jc * LE;
      for (int k = lane; k < LE; k += 64) li[k] = src[k];
    }
    g = lane % G;
    i = i0 + lane / G;
    valid = (i < p.nA) && (j < p.nB) && (!p.self_mode || i > j);
    if (p.sweep_counter != nullptr && lane == 0) {
      atomicAdd(&p.sweep_counter[0], (unsigned long long)sweeps);
      atomicAdd(&p.sweep_counter[1], 1ULL);
    }

    // ---- 3. eigenvalues, distance ----------------------------------------------------
    T lam[CPL], loglam[CPL];
    T part = T(0);
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int col = c * G + g;  // columns dealt round-robin: slot c of lane g is column c*G + g
      T a = T(0);
#pragma unroll
      for (int r = 0; r < MR; ++r) a = R::fma_(x[c][r], x[c][r], a);
      const bool real_col = col < p.m;  // identity-padded and empty slots carry no signal
      lam[c] = real_col ? a : T(1);
      loglam[c] = real_col ? R::log_(a) : T(0);
      part = R::fma_(loglam[c], loglam[c], part);
    }
    const T d2 = scale * group_sum<G>(part);
    const T dist = p.sqrt_mode ? R::sqrt_(d2 + eps) : d2;
    const int io = i;
    T w = T(0);
    if (valid) {
      if (Wt != nullptr) {
        w = Wt[(size_t)io * p.nB + j];
        if (p.self_mode) w += Wt[(size_t)j * p.nB + io];
      } else {
        w = param_uniform_weight<T>(p);
      }
    }
    const bool head = valid && g == 0;  // one lane per pair reports
    loss_acc = wave_uniform(loss_acc + wave_sum(head ? w * dist : T(0)));
    n_nan += __popcll(__ballot(head && dist != dist));
    n_inf += __popcll(__ballot(head && dist == dist && !R::finite(dist)));
    if (head) {
      if (p.dist_out != nullptr) {
        T* D = static_cast<T*>(p.dist_out);
        D[(size_t)io * p.nB + j] = dist;
        if (p.self_mode) D[(size_t)j * p.nB + io] = dist;
      }
    }
    if (valid && p.eig_out != nullptr) {
      T* E = static_cast<T*>(p.eig_out);
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int col = c * G + g;  // columns dealt round-robin: slot c of lane g is column c*G + g
        if (col < p.m) {
          E[((size_t)io * p.nB + j) * p.m + col] = lam[c];
          if (p.self_mode) E[((size_t)j * p.nB + io) * p.m + col] = T(1) / lam[c];
        }
      }
    }

    // ---- 4. backward -----------------------------------------------------------------
    if (p.want_grad) {
      const T dd = p.sqrt_mode ? T(0.5) / dist : T(1);
      const T coef = valid ? w * dd * scale * T(2) : T(0);
      T coefA[CPL], coefB[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const T q = coef * loglam[c] / lam[c];
        coefB[c] = -q;
        coefA[c] = q / lam[c];
      }
      if constexpr (EIG_BWD) {
        // backward of the eigenvalues themselves: d lambda_k/dA = u u^T = u~ u~^T / lambda_k,
        // d lambda_k/dB = -lambda_k u u^T = -u~ u~^T  (u~ = sigma_k u).  Self mode also carries the
        // mirrored entry eig[j,i,k] = 1/lambda_k, whose derivative is -1/lambda_k^2.
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const int col = c * G + g;
          T wk = T(0);
          if (valid && col < p.m) {
            wk = EWt[((size_t)io * p.nB + j) * p.m + col];
            if (p.self_mode) wk -= EWt[((size_t)j * p.nB + io) * p.m + col] / (lam[c] * lam[c]);
          }
          coefB[c] = -wk;
          coefA[c] = wk / lam[c];
        }
      }
      // u~ = L_j^-T y in place: u~[r] = sum_{q>=r} Linv[q][r] y[q], rows in ascending order
      if (BWD_PRIO & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int r = 0; r < MR; ++r) {
        T acc[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c] = T(0);
#pragma unroll
        for (int q = r; q < MR; ++q) {
          const T l = li[li_at(q, r)];
#pragma unroll
          for (int c = 0; c < CPL; ++c) acc[c] = R::fma_(l, x[c][q], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) x[c][r] = acc[c];
        // One scheduling barrier per row where it was measured to pay: left alone, the scheduler hoists the strided LDS reads
        // of many rows and the backward phase spills (m=32: 43 VGPRs -> 0 and 6.38 -> 6.29 ms, m=48 154 -> 0 and -1.5 %, m=64
        // 260 -> 19 and -1.3 %, m=33 18 -> 10 and +0.3 %; m=16: 10 -> 0 but 0.686 -> 0.692 ms, m=17 / 24 / float64 within
        // +-0.3 %: float32 from 32 rows on -- profiles/r4_pairs_slot_exchange.txt)
        if constexpr (SQFA_BACK_SCHED_BARRIER > 0 || (SQFA_BACK_SCHED_BARRIER < 0 && sizeof(T) == 4 && MR >= 32)) __builtin_amdgcn_sched_barrier(0);
      }
      if (BWD_PRIO & 2) __builtin_amdgcn_s_setprio(0);
      if (BWD_PRIO & 4) __builtin_amdgcn_s_setprio(1);
      // rank-one sums (lower triangles) with transposing tree reductions:
      //   A side: over the G lanes of the pair; lane g finishes entries idx = G*i + g and adds
      //           them to this wave's private LDS accumulator of class i
      //   B side: over all 64 lanes (every pair of the wave shares j); lane l finishes entries
      //           idx = 64*i + l and stores them to the slab (this wave is the only writer)
#ifndef SQFA_SKIP_OUTER   // development switch: upper bound of what moving the rank-one sums off the VALU could save
      {
        const int lo = lane;
        T* ga = s_ga + (size_t)(wave * TI + lo / G) * TRIP;
        const OuterProduct<T, MR, CPL> prodA{x, coefA};
        constexpr int LG = ilog2(G);
        tree_reduce_blocks<LG, 0, (TRI + G - 1) / G, TRI, T>(prodA, lo, [&](int idx, T v) { ga[idx] += v; });
        T* gb = static_cast<T*>(p.slab_grad) + ((size_t)tile * (TI + tj) + TI + jj) * TRI;
        const OuterProduct<T, MR, CPL> prodB{x, coefB};
        tree_reduce_blocks<6, 0, (TRI + 63) / 64, TRI, T>(prodB, lo, [&](int idx, T v) { gb[idx] = v; });
      }
#else
      {  // keep u~ alive so that the back-transform is not optimised away
        T* gb = static_cast<T*>(p.slab_grad) + ((size_t)tile * (TI + tj) + TI + jj) * TRI;
        T acc = T(0);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
#pragma unroll
          for (int r = 0; r < MR; ++r) acc += x[c][r] * (coefA[c] + coefB[c]);
        }
        if (lane < TRI) gb[lane] = acc;
      }
#endif
      if (BWD_PRIO & 4) __builtin_amdgcn_s_setprio(0);
    }
  }

  // ---- tile epilogue: flush to the slab ------------------------------------------------
  const int tid_end = tid;
  if ((tid_end & 63) == 0) {
    s_red[wave] = loss_acc;
    s_redi[2 * wave] = n_nan;
    s_redi[2 * wave + 1] = n_inf;
  }
  __syncthreads();
  if (tid_end == 0) {
    T l = T(0);
    int nn = 0, ni = 0;
    for (int wv = 0; wv < WAVES; ++wv) {
      l += s_red[wv];
      nn += s_redi[2 * wv];
      ni += s_redi[2 * wv + 1];
    }
    static_cast<T*>(p.slab_loss)[tile] = l;
    p.slab_flag[2 * tile] = nn;
    p.slab_flag[2 * tile + 1] = ni;
  }
  if (p.want_grad) {
    T* slab = static_cast<T*>(p.slab_grad) + (size_t)tile * (TI + tj) * TRI;
    for (int k = tid_end; k < TI * TRI; k += NT) {
      const int pi = k / TRI, idx = k % TRI;
      T acc = T(0);
#pragma unroll
      for (int wv = 0; wv < WAVES; ++wv) acc += s_ga[(size_t)(wv * TI + pi) * TRIP + idx];
      slab[k] = acc;
    }
  }
}

// host-side launcher, instantiated once per configuration in its own translation unit
template <typename Cfg>
hipError_t launch_pair_tiles(const PairParams& p, hipStream_t stream) {
  long n_tiles = 0;
  for (int bi = 0; bi < p.nbi; ++bi) {
    int first;
    n_tiles += shard_tiles_in_row(bi, tiles_in_row(bi, p.nbj, Cfg::TI, p.tj, p.self_mode), p.shard_index,
                                  p.shard_count, &first);
  }
  if (n_tiles == 0) return hipSuccess;  // this shard owns no tile (more shards than tiles)
  dim3 grid((unsigned)n_tiles, 1, 1);
  using T = typename Cfg::type;
  if (p.EW != nullptr)
    hipLaunchKernelGGL((pair_tile_kernel<Cfg, true>), grid, dim3(Cfg::THREADS), 0, stream, p, static_cast<const T*>(p.LT),
                       static_cast<const T*>(p.Linv), static_cast<const T*>(p.W), static_cast<const T*>(p.EW));
  else
    hipLaunchKernelGGL((pair_tile_kernel<Cfg, false>), grid, dim3(Cfg::THREADS), 0, stream, p, static_cast<const T*>(p.LT),
                       static_cast<const T*>(p.Linv), static_cast<const T*>(p.W), static_cast<const T*>(p.EW));
  return hipGetLastError();
}

// ---- K0b: per-class factor pass ------------------------------------------------------------
// One lane group per CLASS (same column layout as the pair kernel: slot c of lane g is column c*G + g): the columns of
// L_i are read from LT, orthogonalised by at most `max_sweeps` sweeps of the pair kernel's own rotations, and written back
// in place.  Any orthogonal V leaves  (L_i V)(L_i V)^T = Sigma_i, so the sweeps need not converge.
//
// The arithmetic is DOUBLE whatever the problem's type (SQFA_FACTOR_F64): the factor is shared by every pair of its class,
// so its rounding errors do not average out over the pairs the way the pair sweeps' own do -- float32 sweeps here
// tripled the float32 gradient error (2.4e-7 -> 7.3e-7 at m=16, 0.9e-6 -> 1.65e-6 at m=32, against the float64 kernels).
#ifndef SQFA_FACTOR_F64
#define SQFA_FACTOR_F64 1
#endif
// The pass has its own lane geometry (LT's layout does not depend on it): a launch is a handful of lone waves whose duration is
// one wave's latency, so a class is spread over MORE lanes than a pair is in the pair kernel (fewer columns per lane =
// fewer dependent instructions per sweep).
template <typename Tio_, int MR_>
struct FactorCfg {
  using io_type = Tio_;
  static constexpr int MR = MR_;
#ifndef SQFA_FACTOR_G
  // measured, C=1000, two double sweeps: m=16 4 lanes (the pair kernel's geometry) 23 us, 8 lanes 14.6, 16 lanes 9.9; m=17 32.5 /
  // 23.5 / 20.0; m=24 8 lanes 42, 16 lanes 37; m=32 8 lanes 173, 16 lanes 72, 32 lanes 45; m=33 227 / 105 / 76; m=48 with 16
  // lanes x 3 slots needs 576 VGPRs of double state and spills (0.9 ms), 32 lanes 0.28 ms; m=64 0.40 ms
  static constexpr int G = MR_ <= 24 ? 16 : 32;
#else
  static constexpr int G = SQFA_FACTOR_G;
#endif
  static constexpr int CPL = (MR_ + G - 1) / G;
  static constexpr int PPW = 64 / G;
};

// At most `max_sweeps` sweeps of the pair kernel's own rotations over the columns a lane group holds (slot c of lane g is
// column c*G + g); stops after the first sweep whose rotations all stay below Real<T>::early2.  Shared by the class factor
// pass (two sweeps, need not converge) and the per-class eigen-decomposition (to convergence).
// MR: the matrix size (decides which slots hold real columns and whether the last one is a lone column); ROWS: the rows a
// column has here -- MR, or 2 MR for the stacked columns of the mean-metric factor pass.
template <typename T, int MR, int G, int CPL, int ROWS = MR>
__device__ __forceinline__ int class_jacobi_sweeps(T (&x)[CPL][ROWS], T (&D)[CPL], int max_sweeps) {
  using R = Real<T>;
  constexpr bool LONE = (MR == G * (CPL - 1) + 1);
  const T tol2 = R::kEps * R::kEps * T(MR);
  T nrm[CPL];
  int sweeps = 0;
  bool more = true;
  while (more && sweeps < max_sweeps) {
    {  // fold the scales back when one leaves the safe range (see the pair kernel's sweep loop): only a long run gets there
      bool far = false;
#pragma unroll
      for (int c = 0; c < CPL; ++c) far = far || !(D[c] > R::kScaleLo && D[c] < R::kScaleHi);
      if (__any(far)) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const T dc = R::sqrt_(D[c]);
#pragma unroll
          for (int r = 0; r < ROWS; ++r) x[c][r] *= dc;
          D[c] = T(1);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      T a = T(0);
#pragma unroll
      for (int r = 0; r < ROWS; ++r) a = R::fma_(x[c][r], x[c][r], a);
      nrm[c] = a * D[c];
    }
    bool big = false;
    if constexpr (SQFA_LOCAL_TOURNAMENT && G == 1) {
      local_rounds<T, ROWS, CPL, 1>(x, nrm, D, tol2, big);
    } else {
      constexpr int CE = z_visits_cfg<G, MR, CPL>() ? CPL - 1 : CPL;
#pragma unroll
      for (int c1 = 0; c1 < CE; ++c1) {
#pragma unroll
        for (int c2 = c1 + 1; c2 < CE; ++c2) {
          const T gh = dot_cols<T, ROWS>(x[c1], x[c2]);
          T u, ru, k, g2;
          rot_scaled<T, ROWS>(nrm[c1], nrm[c2], gh, D[c1], D[c2], tol2, T(1), u, ru, k, g2, big);
          const T kgh = k * gh, kg2 = k * g2;
          const T a1 = -(kgh * D[c2]), a2 = kgh * D[c1];
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            const T xp = x[c1][r];
            x[c1][r] = R::fma_(a1, x[c2][r], xp);
            x[c2][r] = R::fma_(a2, xp, x[c2][r]);
          }
          D[c1] *= u;
          D[c2] *= u;
          nrm[c1] -= kg2;
          nrm[c2] += kg2;
        }
      }
    }
    constexpr int STATIC_G = sizeof(T) == 4 ? 32 : 16;  // PairCfg::STATIC_G's rule for the arithmetic type used HERE
    if constexpr (G > 1 && G <= STATIC_G) {
      cross_rounds_static<T, ROWS, G, CPL, 1, LONE ? 1 : 0>(x, nrm, D, tol2, big);
    } else if constexpr (G > STATIC_G) {
#pragma unroll 1
      for (int s = 1; s < G; ++s) cross_round<T, ROWS, CPL, 0, 0, LONE>(x, nrm, D, s, tol2, big);
    }
    if constexpr (z_visits_cfg<G, MR, CPL>()) z_visits<T, ROWS, G, CPL, swizzled_rows_of_8<T, G, MR>(), 0>(x, nrm, D, tol2, big);
    more = __any(big);
    ++sweeps;
  }
  return sweeps;
}

template <typename Cfg>
__global__ __launch_bounds__(64) void class_factor_kernel(typename Cfg::io_type* __restrict__ LT, int nA, int max_sweeps) {
  using Tio = typename Cfg::io_type;
#if SQFA_FACTOR_F64
  using T = double;
#else
  using T = Tio;
#endif
  using R = Real<T>;
  constexpr int MR = Cfg::MR, G = Cfg::G, CPL = Cfg::CPL, PPW = Cfg::PPW;
  const int lane = threadIdx.x & 63;
  const int g = lane % G;
  const int cls = blockIdx.x * PPW + lane / G;
  Tio* lt = LT + (size_t)(cls < nA ? cls : nA - 1) * (MR * MR);
  T x[CPL][MR];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int col = c * G + g;
    const Tio* src = lt + (size_t)(col < MR ? col : 0) * MR;
    const bool real_col = col < MR;
#pragma unroll
    for (int k = 0; k < MR; ++k) x[c][k] = real_col ? (T)src[k] : T(0);
  }
  T D[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) D[c] = T(1);
  class_jacobi_sweeps<T, MR, G, CPL>(x, D, max_sweeps);
  // (Writing the columns back sorted by norm was tried -- a numpy emulation with a round-robin tournament gained most of a
  // sweep on unrelated ill-conditioned pencils -- and changes nothing under this kernel's pairing order: cond 1e3, m=16 / 32,
  // 6.54 / 7.35 sweeps unsorted, 6.67 / 7.46 ascending, 6.57 / 7.43 descending.  Columns stay where they are.)
  if (cls < nA) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int col = c * G + g;
      if (col < MR) {
        const T dc = R::sqrt_(D[c]);
        Tio* dst = lt + (size_t)col * MR;
#pragma unroll
        for (int r = 0; r < MR; ++r) dst[r] = (Tio)(x[c][r] * dc);
      }
    }
  }
}

// ---- K0b in the metric of the MEAN class (round 4) ---------------------------------------------------------------------
// K1's sweeps start from the Gram matrix F_i^T Sigma_j^-1 F_i.  Orthogonal columns of F_i (the pass above) make it diagonal
// when Sigma_j is a multiple of I -- the best class-level choice for classes that scatter around one, and no help at all for
// classes that share a dominant covariance, Sigma_c = Sbar^1/2 (I + E_c) Sbar^1/2 (real class statistics do): there the
// class-level choice is F_i^T Sbar^-1 F_i diagonal, i.e. the columns of  W_i = Lbar^-1 F_i  orthogonal (Sbar = Lbar Lbar^T
// the mean of the classes).  Emulated (tools/mean_metric_probe.py, sweeps per wave): shared structure 5.8-6.7 -> 5.0-6.1,
// BASELINE generator 5.2 / 5.5 / 6.0 -> 5.1 / 5.1 / 6.0 (m = 16 / 17 / 32).
// The sweeps run on STACKED columns [W; eps L] (2 MR rows): the inner products are those of W up to eps^2 |L|^2 (eps = 2^-30:
// ~1e-18 relative -- these sweeps need not even converge), and the rows of L ride along, so that what is written back is
// L_i V with V EXACTLY the product of the plane rotations applied: (L_i V)(L_i V)^T = Sigma_i to rounding whatever the
// condition of Sbar (two triangular products Lbar (Lbar^-1 L_i V) would lose cond(Lbar) digits).  Double arithmetic.
// mean_linv: Lbar^-1, MR x MR row-major, identity padded (written by the mean block of the Cholesky prologue).
template <typename Cfg>
__global__ __launch_bounds__(64) void class_factor_mean_kernel(typename Cfg::io_type* __restrict__ LT, int nA, int max_sweeps,
                                                               const double* __restrict__ mean_linv) {
  using Tio = typename Cfg::io_type;
  using T = double;
  using R = Real<T>;
  constexpr int MR = Cfg::MR, G = Cfg::G, CPL = Cfg::CPL, PPW = Cfg::PPW, ROWS = 2 * MR;
  constexpr T kStack = T(1.0 / (1 << 30)), kUnstack = T(1 << 30);
  __shared__ T s_li[MR * MR];
  const int lane = threadIdx.x & 63;
  for (int k = lane; k < MR * MR; k += 64) s_li[k] = mean_linv[k];
  __syncthreads();
  const int g = lane % G;
  const int cls = blockIdx.x * PPW + lane / G;
  Tio* lt = LT + (size_t)(cls < nA ? cls : nA - 1) * (MR * MR);
  T x[CPL][ROWS];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int col = c * G + g;
    const Tio* src = lt + (size_t)(col < MR ? col : 0) * MR;
    const bool real_col = col < MR;
#pragma unroll
    for (int k = 0; k < MR; ++k) x[c][MR + k] = real_col ? (T)src[k] : T(0);
    // W = Lbar^-1 l  (lower triangular)
#pragma unroll
    for (int r = 0; r < MR; ++r) {
      T acc = T(0);
#pragma unroll
      for (int k = 0; k <= r; ++k) acc = R::fma_(s_li[r * MR + k], x[c][MR + k], acc);
      x[c][r] = acc;
    }
#pragma unroll
    for (int k = 0; k < MR; ++k) x[c][MR + k] *= kStack;
  }
  T D[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) D[c] = T(1);
  class_jacobi_sweeps<T, MR, G, CPL, ROWS>(x, D, max_sweeps);
  if (cls < nA) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int col = c * G + g;
      if (col < MR) {
        const T dc = R::sqrt_(D[c]) * kUnstack;
        Tio* dst = lt + (size_t)col * MR;
#pragma unroll
        for (int r = 0; r < MR; ++r) dst[r] = (Tio)(x[c][MR + r] * dc);
      }
    }
  }
}

// ---- per-class symmetric eigen-decomposition (spd_log / spd_sqrt of the reference, src/sqfa/linalg.py:121-141,165-183) ----
// The same lane-group Jacobi run to CONVERGENCE on the Cholesky factor of each class: L V = Q Sigma (columns orthogonal),
// so S = L L^T = Q Sigma^2 Q^T -- eigenvalues lambda_k = |y_k|^2 with the RELATIVE accuracy of one-sided Jacobi on a
// factor (what log needs), eigenvectors q_k = y_k / |y_k|.  Always double arithmetic; U (n,m,m) row-major with the
// eigenvectors as COLUMNS and lam (n,m), both double, unsorted.  LT: the (identity-padded, column-contiguous) factors
// written by the Cholesky prologue, read only.
template <typename Cfg>
__global__ __launch_bounds__(64) void class_eig_kernel(const typename Cfg::io_type* __restrict__ LT, int n, int m,
                                                       double* __restrict__ U, double* __restrict__ lam) {
  using Tio = typename Cfg::io_type;
  using T = double;
  using R = Real<T>;
  constexpr int MR = Cfg::MR, G = Cfg::G, CPL = Cfg::CPL, PPW = Cfg::PPW;
  const int lane = threadIdx.x & 63;
  const int g = lane % G;
  const int cls = blockIdx.x * PPW + lane / G;
  const Tio* lt = LT + (size_t)(cls < n ? cls : n - 1) * (MR * MR);
  T x[CPL][MR];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int col = c * G + g;
    const Tio* src = lt + (size_t)(col < MR ? col : 0) * MR;
    const bool real_col = col < MR;
#pragma unroll
    for (int k = 0; k < MR; ++k) x[c][k] = real_col ? (T)src[k] : T(0);
  }
  T D[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) D[c] = T(1);
  class_jacobi_sweeps<T, MR, G, CPL>(x, D, 40);
  if (cls < n) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int col = c * G + g;
      if (col < m) {   // identity-padded columns (col >= m) never mix with the real ones: their inner products are exactly 0
        T a = T(0);
#pragma unroll
        for (int r = 0; r < MR; ++r) a = R::fma_(x[c][r], x[c][r], a);
        const T l = a * D[c];
        lam[(size_t)cls * m + col] = l;
        const T inv = R::sqrt_(D[c]) / R::sqrt_(l);   // NaN for a class that is not positive definite: surfaces downstream
#pragma unroll
        for (int r = 0; r < MR; ++r) {
          if (r < m) U[((size_t)cls * m + r) * m + col] = x[c][r] * inv;
        }
      }
    }
  }
}

template <typename Cfg>
hipError_t launch_class_eig(const void* LT, int n, int m, double* U, double* lam, hipStream_t stream) {
  using T = typename Cfg::type;
  using FC = FactorCfg<T, Cfg::MR>;
  const int blocks = (n + FC::PPW - 1) / FC::PPW;
  hipLaunchKernelGGL((class_eig_kernel<FC>), dim3(blocks), dim3(64), 0, stream, static_cast<const T*>(LT), n, m, U, lam);
  return hipGetLastError();
}

template <typename Cfg>
hipError_t launch_class_factors(const PairParams& p, hipStream_t stream) {
  if constexpr (!Cfg::DENSE_FACTOR) {
    return hipSuccess;
  } else {
    using T = typename Cfg::type;
    using FC = FactorCfg<T, Cfg::MR>;
    const long pairs = p.self_mode ? (long)p.nA * (p.nA - 1) / 2 : (long)p.nA * p.nB;
    if (p.factor_mode < 0 || (p.factor_mode == 0 && pairs / p.shard_count < Cfg::FACTOR_MIN_PAIRS)) return hipSuccess;  // same decision on every shard of a job
    const int blocks = (p.nA + FC::PPW - 1) / FC::PPW;
    if constexpr (Cfg::MEAN_METRIC) {
      if (p.mean_linv != nullptr) {
        hipLaunchKernelGGL((class_factor_mean_kernel<FC>), dim3(blocks), dim3(64), 0, stream,
                           static_cast<T*>(const_cast<void*>(p.LT)), p.nA, Cfg::FACTOR_SWEEPS, p.mean_linv);
        return hipGetLastError();
      }
    }
    hipLaunchKernelGGL((class_factor_kernel<FC>), dim3(blocks), dim3(64), 0, stream,
                       static_cast<T*>(const_cast<void*>(p.LT)), p.nA, Cfg::FACTOR_SWEEPS);
    return hipGetLastError();
  }
}

}  // namespace sqfa
