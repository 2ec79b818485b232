// pair_kernel_2d.hpp -- the pair tile kernel with a 2-D lane layout (round 3).
//
// pair_kernel.hpp gives every lane whole columns of X = L_j^-1 L_i: CPL column slots x MR rows of registers.  For
// MR >= 24 that state (128 VGPRs at m=32 float32, 256 at m=32 float64) leaves two waves -- float64: one -- per SIMD.
// Here a pair is worked on by GC column lanes x 2 row lanes: lane (g, h) holds the rows r = 2q + h, q < MRL = ceil(MR/2),
// of the CPL columns c*GC + g -- half the registers per lane, twice the lanes per pair.  The two row lanes of a column
// are lane and lane ^ RS (RS = 16 or 32), so that one v_permlane{16,32}_swap (a) adds their partial inner products
// (row_total) and (b) converts between this layout and a whole-column layout with 2*GC lanes x ceil(CPL/2) slots, in
// which the X formation, the back-transform and the rank-one sums of pair_kernel.hpp run unchanged:
//
//     whole columns (lane (g,h), slot c2 = column (2 c2 + h) GC + g, all MR rows)
//        -- swap(xf[c2][2q], xf[c2][2q+1]) -->  x[2 c2][q], x[2 c2 + 1][q]      (rows 2q + h of columns 2 c2, 2 c2 + 1 of
//                                                                               column lane g: slot c = column c GC + g,
//                                                                               the numbering of pair_kernel.hpp)
//
// The sweeps are pair_kernel.hpp's (cross_rounds_static / z_visits) on MRL rows with RS set: same tournament over the GC
// column lanes, same scaled rotations; the rotation parameters are evaluated redundantly -- and bit-identically -- by both
// row lanes.  Everything per rotation is therefore executed twice per pair (+20 % instructions at m=32 float32, DESIGN.md
// section 4 "Round 3"); what the layout buys is occupancy.  Measured verdicts per configuration are in configs.hpp.
#pragma once
#include "pair_kernel.hpp"

namespace sqfa {

template <typename T, int MR_, int GC_, int CPL_, int TJ_, int WAVES_, int RS_>
struct PairCfg2D {
  using type = T;
  static constexpr int MR = MR_;        // padded matrix size
  static constexpr int GC = GC_;        // column lanes per pair
  static constexpr int RS = RS_;        // the row partner of a lane is lane ^ RS (16 or 32)
  static constexpr int G = 2 * GC_;     // lanes per pair
  static constexpr int CPL = CPL_;      // column slots per lane in the sweep layout (GC*CPL >= MR)
  static constexpr int CF = (CPL_ + 1) / 2;  // column slots per lane in the whole-column layout
  static constexpr int MRL = (MR_ + 1) / 2;  // rows per lane in the sweep layout
  static constexpr int TJ = TJ_;
  static constexpr int WAVES = WAVES_;
  static constexpr int THREADS = 64 * WAVES_;
  static constexpr int PPW = 64 / G;
  static constexpr int TI = PPW;
  static constexpr int TRI = MR * (MR + 1) / 2;
  static constexpr int TRIP = TRI | 1;
  static constexpr int MAX_SWEEPS = SQFA_MAX_SWEEPS;
  static constexpr bool LONE = (MR_ == GC_ * (CPL_ - 1) + 1);
  static constexpr int XREGS = CPL * MRL * (int)(sizeof(T) / 4);
  static constexpr int FREGS = CF * MR * (int)(sizeof(T) / 4);
  static constexpr int PEAK = XREGS > FREGS ? XREGS : FREGS;
#ifndef SQFA_2D_MIN_WAVES
#define SQFA_2D_MIN_WAVES 0  // 0: from the register estimate
#endif
  // (float64, 64 registers of state: the backward phase needs ~250 VGPRs, two waves per SIMD is what the compiler reaches)
  static constexpr int MIN_WAVES = SQFA_2D_MIN_WAVES > 0 ? SQFA_2D_MIN_WAVES
                                   : (PEAK <= 72 ? (sizeof(T) == 8 ? 2 : 4) : (PEAK <= 110 ? 3 : (PEAK <= 180 ? 2 : 1)));
  // class factor pass (pair_kernel.hpp, K0b): the same policy as the whole-column rows
  // (measured, C=1000, ms per evaluation without -> with: float32 m=40 19.1 -> 17.6; float64 m=24 10.4 -> 10.2, m=32 24.7 -> 23.2,
  // m=33 38.9 -> 36.2; float64 m=48 at C=300 11.0 -> 12.2 -- its X formation without the skipped zeros costs more than the
  // half sweep saved -- so that row keeps triangular factors)
  static constexpr int FACTOR_SWEEPS = SQFA_FACTOR_SWEEPS >= 0 ? SQFA_FACTOR_SWEEPS : ((sizeof(T) == 8 && MR_ >= 48) ? 0 : 2);
  static constexpr bool DENSE_FACTOR = FACTOR_SWEEPS > 0;
  static constexpr bool MEAN_METRIC = SQFA_FACTOR_MEAN && DENSE_FACTOR && (MR_ <= 17 || MR_ == 32);   // as PairCfg::MEAN_METRIC
  static constexpr long FACTOR_MIN_PAIRS_F32 = MR_ <= 33 ? 45000 : (MR_ <= 48 ? 40000 : 25000);
  static constexpr long FACTOR_MIN_PAIRS = sizeof(T) == 4 ? FACTOR_MIN_PAIRS_F32 : FACTOR_MIN_PAIRS_F32 * 2 / 5;
  static constexpr bool PACK_LINV = MR_ >= 32;
  static constexpr int LINV_ELEMS = PACK_LINV ? MR_ * (MR_ + 1) / 2 : MR_ * MR_;
  static constexpr int LGC = ilog2(GC_), HB = ilog2(RS_);
  static_assert(GC * CPL >= MR, "not enough column slots");
  static_assert(RS == 16 || RS == 32, "row partners are one v_permlane16/32_swap apart");
  static_assert(GC <= RS && (GC & (GC - 1)) == 0, "the column lanes sit below the row bit");
  static_assert(TJ % WAVES == 0 && (TJ & (TJ - 1)) == 0, "TJ: a power of two, a multiple of the wave count");
};

// lane-distance of reduction level l and the position of a lane among the 2^levels lanes it reduces with:
//   the pair's lanes first (column bits 1 .. GC/2, then the row bit RS), then the other lane bits in ascending order
template <typename Cfg> struct LaneMap2D {
  static constexpr int dist(int level) {
    if (level < Cfg::LGC) return 1 << level;
    if (level == Cfg::LGC) return Cfg::RS;
    // remaining bits, ascending, skipping the column bits and the row bit
    int seen = Cfg::LGC + 1;
    for (int b = Cfg::LGC; b < 6; ++b) {
      if (b == Cfg::HB) continue;
      if (seen == level) return 1 << b;
      ++seen;
    }
    return 0;
  }
  static __device__ __forceinline__ int pos(int lane, int levels) {
    int p = 0;
    for (int l = 0; l < levels; ++l) p |= ((lane / dist(l)) & 1) << l;
    return p;
  }
};

template <int LEVEL, int BASE, typename Map, typename T, typename P>
__device__ __forceinline__ T tree_reduce_mapped(const P& prod, int lane) {
  if constexpr (LEVEL == 0) {
    return prod.template get<BASE>();
  } else {
    constexpr int H = 1 << (LEVEL - 1);
    constexpr int DIST = Map::dist(LEVEL - 1);
    const T a = tree_reduce_mapped<LEVEL - 1, BASE, Map, T>(prod, lane);
    const T b = tree_reduce_mapped<LEVEL - 1, BASE + H, Map, T>(prod, lane);
    if constexpr (DIST >= 16) {
      return row_swap_sum<DIST>(a, b);
    } else {
      const bool upper = (lane & DIST) != 0;
      const T keep = upper ? b : a;
      const T send = upper ? a : b;
      return keep + xor_fetch<DIST>(send, lane);
    }
  }
}
template <int LEVEL, int I, int N, int TRI, typename Map, typename T, typename P, typename F>
__device__ __forceinline__ void tree_reduce_blocks_mapped(const P& prod, int lane, int pos, F&& sink) {
  if constexpr (I < N) {
    constexpr int W = 1 << LEVEL;
    const T v = tree_reduce_mapped<LEVEL, I * W, Map, T>(prod, lane);
    const int idx = I * W + pos;
    if constexpr ((I + 1) * W <= TRI) {
      sink(idx, v);
    } else {
      if (idx < TRI) sink(idx, v);
    }
    __builtin_amdgcn_sched_barrier(0);
    tree_reduce_blocks_mapped<LEVEL, I + 1, N, TRI, Map, T>(prod, lane, pos, sink);
  }
}

// both operands of a row swap: (a', b') with a' = {own a | partner's b}, b' = {partner's a | own b} for {h = 0 | h = 1}
template <int RS> __device__ __forceinline__ void row_swap_pair(float a, float b, float& ra, float& rb) {
  int o;
  const int k = row_swap_sum_i<RS>(__builtin_bit_cast(int, a), __builtin_bit_cast(int, b), o);
  ra = __builtin_bit_cast(float, k);
  rb = __builtin_bit_cast(float, o);
}
template <int RS> __device__ __forceinline__ void row_swap_pair(double a, double b, double& ra, double& rb) {
  int ohi, olo;
  const int khi = row_swap_sum_i<RS>(__double2hiint(a), __double2hiint(b), ohi);
  const int klo = row_swap_sum_i<RS>(__double2loint(a), __double2loint(b), olo);
  ra = __hiloint2double(khi, klo);
  rb = __hiloint2double(ohi, olo);
}

template <typename Cfg, bool EIG_BWD>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MIN_WAVES) void pair_tile_kernel_2d(
    const PairParams p, const typename Cfg::type* __restrict__ LT,
    const typename Cfg::type* __restrict__ LinvAll, const typename Cfg::type* __restrict__ Wt,
    const typename Cfg::type* __restrict__ EWt) {
  using T = typename Cfg::type;
  using R = Real<T>;
  using Map = LaneMap2D<Cfg>;
  constexpr int MR = Cfg::MR, GC = Cfg::GC, G = Cfg::G, CPL = Cfg::CPL, CF = Cfg::CF, MRL = Cfg::MRL, TI = Cfg::TI, RS = Cfg::RS;
  constexpr int WAVES = Cfg::WAVES, TRI = Cfg::TRI, TRIP = Cfg::TRIP, NT = Cfg::THREADS;
  constexpr int LGC = Cfg::LGC, HB = Cfg::HB, LG = LGC + 1;

  __shared__ T s_ga[WAVES * TI * TRIP];
  __shared__ T s_li[WAVES * Cfg::LINV_ELEMS];
  __shared__ T s_red[WAVES];
  __shared__ int s_redi[2 * WAVES];

  const int tj = p.tj;
  int bi = 0, bj = 0;
  {
    int w = blockIdx.x;
    for (; bi < p.nbi; ++bi) {
      int first;
      const int cnt = shard_tiles_in_row(bi, tiles_in_row(bi, p.nbj, TI, tj, p.self_mode), p.shard_index, p.shard_count, &first);
      if (w < cnt) {
        bj = first + w * p.shard_count;
        break;
      }
      w -= cnt;
    }
    if (bi >= p.nbi) return;
  }
  const int i0 = bi * TI, j0 = bj * tj;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x;

  if (p.want_grad) {
    for (int k = tid; k < WAVES * TI * TRIP; k += NT) s_ga[k] = T(0);
  }
  __syncthreads();

  const T tol2 = R::kEps * R::kEps * T(MR);
  const T scale = param_scale<T>(p), eps = param_eps<T>(p);
  T loss_acc = T(0);
  int n_nan = 0, n_inf = 0;

  auto opaque_lane = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));
    return t & 63;
  };
  // lane -> (column lane g, row lane h, pair of the wave)
  auto lane_g = [](int lane) { return lane & (GC - 1); };
  auto lane_h = [](int lane) { return (lane >> HB) & 1; };
  auto lane_pair = [](int lane) { return ((lane & (RS - 1)) >> LGC) | ((lane >> (HB + 1)) << (HB - LGC)); };

  for (int jj = wave; jj < tj; jj += WAVES) {
    const int j = j0 + jj;
    int lane = opaque_lane();
    int g = lane_g(lane), h = lane_h(lane);
    int i = i0 + lane_pair(lane);
    bool valid = (i < p.nA) && (j < p.nB) && (!p.self_mode || i > j);
    if (!__any(valid)) {
      if (p.want_grad) {
        T* gbz = static_cast<T*>(p.slab_grad) + ((size_t)tile * (TI + tj) + TI + jj) * TRI;
        for (int k = lane; k < TRI; k += 64) gbz[k] = T(0);
      }
      continue;
    }
    const T* lt = LT + (size_t)(i < p.nA ? i : p.nA - 1) * (MR * MR);
    const int jc = __builtin_amdgcn_readfirstlane(j < p.nB ? j : p.nB - 1);
    constexpr int LE = Cfg::LINV_ELEMS;
    auto li_at = [](int r, int k) constexpr { return Cfg::PACK_LINV ? tri_index(r, k) : r * Cfg::MR + k; };
    T* li = s_li + wave * LE;
    {
      const T* __restrict__ src = LinvAll + (size_t)jc * LE;
      for (int k = lane; k < LE; k += 64) li[k] = src[k];
    }

    // ---- 1. X = L_j^-1 L_i in the whole-column layout, then split the rows over the two row lanes ----------
    T x[CPL][MRL];
    {
      T xf[CF][MR];
#pragma unroll
      for (int c2 = 0; c2 < CF; ++c2) {
        const int col = (2 * c2 + h) * GC + g;
        const bool real_col = (2 * c2 + h) < CPL && col < MR;
        const T* src = lt + (size_t)(real_col ? col : 0) * MR;
#pragma unroll
        for (int k = 0; k < MR; ++k) xf[c2][k] = real_col ? src[k] : T(0);
      }
#pragma unroll
      for (int r = MR - 1; r >= 0; --r) {
        T acc[CF];
#pragma unroll
        for (int c2 = 0; c2 < CF; ++c2) acc[c2] = T(0);
#pragma unroll
        for (int k = 0; k <= r; ++k) {
          const T l = li[li_at(r, k)];
#pragma unroll
          for (int c2 = 0; c2 < CF; ++c2) {
            // column (2 c2 + h) GC + g >= 2 c2 GC of the lower triangular L_i: entries k < 2 c2 GC vanish in every lane
            // (only without the class factor pass: its factors are dense)
            if (Cfg::DENSE_FACTOR || k >= 2 * c2 * GC) acc[c2] = R::fma_(l, xf[c2][k], acc[c2]);
          }
        }
#pragma unroll
        for (int c2 = 0; c2 < CF; ++c2) xf[c2][r] = acc[c2];
      }
#pragma unroll
      for (int c2 = 0; c2 < CF; ++c2) {
#pragma unroll
        for (int q = 0; q < MRL; ++q) {
          T ra, rb;
          row_swap_pair<RS>(xf[c2][2 * q], (2 * q + 1 < MR) ? xf[c2][2 * q + 1] : T(0), ra, rb);
          x[2 * c2][q] = ra;
          if (2 * c2 + 1 < CPL) x[(2 * c2 + 1 < CPL) ? 2 * c2 + 1 : 0][q] = rb;   // (odd CPL: the last slot has no partner slot)
        }
      }
    }

    // ---- 2. one-sided Jacobi on the columns, rows split over (lane, lane ^ RS) ---------------------------
    T nrm[CPL], D[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) D[c] = T(1);
    int sweeps = 0;
    bool more = true;
    while (more && sweeps < Cfg::MAX_SWEEPS) {
      {
        bool far = false;
#pragma unroll
        for (int c = 0; c < CPL; ++c) far = far || !(D[c] > R::kScaleLo && D[c] < R::kScaleHi);
        if (__any(far)) {
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            const T dc = R::sqrt_(D[c]);
#pragma unroll
            for (int r = 0; r < MRL; ++r) x[c][r] *= dc;
            D[c] = T(1);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        T a = T(0);
#pragma unroll
        for (int r = 0; r < MRL; ++r) a = R::fma_(x[c][r], x[c][r], a);
        nrm[c] = row_total<RS>(a) * D[c];
      }
      bool big = false;
      constexpr int CE = (Cfg::LONE && SQFA_Z_VISITS) ? CPL - 1 : CPL;
      if constexpr (slot_exchange_ok<GC, CE>()) {  // travelling columns (pair_kernel.hpp, exchange_slots)
        exchange_sweep<T, MRL, GC, CPL, Cfg::LONE, false, RS>(x, nrm, D, tol2, big);
      } else {
#pragma unroll
      for (int c1 = 0; c1 < CE; ++c1) {
#pragma unroll
        for (int c2 = c1 + 1; c2 < CE; ++c2) {
          const T gh = row_total<RS>(dot_cols<T, MRL>(x[c1], x[c2]));
          T u, ru, k, g2;
          rot_scaled<T, MRL>(nrm[c1], nrm[c2], gh, D[c1], D[c2], tol2, T(1), u, ru, k, g2, big);
          const T kgh = k * gh, kg2 = k * g2;
          const T a1 = -(kgh * D[c2]), a2 = kgh * D[c1];
#pragma unroll
          for (int r = 0; r < MRL; ++r) {
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
      if constexpr (GC > 1) cross_rounds_static<T, MRL, GC, CPL, 1, Cfg::LONE ? 1 : 0, RS>(x, nrm, D, tol2, big);
      }
      if constexpr (Cfg::LONE && SQFA_Z_VISITS && GC > 1)
        z_visits<T, MRL, GC, CPL, swizzled_rows_of_8<T, GC, MRL>(), 0, RS>(x, nrm, D, tol2, big);
      more = __any(big);
      ++sweeps;
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const T dc = R::sqrt_(D[c]);
#pragma unroll
      for (int r = 0; r < MRL; ++r) x[c][r] *= dc;
    }
    lane = opaque_lane();
    g = lane_g(lane);
    h = lane_h(lane);
    i = i0 + lane_pair(lane);
    valid = (i < p.nA) && (j < p.nB) && (!p.self_mode || i > j);
    if (p.sweep_counter != nullptr && lane == 0) {
      atomicAdd(&p.sweep_counter[0], (unsigned long long)sweeps);
      atomicAdd(&p.sweep_counter[1], 1ULL);
    }

    // ---- 3. eigenvalues, distance (both row lanes hold identical lam) ----------------------------------
    T lam[CPL], loglam[CPL];
    T part = T(0);
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int col = c * GC + g;
      T a = T(0);
#pragma unroll
      for (int r = 0; r < MRL; ++r) a = R::fma_(x[c][r], x[c][r], a);
      a = row_total<RS>(a);
      const bool real_col = col < p.m;
      lam[c] = real_col ? a : T(1);
      loglam[c] = real_col ? R::log_(a) : T(0);
      part = R::fma_(loglam[c], loglam[c], part);
    }
    const T d2 = scale * group_sum<GC>(part);
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
    const bool head = valid && g == 0 && h == 0;
    loss_acc = wave_uniform(loss_acc + wave_sum(head ? w * dist : T(0)));
    n_nan += __popcll(__ballot(head && dist != dist));
    n_inf += __popcll(__ballot(head && dist == dist && !R::finite(dist)));
    if (head && p.dist_out != nullptr) {
      T* Dm = static_cast<T*>(p.dist_out);
      Dm[(size_t)io * p.nB + j] = dist;
      if (p.self_mode) Dm[(size_t)j * p.nB + io] = dist;
    }
    if (valid && h == 0 && p.eig_out != nullptr) {
      T* E = static_cast<T*>(p.eig_out);
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int col = c * GC + g;
        if (col < p.m) {
          E[((size_t)io * p.nB + j) * p.m + col] = lam[c];
          if (p.self_mode) E[((size_t)j * p.nB + io) * p.m + col] = T(1) / lam[c];
        }
      }
    }

    // ---- 4. backward: back to whole columns, then pair_kernel.hpp's back-transform and rank-one sums ----
    if (p.want_grad) {
      const T dd = p.sqrt_mode ? T(0.5) / dist : T(1);
      const T coef = valid ? w * dd * scale * T(2) : T(0);
      T coefA[CPL], coefB[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const T qq = coef * loglam[c] / lam[c];
        coefB[c] = -qq;
        coefA[c] = qq / lam[c];
      }
      if constexpr (EIG_BWD) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const int col = c * GC + g;
          T wk = T(0);
          if (valid && col < p.m) {
            wk = EWt[((size_t)io * p.nB + j) * p.m + col];
            if (p.self_mode) wk -= EWt[((size_t)j * p.nB + io) * p.m + col] / (lam[c] * lam[c]);
          }
          coefB[c] = -wk;
          coefA[c] = wk / lam[c];
        }
      }
      T xf[CF][MR];
      T cAf[CF], cBf[CF];
#pragma unroll
      for (int c2 = 0; c2 < CF; ++c2) {
        const bool has_odd = 2 * c2 + 1 < CPL;
        cAf[c2] = h ? (has_odd ? coefA[has_odd ? 2 * c2 + 1 : 0] : T(0)) : coefA[2 * c2];
        cBf[c2] = h ? (has_odd ? coefB[has_odd ? 2 * c2 + 1 : 0] : T(0)) : coefB[2 * c2];
#pragma unroll
        for (int q = 0; q < MRL; ++q) {
          T ra, rb;
          row_swap_pair<RS>(x[2 * c2][q], has_odd ? x[has_odd ? 2 * c2 + 1 : 0][q] : T(0), ra, rb);
          xf[c2][2 * q] = ra;
          if (2 * q + 1 < MR) xf[c2][(2 * q + 1 < MR) ? 2 * q + 1 : 0] = rb;
        }
      }
      // u~ = L_j^-T y in place, rows in ascending order
#pragma unroll
      for (int r = 0; r < MR; ++r) {
        T acc[CF];
#pragma unroll
        for (int c2 = 0; c2 < CF; ++c2) acc[c2] = T(0);
#pragma unroll
        for (int q = r; q < MR; ++q) {
          const T l = li[li_at(q, r)];
#pragma unroll
          for (int c2 = 0; c2 < CF; ++c2) acc[c2] = R::fma_(l, xf[c2][q], acc[c2]);
        }
#pragma unroll
        for (int c2 = 0; c2 < CF; ++c2) xf[c2][r] = acc[c2];
      }
      {
        const int lo = lane;
        T* ga = s_ga + (size_t)(wave * TI + lane_pair(lo)) * TRIP;
        const OuterProduct<T, MR, CF> prodA{xf, cAf};
        const int posA = Map::pos(lo, LG);
        tree_reduce_blocks_mapped<LG, 0, (TRI + G - 1) / G, TRI, Map, T>(prodA, lo, posA, [&](int idx, T v) { ga[idx] += v; });
        T* gb = static_cast<T*>(p.slab_grad) + ((size_t)tile * (TI + tj) + TI + jj) * TRI;
        const OuterProduct<T, MR, CF> prodB{xf, cBf};
        const int posB = Map::pos(lo, 6);
        tree_reduce_blocks_mapped<6, 0, (TRI + 63) / 64, TRI, Map, T>(prodB, lo, posB, [&](int idx, T v) { gb[idx] = v; });
      }
    }
  }

  // ---- tile epilogue: flush to the slab (as pair_kernel.hpp) --------------------------------------------
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

template <typename Cfg>
hipError_t launch_pair_tiles_2d(const PairParams& p, hipStream_t stream) {
  long n_tiles = 0;
  for (int bi = 0; bi < p.nbi; ++bi) {
    int first;
    n_tiles += shard_tiles_in_row(bi, tiles_in_row(bi, p.nbj, Cfg::TI, p.tj, p.self_mode), p.shard_index, p.shard_count, &first);
  }
  if (n_tiles == 0) return hipSuccess;
  dim3 grid((unsigned)n_tiles, 1, 1);
  using T = typename Cfg::type;
  if (p.EW != nullptr)
    hipLaunchKernelGGL((pair_tile_kernel_2d<Cfg, true>), grid, dim3(Cfg::THREADS), 0, stream, p, static_cast<const T*>(p.LT),
                       static_cast<const T*>(p.Linv), static_cast<const T*>(p.W), static_cast<const T*>(p.EW));
  else
    hipLaunchKernelGGL((pair_tile_kernel_2d<Cfg, false>), grid, dim3(Cfg::THREADS), 0, stream, p, static_cast<const T*>(p.LT),
                       static_cast<const T*>(p.Linv), static_cast<const T*>(p.W), static_cast<const T*>(p.EW));
  return hipGetLastError();
}

}  // namespace sqfa
