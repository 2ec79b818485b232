// sqfa_api.hip -- C ABI (include/sqfa_hip.h), per-class Cholesky prologue and slab reduction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/sqfa_hip.h"
#include "configs.hpp"
#include "pair_kernel.hpp"
#include "pair_kernel_2d.hpp"

namespace sqfa {

// ---- per-configuration launchers (defined in pair_inst.hip translation units) -----------
#define SQFA_DECL_F32(T, MR, G, CPL, TJ, WV) \
  hipError_t launch_pair_f32_##MR(const PairParams&, hipStream_t);   \
  hipError_t launch_factor_f32_##MR(const PairParams&, hipStream_t); \
  hipError_t launch_classeig_f32_##MR(const void*, int, int, double*, double*, hipStream_t);
#define SQFA_DECL_F64(T, MR, G, CPL, TJ, WV) \
  hipError_t launch_pair_f64_##MR(const PairParams&, hipStream_t);   \
  hipError_t launch_factor_f64_##MR(const PairParams&, hipStream_t); \
  hipError_t launch_classeig_f64_##MR(const void*, int, int, double*, double*, hipStream_t);
SQFA_CONFIGS_F32(SQFA_DECL_F32)
SQFA_CONFIGS_F64(SQFA_DECL_F64)
#define SQFA_DECL_F32S(T, MR, G, CPL, TJ, WV) \
  hipError_t launch_pair_f32s_##MR(const PairParams&, hipStream_t);  \
  hipError_t launch_factor_f32s_##MR(const PairParams&, hipStream_t);
SQFA_CONFIGS_F32_SMALL(SQFA_DECL_F32S)
#define SQFA_DECL_F64S(T, MR, G, CPL, TJ, WV) \
  hipError_t launch_pair_f64s_##MR(const PairParams&, hipStream_t);  \
  hipError_t launch_factor_f64s_##MR(const PairParams&, hipStream_t);
SQFA_CONFIGS_F64_SMALL(SQFA_DECL_F64S)

#define SQFA_DECL2D_F32(T, MR, GC, CPL, TJ, WV, RS) \
  hipError_t launch_pair2d_f32_##MR(const PairParams&, hipStream_t); \
  hipError_t launch_factor2d_f32_##MR(const PairParams&, hipStream_t); \
  hipError_t launch_classeig2d_f32_##MR(const void*, int, int, double*, double*, hipStream_t);
#define SQFA_DECL2D_F64(T, MR, GC, CPL, TJ, WV, RS) \
  hipError_t launch_pair2d_f64_##MR(const PairParams&, hipStream_t); \
  hipError_t launch_factor2d_f64_##MR(const PairParams&, hipStream_t); \
  hipError_t launch_classeig2d_f64_##MR(const void*, int, int, double*, double*, hipStream_t);
SQFA_CONFIGS2D_F32(SQFA_DECL2D_F32)
SQFA_CONFIGS2D_F64(SQFA_DECL2D_F64)

struct Geometry {
  int MR, G, CPL, TJ, TI, WV;  // TJ: widest tile (B classes); a launch may use TJ/2, TJ/4 ... >= WV
  hipError_t (*launch)(const PairParams&, hipStream_t);
  hipError_t (*factor)(const PairParams&, hipStream_t);  // K0b, the class factor pass
  hipError_t (*eig)(const void*, int, int, double*, double*, hipStream_t) = nullptr;  // per-class eigen-decomposition (regular rows)
  bool mean_metric = false;     // the row's factor pass runs in the metric of the mean class (Cfg::MEAN_METRIC)
  long factor_min_pairs = 0;    // Cfg::FACTOR_MIN_PAIRS: launches with fewer pairs per shard skip the factor pass
};

// The geometry table: every whole-column row (pair_kernel.hpp) and every 2-D row (pair_kernel_2d.hpp: GC column lanes x 2
// row lanes per pair, G = 2 GC lanes per pair) of configs.hpp; a problem of size m runs on the smallest MR >= m.  A launch
// with few pairs (`pairs` = pairs per shard; < 0: not known, regular rows only) takes the small-launch row of that MR if
// there is one (configs.hpp, SQFA_CONFIGS_F32_SMALL).  geometry_mode: sqfa_airm_options::geometry_policy of the call (0 by
// pair count, 1 small-launch rows wherever one exists, -1 never) -- a per-call argument, no process-wide state.
static bool find_geometry(int m, int dtype, long pairs, Geometry* out, int geometry_mode = 0) {
  bool found = false;
  Geometry best{};
  auto consider = [&](int dt, const Geometry& g) {
    if (dt == dtype && m <= g.MR && (!found || g.MR < best.MR)) {
      best = g;
      found = true;
    }
  };
#define SQFA_ROW_F32(T, MR_, G_, CPL_, TJ_, WV_) consider(SQFA_F32, Geometry{MR_, G_, CPL_, TJ_, 64 / G_, WV_, launch_pair_f32_##MR_, launch_factor_f32_##MR_, launch_classeig_f32_##MR_, PairCfg<float, MR_, G_, CPL_, TJ_, WV_>::MEAN_METRIC, PairCfg<float, MR_, G_, CPL_, TJ_, WV_>::FACTOR_MIN_PAIRS});
#define SQFA_ROW_F64(T, MR_, G_, CPL_, TJ_, WV_) consider(SQFA_F64, Geometry{MR_, G_, CPL_, TJ_, 64 / G_, WV_, launch_pair_f64_##MR_, launch_factor_f64_##MR_, launch_classeig_f64_##MR_, PairCfg<double, MR_, G_, CPL_, TJ_, WV_>::MEAN_METRIC, PairCfg<double, MR_, G_, CPL_, TJ_, WV_>::FACTOR_MIN_PAIRS});
#define SQFA_ROW2D_F32(T, MR_, GC_, CPL_, TJ_, WV_, RS_) \
  consider(SQFA_F32, Geometry{MR_, 2 * GC_, CPL_, TJ_, 64 / (2 * GC_), WV_, launch_pair2d_f32_##MR_, launch_factor2d_f32_##MR_, launch_classeig2d_f32_##MR_, PairCfg2D<float, MR_, GC_, CPL_, TJ_, WV_, RS_>::MEAN_METRIC, PairCfg2D<float, MR_, GC_, CPL_, TJ_, WV_, RS_>::FACTOR_MIN_PAIRS});
#define SQFA_ROW2D_F64(T, MR_, GC_, CPL_, TJ_, WV_, RS_) \
  consider(SQFA_F64, Geometry{MR_, 2 * GC_, CPL_, TJ_, 64 / (2 * GC_), WV_, launch_pair2d_f64_##MR_, launch_factor2d_f64_##MR_, launch_classeig2d_f64_##MR_, PairCfg2D<double, MR_, GC_, CPL_, TJ_, WV_, RS_>::MEAN_METRIC, PairCfg2D<double, MR_, GC_, CPL_, TJ_, WV_, RS_>::FACTOR_MIN_PAIRS});
  SQFA_CONFIGS_F32(SQFA_ROW_F32)
  SQFA_CONFIGS_F64(SQFA_ROW_F64)
  SQFA_CONFIGS2D_F32(SQFA_ROW2D_F32)
  SQFA_CONFIGS2D_F64(SQFA_ROW2D_F64)
  if (found && geometry_mode >= 0 && pairs >= 0) {
    // same padded size, more lanes per pair
#define SQFA_ROW_F32S(T, MR_, G_, CPL_, TJ_, WV_)                                                                        \
    if (dtype == SQFA_F32 && best.MR == MR_ && (geometry_mode > 0 || pairs < small_launch_max_pairs(MR_)))                \
      best = Geometry{MR_, G_, CPL_, TJ_, 64 / G_, WV_, launch_pair_f32s_##MR_, launch_factor_f32s_##MR_, best.eig, PairCfg<float, MR_, G_, CPL_, TJ_, WV_>::MEAN_METRIC, PairCfg<float, MR_, G_, CPL_, TJ_, WV_>::FACTOR_MIN_PAIRS};
    SQFA_CONFIGS_F32_SMALL(SQFA_ROW_F32S)
#define SQFA_ROW_F64S(T, MR_, G_, CPL_, TJ_, WV_)                                                                        \
    if (dtype == SQFA_F64 && best.MR == MR_ && (geometry_mode > 0 || pairs < small_launch_max_pairs_f64(MR_)))              \
      best = Geometry{MR_, G_, CPL_, TJ_, 64 / G_, WV_, launch_pair_f64s_##MR_, launch_factor_f64s_##MR_, best.eig, PairCfg<double, MR_, G_, CPL_, TJ_, WV_>::MEAN_METRIC, PairCfg<double, MR_, G_, CPL_, TJ_, WV_>::FACTOR_MIN_PAIRS};
    SQFA_CONFIGS_F64_SMALL(SQFA_ROW_F64S)
  }
  if (found) *out = best;
  return found;
}
static long pair_count(int nA, int nB, int shard_count) {  // pairs per shard of a call (nB == 0: self mode)
  const long p = nB == 0 ? (long)nA * (nA - 1) / 2 : (long)nA * nB;
  return p / (shard_count > 0 ? shard_count : 1);
}

static int max_dim() {
  int mx = 0;
#define SQFA_MAX(T, MR_, G_, CPL_, TJ_, WV_) if (MR_ > mx) mx = MR_;
#define SQFA_MAX2D(T, MR_, GC_, CPL_, TJ_, WV_, RS_) if (MR_ > mx) mx = MR_;
  SQFA_CONFIGS_F32(SQFA_MAX)
  SQFA_CONFIGS2D_F32(SQFA_MAX2D)
  return mx;
}

static thread_local char g_last_error[256] = "";

// optional per-launch timing of the pair tile kernel with HIP events on the caller's stream
struct EventPair { hipEvent_t a, b; };
static std::atomic<bool> g_profile{false};
static std::vector<EventPair> g_events;
static std::mutex g_events_mutex;  // the event lists are shared by every host thread that calls into the library
}  // namespace sqfa
// shared with project_kernel.hip
bool sqfa_profile_enabled() { return sqfa::g_profile.load(); }
std::vector<std::pair<hipEvent_t, hipEvent_t>>& sqfa_project_events() {
  static std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  return ev;
}
std::mutex& sqfa_project_events_mutex() {
  static std::mutex m;
  return m;
}
namespace sqfa {

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct WorkspaceLayout {
  size_t off_lt, off_linv, off_slab, off_loss, off_flag, off_rows, off_mean, total;
};
constexpr int kMeanParts = 32;   // class groups of the mean-class partial sums (mean_partial_kernel)

// Tiles of a shard for tile width tj (same enumeration as the kernel's compact grid).
static long shard_tiles(int nA, int nBeff, const Geometry& g, int tj, int self_mode, int shard_index, int shard_count) {
  const int nbi = (nA + g.TI - 1) / g.TI, nbj = (nBeff + tj - 1) / tj;
  long n = 0;
  for (int bi = 0; bi < nbi; ++bi) {
    int first;
    n += shard_tiles_in_row(bi, tiles_in_row(bi, nbj, g.TI, tj, self_mode), shard_index, shard_count, &first);
  }
  return n;
}

// Workgroups the chip holds at once (4 per CU for the 256-thread configurations); a launch with
// fewer tiles than this runs for one tile's latency however few they are, so tiles are narrowed.
static int resident_workgroups() {
  static int cached = 0;
  if (cached == 0) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cached = 4 * (cus > 0 ? cus : 256);
  }
  return cached;
}

// Tile widths a launch may use: the configuration's TJ, and its halvings (down to the wave
// count) as long as the job then has at most 16 x the resident workgroups in tiles -- narrowing
// only ever helps launches that do not fill the chip, and the bound keeps the slab small.
static bool width_allowed(int nA, int nBeff, const Geometry& g, int tj, int self_mode) {
  if (tj == g.TJ) return true;
  if (tj < g.WV || tj < 1 || g.TJ % tj != 0) return false;
  return shard_tiles(nA, nBeff, g, tj, self_mode, 0, 1) <= 16L * resident_workgroups();
}

// ... while the launch has fewer than SQFA_TILE_ROUNDS x the resident workgroups in tiles.  2 since round 3: a launch of
// just over one round of workgroups (m <= 8 at C=1000: ~1000 tiles of 64 x 8 pairs) ends with half the chip idle for
// one tile's duration; measured C=1000, m=8 0.218 -> 0.202 ms, m=4 likewise; sizes with >= 2 rounds are unaffected
// (m=16 1.050 vs 1.041-1.048 ms with the narrowest tiles: within noise, and they double the A-side slab).
#ifndef SQFA_TILE_ROUNDS
#define SQFA_TILE_ROUNDS 2
#endif
// Tile width a call with `shard_count` shards uses: halved while a shard's launch would leave workgroup
// slots empty.  Decided from the TOTAL tile count and shard_count only, so that every shard of a job
// picks the same tiling (tile ownership (bi + bj) % shard_count is defined on that tiling).
static int choose_tile_width(int nA, int nBeff, const Geometry& g, int self_mode, int shard_count) {
  int tj = g.TJ;
  while (tj % 2 == 0 && width_allowed(nA, nBeff, g, tj / 2, self_mode) &&
         shard_tiles(nA, nBeff, g, tj, self_mode, 0, 1) / shard_count < (long)SQFA_TILE_ROUNDS * resident_workgroups())
    tj /= 2;
  return tj;
}

// Workspace for one tiling (only_tj > 0: exactly the tile width a call will use), or for the narrowest
// tiles any call may choose (only_tj == 0: most tiles, largest slab).
static WorkspaceLayout layout(int nA, int nBeff, const Geometry& g, size_t esz, int self_mode, int only_tj = 0,
                              int shard_count = 1) {
  WorkspaceLayout w;
  const size_t mat = (size_t)g.MR * g.MR * esz;
  const size_t tri = (size_t)g.MR * (g.MR + 1) / 2;
  const size_t nbi = (nA + g.TI - 1) / g.TI;
  size_t slab = 0, tiles = 0;
  for (int tj = g.TJ; tj >= 1 && width_allowed(nA, nBeff, g, tj, self_mode); tj /= 2) {
    if (only_tj > 0 && tj != only_tj) {
      if (tj % 2) break;
      continue;
    }
    // the slab holds the tiles one shard owns (compact numbering): the largest shard decides
    size_t owned = 0;
    for (int r = 0; r < shard_count; ++r)
      owned = std::max(owned, (size_t)shard_tiles(nA, nBeff, g, tj, self_mode, r, shard_count));
    slab = std::max(slab, owned * (size_t)(g.TI + tj) * tri * esz);
    tiles = std::max(tiles, owned);
    if (tj % 2) break;
  }
  size_t o = 0;
  w.off_lt = o;   o = align_up(o + (size_t)nA * mat);
  w.off_linv = o; o = align_up(o + (size_t)nBeff * mat);  // (packed for MR >= 32: uses about half)
  w.off_slab = o; o = align_up(o + slab);
  w.off_loss = o; o = align_up(o + tiles * esz);
  w.off_flag = o; o = align_up(o + tiles * 2 * sizeof(int));
  w.off_rows = o; o = align_up(o + (nbi + 1) * sizeof(int));
  // mean-metric factor pass: kMeanParts partial sums of the A classes (MR x MR doubles each), then Lbar^-1 (MR x MR doubles)
  w.off_mean = o; o = align_up(o + (size_t)(kMeanParts + 1) * g.MR * g.MR * sizeof(double));
  w.total = o;
  return w;
}

// ---- K0: per-class Cholesky factor and its inverse (always evaluated in double) ----------
// One 256-thread workgroup per class, matrix in LDS.  LT[c][col*MR + k] = L[k][col];
// Linv[c][r*MR + k] = (L^-1)[r][k] (MR < 32) or packed Linv[c][r(r+1)/2 + k], k <= r (MR >= 32).
// Both are padded to MR x MR with an identity block.
// A non-SPD input produces NaNs, which surface as non-finite distances (nonfinite_out),
// never as a fault.
//   Cholesky: right-looking, all (r,c) entries of the trailing block updated in parallel
//   per pivot column (one LDS round trip per entry and pivot instead of a serial row loop).
//   Inverse: X = L^-1 row by row; row r needs rows < r, columns are independent; the inner
//   sum runs over k in parallel chunks of 4 lanes per column.
// 1/sqrt(x) in double without the IEEE sqrt/divide sequences: hardware estimate + 2 Newton steps
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}

// Partial sums of the class matrices for the mean class of the mean-metric factor pass: block b adds the classes b, b + P,
// b + 2P, ... (P = gridDim.x) entry by entry in that order -- fixed association, bitwise reproducible.  out[b][m*m] doubles.
template <typename T>
__global__ __launch_bounds__(256) void mean_partial_kernel(const T* __restrict__ S, int n, int m, double* __restrict__ out) {
  const int b = blockIdx.x, P = gridDim.x;
  for (int e = threadIdx.x; e < m * m; e += 256) {
    double a0 = 0.0, a1 = 0.0;
    int c = b;
    for (; c + P < n; c += 2 * P) {
      a0 += (double)S[(size_t)c * m * m + e];
      a1 += (double)S[(size_t)(c + P) * m * m + e];
    }
    if (c < n) a0 += (double)S[(size_t)c * m * m + e];
    out[(size_t)b * m * m + e] = a0 + a1;
  }
}

// mean_parts != nullptr: ONE extra block (blockIdx.x == n_classes) factorises the mean of the classes (the sum of the
// n_parts partial sums / n_classes) and writes Lbar^-1 as MR x MR row-major doubles, identity padded, to mean_linv.
template <typename T, int MAXM>
__global__ __launch_bounds__(256) void cholesky_kernel(const T* __restrict__ S, int m, int MR,
                                                       T* __restrict__ LT, T* __restrict__ Linv,
                                                       int* __restrict__ row_start, PairParams pp, int TI,
                                                       int n_classes = 0, const double* __restrict__ mean_parts = nullptr,
                                                       int n_parts = 0, double* __restrict__ mean_linv = nullptr) {
  if (row_start != nullptr && blockIdx.x == 0) {
    // slab slot table for K2: owned tiles before each block-row, in the compact grid's order.  The per-row
    // counts (integer divisions) are evaluated by 256 threads at once, thread 0 only adds them up: a serial
    // loop here put 9 us on the critical path of the whole prologue (block 0 finished last).
    __shared__ int s_cnt[256];
    int base = 0;
    for (int b0 = 0; b0 < pp.nbi; b0 += 256) {
      const int bi = b0 + (int)threadIdx.x;
      int cnt = 0, first;
      if (bi < pp.nbi)
        cnt = shard_tiles_in_row(bi, tiles_in_row(bi, pp.nbj, TI, pp.tj, pp.self_mode), pp.shard_index, pp.shard_count, &first);
      s_cnt[threadIdx.x] = cnt;
      __syncthreads();
      if (threadIdx.x == 0) {
        int acc = base;
        for (int k = 0; k < 256 && b0 + k < pp.nbi; ++k) {
          row_start[b0 + k] = acc;
          acc += s_cnt[k];
        }
        s_cnt[0] = acc;
      }
      __syncthreads();
      base = s_cnt[0];
      __syncthreads();
    }
    if (threadIdx.x == 0) row_start[pp.nbi] = base;
  }
  // LDS sized for the padded size class (MAXM >= m) so that small problems keep many
  // workgroups per CU resident
  __shared__ double a[MAXM][MAXM + 1];
  __shared__ double b[MAXM][MAXM + 1];
  __shared__ double rd[MAXM];  // 1 / L[k][k]
  const int c = blockIdx.x, t = threadIdx.x;
  const bool mean_block = mean_parts != nullptr && c == n_classes;
  if (mean_block) {
    const double inv_n = 1.0 / (double)n_classes;
    for (int idx = t; idx < m * m; idx += 256) {
      double acc = 0.0;
      for (int q = 0; q < n_parts; ++q) acc += mean_parts[(size_t)q * m * m + idx];
      a[idx / m][idx % m] = acc * inv_n;
      b[idx / m][idx % m] = 0.0;
    }
  } else {
    const T* s = S + (size_t)c * m * m;
    for (int idx = t; idx < m * m; idx += 256) {
      a[idx / m][idx % m] = (double)s[idx];
      b[idx / m][idx % m] = 0.0;
    }
  }
  __syncthreads();
#ifndef SQFA_CHOL_ONE_BARRIER
#define SQFA_CHOL_ONE_BARRIER 1
#endif
#if SQFA_CHOL_ONE_BARRIER
  // Right-looking elimination on the UNSCALED columns, one workgroup barrier per pivot (round 3; three before: pivot, scaled
  // column, trailing update): step k only reads column k and the pivot, which no later step writes, and subtracts
  // a[r][k] a[c][k] / a[k][k] from the trailing block; the columns are scaled by 1/sqrt(pivot) once at the end.
  for (int k = 0; k < m; ++k) {
    const double akk = a[k][k];
    // a non-positive or NaN pivot poisons the trailing block: NaN here and everywhere downstream, as with the scaled form
    double rk = __builtin_amdgcn_rcp(akk);   // hardware estimate + two Newton steps instead of the IEEE divide sequence
    rk = rk * (2.0 - akk * rk);
    rk = rk * (2.0 - akk * rk);
    if (!(akk > 0.0)) rk = __builtin_nan("");
    const int n = m - k - 1;
    for (int e = t; e < n * n; e += 256) {
      const int r = k + 1 + e / n, c2 = k + 1 + e % n;
      if (c2 <= r) a[r][c2] -= a[r][k] * a[c2][k] * rk;
    }
    __syncthreads();
  }
  for (int k = t; k < m; k += 256) rd[k] = fast_rsqrt(a[k][k]);
  __syncthreads();
  for (int e = t; e < m * m; e += 256) {
    const int r = e / m, k = e % m;
    if (k <= r) a[r][k] *= rd[k];  // r == k: akk * rs = sqrt(akk)
  }
#else
  for (int k = 0; k < m; ++k) {
    // a non-positive or NaN pivot gives NaN here and everywhere downstream
    const double rs = fast_rsqrt(a[k][k]);
    __syncthreads();
    for (int r = k + t; r < m; r += 256) {
      a[r][k] *= rs;  // r == k: akk * rs = sqrt(akk)
      if (r == k) rd[k] = rs;
    }
    __syncthreads();
    // trailing update of the lower triangle: entries (r, c2) with k < c2 <= r < m
    const int n = m - k - 1;
    for (int e = t; e < n * n; e += 256) {
      const int r = k + 1 + e / n, c2 = k + 1 + e % n;
      if (c2 <= r) a[r][c2] -= a[r][k] * a[c2][k];
    }
  }
#endif
  __syncthreads();
  // inverse X = L^-1, row by row; 4 lanes per column (always inside one wave, so rows only
  // need the wave's own program order, no workgroup barrier)
  {
    const int col = t >> 2, part = t & 3;
    for (int r = 0; r < m; ++r) {
      double acc = 0.0;
      if (col < r) {
        for (int k = col + part; k < r; k += 4) acc += a[r][k] * b[k][col];
      }
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      if (part == 0 && col <= r && col < m) b[r][col] = ((col == r ? 1.0 : 0.0) - acc) * rd[r];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  if (mean_block) {
    for (int idx = t; idx < MR * MR; idx += 256) {
      const int r = idx / MR, k = idx % MR;
      mean_linv[idx] = (r < m && k < m) ? (k <= r ? b[r][k] : 0.0) : (r == k ? 1.0 : 0.0);
    }
    return;
  }
  const size_t base = (size_t)c * MR * MR;
  for (int idx = t; idx < MR * MR; idx += 256) {
    const int r = idx / MR, k = idx % MR;
    if (LT != nullptr) {  // here r = column of L, k = row of L
      double v = (r < m && k < m) ? (k >= r ? a[k][r] : 0.0) : (r == k ? 1.0 : 0.0);
      LT[base + idx] = (T)v;
    }
    if (Linv != nullptr) {
      double v = (r < m && k < m) ? (k <= r ? b[r][k] : 0.0) : (r == k ? 1.0 : 0.0);
      if (MR >= 32) {  // packed lower triangle (PairCfg::PACK_LINV): less LDS per wave in the pair kernel
        if (k <= r) Linv[(size_t)c * (MR * (MR + 1) / 2) + tri_index(r, k)] = (T)v;
      } else {
        Linv[base + idx] = (T)v;
      }
    }
  }
}

// ---- K2: fixed-order reduction of the tile slabs ------------------------------------------
__device__ inline bool tile_processed(const PairParams& p, int bi, int bj, int TI, int TJ) {
  if (p.shard_count > 1 && (bi + bj) % p.shard_count != p.shard_index) return false;
  if (p.self_mode && (bi * TI + TI - 1 <= bj * TJ)) return false;
  return true;
}

#ifndef SQFA_K2_THREADS
#define SQFA_K2_THREADS 512  // 1024 threads (7 summation groups per class instead of 3): 29 vs 28 us at c3, no gain
#endif
template <typename T>
__global__ __launch_bounds__(SQFA_K2_THREADS) void finalize_kernel(const PairParams p, int TI, int TJ, int MR,
                                                       T* __restrict__ gradA, T* __restrict__ gradB,
                                                       T* __restrict__ loss_out, int* __restrict__ nonfinite_out) {
  const int tid = threadIdx.x;
  const int TRI = MR * (MR + 1) / 2;
  const int n_cls = p.nA + (p.self_mode ? 0 : p.nB);
  const int b = blockIdx.x;
  __shared__ T s_part[SQFA_K2_THREADS];
  // slab slot table (written by the Cholesky prologue) staged in LDS: read from global memory inside the
  // summation loops it put a second dependent L2 round trip in front of every slab load
  constexpr int ROWS_LDS = 2048;
  __shared__ int s_rows[ROWS_LDS];
  const bool rows_staged = p.nbi + 1 <= ROWS_LDS;
  if (rows_staged) {
    for (int k = tid; k <= p.nbi; k += SQFA_K2_THREADS) s_rows[k] = p.row_start[k];
  }
  __syncthreads();
  auto row_start_of = [&](int bi) { return rows_staged ? s_rows[bi] : p.row_start[bi]; };
  // K2 geometry: a class is split over BPC workgroups of EPB consecutive lower-triangle entries each; inside a
  // workgroup NG = THREADS / EPB thread groups each sum every NG-th contributing tile (a fixed subsequence),
  // then the NG partial sums are combined in group order: short dependent chains, bitwise reproducible.
  // (One workgroup per class with THREADS / TRI groups left m >= 32 -- 528 entries -- with ONE group walking
  // ~190 tiles serially: 275 us at C=1000, m=32.)
  // Small sizes (TRI <= 256, m <= 22) keep one workgroup per class (splitting 136 entries over two
  // workgroups cost 41 vs 30 us at m=16); m=32: 289 -> 199 us.
  const int EPB = TRI <= 256 ? TRI : 128, NG = SQFA_K2_THREADS / EPB;
  const int BPC = (TRI + EPB - 1) / EPB;
  if (b < n_cls * BPC) {
    if (!p.want_grad) return;
    const int cls = b / BPC, e_in = tid % EPB, idx = (b % BPC) * EPB + e_in, grp = tid / EPB;
    const bool a_side = cls < p.nA;
    const int c = a_side ? cls : cls - p.nA;
    T* out = a_side ? gradA : gradB;
    if (out == nullptr) return;
    const T* slab = static_cast<const T*>(p.slab_grad);
    const size_t tile_stride = (size_t)(TI + TJ) * TRI;
    const int bi_a = c / TI, pi = c % TI, bj_b = c / TJ, pj = c % TJ;
    // Only the tiles this shard owns are visited (same enumeration as the pair kernel's grid):
    //   as A class: tiles (bi_a, first_a + k N), k < n_a   (row bi_a; self mode: up to the diagonal)
    //   as B class: tiles (first_b + k N, bj_b), k < n_b   (column bj_b; self mode: checked per tile)
    const int N = p.shard_count;
    int first_a = 0, n_a = 0, first_b = 0, n_b = 0;
    if (a_side) n_a = shard_tiles_in_row(bi_a, tiles_in_row(bi_a, p.nbj, TI, TJ, p.self_mode), p.shard_index, N, &first_a);
    if (!a_side || p.self_mode) {
      first_b = ((p.shard_index - bj_b) % N + N) % N;
      n_b = p.nbi > first_b ? (p.nbi - 1 - first_b) / N + 1 : 0;
    }
    T acc = T(0);
    if (idx < TRI && grp < NG) {
      // group `grp` sums every NG-th A-side tile and every NG-th B-side tile, four independent partial sums each:
      // the loads of a thread do not depend on each other, and one accumulator made the ~60 of them one chain
      // (c3: 30 -> 17 us, m=32: 202 -> 120 us).  Fixed association order: reproducible.  (Round 2 kept a single chain
      // for float64 so that the chaotic c5 trajectory golden stayed matched; round 3 pins filter parity at well-posed
      // points instead -- goldens G6c / G7e -- and float64 takes the same order as float32.)
      T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
      {
        // the q-th owned tile of block-row bi_a (bj = first_a + q N) sits in slab slot row_start + q
        const T* base = slab + (size_t)row_start_of(bi_a) * tile_stride + (size_t)pi * TRI + idx;
        int q = grp;
        for (; q + 3 * NG < n_a; q += 4 * NG) {
          a0 += base[(size_t)q * tile_stride];
          a1 += base[(size_t)(q + NG) * tile_stride];
          a2 += base[(size_t)(q + 2 * NG) * tile_stride];
          a3 += base[(size_t)(q + 3 * NG) * tile_stride];
        }
        for (; q < n_a; q += NG) a0 += base[(size_t)q * tile_stride];
      }
      {
        auto b_tile = [&](int k) -> T {
          const int bi = first_b + k * N;
          if (!tile_processed(p, bi, bj_b, TI, TJ)) return T(0);
          int slot = row_start_of(bi) + bj_b;  // single shard: every tile of the row is owned
          if (N > 1) {
            const int first_in_row = ((p.shard_index - bi) % N + N) % N;
            slot = row_start_of(bi) + (bj_b - first_in_row) / N;
          }
          return slab[(size_t)slot * tile_stride + (size_t)(TI + pj) * TRI + idx];
        };
        int k = grp;
        for (; k + 3 * NG < n_b; k += 4 * NG) {
          a0 += b_tile(k);
          a1 += b_tile(k + NG);
          a2 += b_tile(k + 2 * NG);
          a3 += b_tile(k + 3 * NG);
        }
        for (; k < n_b; k += NG) a0 += b_tile(k);
      }
      acc = (a0 + a1) + (a2 + a3);
    }
    s_part[tid] = acc;   // [grp][entry]: tid = grp * EPB + e_in
    __syncthreads();
    if (grp == 0 && idx < TRI) {
      T tot = T(0);
      for (int g2 = 0; g2 < NG; ++g2) tot += s_part[g2 * EPB + e_in];
      int r = 0;
      while ((r + 1) * (r + 2) / 2 <= idx) ++r;
      const int cc = idx - r * (r + 1) / 2;
      if (r < p.m && cc < p.m) {
        out[(size_t)c * p.m * p.m + (size_t)r * p.m + cc] = tot;
        out[(size_t)c * p.m * p.m + (size_t)cc * p.m + r] = tot;
      }
    }
    return;
  }
  // last block: loss, flag, diagonals
  __shared__ double s_l[SQFA_K2_THREADS];
  __shared__ int s_f[SQFA_K2_THREADS];
  __shared__ int s_f2[SQFA_K2_THREADS];
  double l = 0.0;
  int f = 0, f2 = 0;
  const int ntiles = row_start_of(p.nbi);   // every slab slot belongs to a tile this shard processed
  for (int tix = tid; tix < ntiles; tix += SQFA_K2_THREADS) {
    l += (double)static_cast<const T*>(p.slab_loss)[tix];
    f += p.slab_flag[2 * tix];
    f2 += p.slab_flag[2 * tix + 1];
  }
  s_l[tid] = l;
  s_f[tid] = f;
  s_f2[tid] = f2;
  __syncthreads();
  for (int st = SQFA_K2_THREADS / 2; st > 0; st >>= 1) {
    if (tid < st) {
      s_l[tid] += s_l[tid + st];
      s_f[tid] += s_f[tid + st];
      s_f2[tid] += s_f2[tid + st];
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (loss_out != nullptr) loss_out[0] = (T)s_l[0];
    if (nonfinite_out != nullptr) {
      nonfinite_out[0] = s_f[0];
      nonfinite_out[1] = s_f2[0];
    }
  }
  if (p.self_mode) {
    if (p.dist_out != nullptr) {
      T* D = static_cast<T*>(p.dist_out);
      const T dv = p.sqrt_mode ? (T)sqrt(p.eps) : T(0);
      for (int c = tid; c < p.nA; c += SQFA_K2_THREADS) D[(size_t)c * p.nB + c] = dv;
    }
    if (p.eig_out != nullptr) {
      T* E = static_cast<T*>(p.eig_out);
      for (int k = tid; k < p.nA * p.m; k += SQFA_K2_THREADS) {
        const int c = k / p.m, q = k % p.m;
        E[((size_t)c * p.nB + c) * p.m + q] = T(1);
      }
    }
  }
}

// ---- per-class matrix functions f(S) = Q f(Lambda) Q^T and their backward -----------------------------------------------
// (spd_log / spd_sqrt of the reference, src/sqfa/linalg.py:121-141, 165-183: torch.linalg.eigh + einsum there.)
// One 256-thread workgroup per class; Q (m x m, eigenvectors as columns) and the small intermediates live in LDS as
// double whatever the problem's type; m <= 64.
__device__ __forceinline__ double spd_fn(int kind, double l) {
  return kind == SQFA_SPD_LOG ? log(l) : (kind == SQFA_SPD_SQRT ? sqrt(l) : 1.0 / sqrt(l));
}
// divided difference (f(a) - f(b)) / (a - b), f'(a) on the diagonal -- in forms that stay accurate when a ~ b
// (torch's eigh backward divides by a - b: inf / NaN for repeated eigenvalues; this is its limit)
__device__ __forceinline__ double spd_fn_dd(int kind, double a, double b) {
  if (kind == SQFA_SPD_SQRT) return 1.0 / (sqrt(a) + sqrt(b));
  if (kind == SQFA_SPD_INV_SQRT) {
    const double ra = sqrt(a), rb = sqrt(b);
    return -1.0 / (ra * rb * (ra + rb));
  }
  const double r = (a - b) / b;                     // log: log1p(r) / (r b)
  if (fabs(r) < 1e-8) return (1.0 - 0.5 * r) / b;
  return log1p(r) / (r * b);
}

template <typename T>
__global__ __launch_bounds__(256) void spd_function_kernel(const double* __restrict__ U, const double* __restrict__ lam, int m,
                                                           int kind, T* __restrict__ F) {
  extern __shared__ double sh[];   // q[m][m], f[m]
  double* q = sh;
  double* f = sh + m * m;
  const int c = blockIdx.x, t = threadIdx.x;
  for (int k = t; k < m * m; k += 256) q[k] = U[(size_t)c * m * m + k];
  for (int k = t; k < m; k += 256) f[k] = spd_fn(kind, lam[(size_t)c * m + k]);
  __syncthreads();
  for (int e = t; e < m * m; e += 256) {
    const int r = e / m, cc = e % m;
    if (cc > r) continue;
    double acc = 0.0;
    for (int k = 0; k < m; ++k) acc += f[k] * q[r * m + k] * q[cc * m + k];
    F[(size_t)c * m * m + r * m + cc] = (T)acc;
    F[(size_t)c * m * m + cc * m + r] = (T)acc;   // exactly symmetric
  }
}

// gradS = Q [ (Q^T sym(G) Q) o Gamma ] Q^T,  Gamma_kl = divided difference of f at (lambda_k, lambda_l)  (Daleckii-Krein)
template <typename T>
__global__ __launch_bounds__(256) void spd_function_backward_kernel(const double* __restrict__ U, const double* __restrict__ lam,
                                                                    const T* __restrict__ G, int m, int kind, T* __restrict__ gradS) {
  extern __shared__ double sh[];   // q[m][m], a[m][m], b[m][m], l[m]
  double* q = sh;
  double* a = sh + m * m;
  double* b = sh + 2 * m * m;
  double* l = sh + 3 * m * m;
  const int c = blockIdx.x, t = threadIdx.x;
  for (int k = t; k < m * m; k += 256) q[k] = U[(size_t)c * m * m + k];
  for (int k = t; k < m; k += 256) l[k] = lam[(size_t)c * m + k];
  for (int e = t; e < m * m; e += 256) {
    const int r = e / m, cc = e % m;
    a[e] = 0.5 * ((double)G[(size_t)c * m * m + r * m + cc] + (double)G[(size_t)c * m * m + cc * m + r]);
  }
  __syncthreads();
  for (int e = t; e < m * m; e += 256) {           // b = sym(G) Q
    const int r = e / m, k = e % m;
    double acc = 0.0;
    for (int j = 0; j < m; ++j) acc += a[r * m + j] * q[j * m + k];
    b[e] = acc;
  }
  __syncthreads();
  for (int e = t; e < m * m; e += 256) {           // a = (Q^T b) o Gamma
    const int k = e / m, k2 = e % m;
    double acc = 0.0;
    for (int j = 0; j < m; ++j) acc += q[j * m + k] * b[j * m + k2];
    a[e] = acc * spd_fn_dd(kind, l[k], l[k2]);
  }
  __syncthreads();
  for (int e = t; e < m * m; e += 256) {           // b = Q a
    const int r = e / m, k2 = e % m;
    double acc = 0.0;
    for (int k = 0; k < m; ++k) acc += q[r * m + k] * a[k * m + k2];
    b[e] = acc;
  }
  __syncthreads();
  for (int e = t; e < m * m; e += 256) {           // gradS = b Q^T  (lower triangle, mirrored: exactly symmetric)
    const int r = e / m, cc = e % m;
    if (cc > r) continue;
    double acc = 0.0;
    for (int k2 = 0; k2 < m; ++k2) acc += b[r * m + k2] * q[cc * m + k2];
    gradS[(size_t)c * m * m + r * m + cc] = (T)acc;
    gradS[(size_t)c * m * m + cc * m + r] = (T)acc;
  }
}

static int fail(int code, const char* what, hipError_t e) {
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, e == hipSuccess ? "" : hipGetErrorString(e));
  return code;
}

}  // namespace sqfa

using namespace sqfa;

extern "C" {

int sqfa_hip_version(void) { return 1000; }
const char* sqfa_hip_arch(void) { return "gfx950"; }
int sqfa_hip_max_dim(void) { return max_dim(); }
const char* sqfa_hip_last_error(void) { return g_last_error; }

int sqfa_airm_profile(int enable) {
  g_profile.store(enable != 0);
  return SQFA_OK;
}

int sqfa_airm_profile_read(double* tile_kernel_ms_total, int* launches) {
  double total = 0.0;
  int n = 0;
  std::lock_guard<std::mutex> lock(g_events_mutex);
  for (auto& ev : g_events) {
    float ms = 0.f;
    if (hipEventSynchronize(ev.b) == hipSuccess && hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
      total += ms;
      ++n;
    }
    (void)hipEventDestroy(ev.a);
    (void)hipEventDestroy(ev.b);
  }
  g_events.clear();
  if (tile_kernel_ms_total) *tile_kernel_ms_total = total;
  if (launches) *launches = n;
  return SQFA_OK;
}

int sqfa_project_profile_read(double* kernel_ms_total, int* launches) {
  double total = 0.0;
  int n = 0;
  std::lock_guard<std::mutex> lock(sqfa_project_events_mutex());
  for (auto& ev : sqfa_project_events()) {
    float ms = 0.f;
    if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
      total += ms;
      ++n;
    }
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  sqfa_project_events().clear();
  if (kernel_ms_total) *kernel_ms_total = total;
  if (launches) *launches = n;
  return SQFA_OK;
}

int sqfa_airm_tiling(int nA, int nB, int m, int dtype, int* tile_i, int* tile_j, int* n_tiles_i,
                     int* n_tiles_j, int* padded_m) {
  if (nA < 1 || nB < 0 || m < 1 || (dtype != SQFA_F32 && dtype != SQFA_F64)) return SQFA_ERR_BAD_ARGUMENT;
  Geometry g;
  if (!find_geometry(m, dtype, pair_count(nA, nB, 1), &g)) return SQFA_ERR_UNSUPPORTED_M;  // the geometry of an unsharded call
  const int nBeff = nB == 0 ? nA : nB;
  if (tile_i) *tile_i = g.TI;
  if (tile_j) *tile_j = g.TJ;
  if (n_tiles_i) *n_tiles_i = (nA + g.TI - 1) / g.TI;
  if (n_tiles_j) *n_tiles_j = (nBeff + g.TJ - 1) / g.TJ;
  if (padded_m) *padded_m = g.MR;
  return SQFA_OK;
}

size_t sqfa_airm_workspace_bytes(int nA, int nB, int m, int dtype) {
  int ti, tj, nbi, nbj, mr;
  if (sqfa_airm_tiling(nA, nB, m, dtype, &ti, &tj, &nbi, &nbj, &mr) != SQFA_OK) return 0;
  const int nBeff = nB == 0 ? nA : nB;
  // enough for any shard count: the regular row's and the small-launch row's layouts both fit
  Geometry g;
  find_geometry(m, dtype, -1, &g, -1);   // the regular row ...
  size_t need = layout(nA, nBeff, g, dtype == SQFA_F32 ? 4 : 8, nB == 0 ? 1 : 0).total;
  find_geometry(m, dtype, 0, &g, 1);     // ... and the small-launch row of the size, whatever policy a call will carry
  need = std::max(need, layout(nA, nBeff, g, dtype == SQFA_F32 ? 4 : 8, nB == 0 ? 1 : 0).total);
  return need;
}

static int pairwise_impl(const void* A, int nA, const void* B, int nB, int m, int dtype, double scale,
                         double eps, int sqrt_mode, const void* pair_weights, double uniform_weight,
                         int shard_index, int shard_count, void* loss_out, void* gradA_out,
                         void* gradB_out, void* dist_out, void* eig_out, int* nonfinite_out,
                         void* workspace, size_t workspace_bytes, void* stream_, const void* eig_weights,
                         const sqfa_airm_options* options) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  auto clamp = [](int v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); };
  const int geometry_mode = options ? clamp(options->geometry_policy) : 0;
  const int factor_mode = options ? clamp(options->class_factor_policy) : 0;
  const int mean_mode = options ? clamp(options->mean_metric_policy) : 0;
  g_last_error[0] = 0;
  if (A == nullptr || nA < 1 || m < 1 || nB < 0 || workspace == nullptr) return fail(SQFA_ERR_BAD_ARGUMENT, "null/size argument", hipSuccess);
  if (dtype != SQFA_F32 && dtype != SQFA_F64) return fail(SQFA_ERR_BAD_ARGUMENT, "dtype", hipSuccess);
  if ((B == nullptr) != (nB == 0)) return fail(SQFA_ERR_BAD_ARGUMENT, "B and nB disagree", hipSuccess);
  if (shard_count < 1 || shard_index < 0 || shard_index >= shard_count) return fail(SQFA_ERR_BAD_ARGUMENT, "shard", hipSuccess);
  const bool self_mode = (B == nullptr);
  if (self_mode && nA < 2) return fail(SQFA_ERR_BAD_ARGUMENT, "self mode needs at least two classes", hipSuccess);
  Geometry g;
  if (!find_geometry(m, dtype, pair_count(nA, nB, shard_count), &g, geometry_mode)) return fail(SQFA_ERR_UNSUPPORTED_M, "matrix size not supported", hipSuccess);
  const size_t esz = dtype == SQFA_F32 ? 4 : 8;
  const int nBeff = self_mode ? nA : nB;
  // tile width: halve while a shard's launch would leave workgroup slots empty.  Decided from
  // the TOTAL tile count and shard_count only, so that every shard of a job picks the same
  // tiling (tile ownership (bi + bj) % shard_count is defined on that tiling).
  const int tj = choose_tile_width(nA, nBeff, g, self_mode ? 1 : 0, shard_count);
  const int nbi = (nA + g.TI - 1) / g.TI, nbj = (nBeff + tj - 1) / tj;
  const WorkspaceLayout w = layout(nA, nBeff, g, esz, self_mode ? 1 : 0, tj, shard_count);
  if (workspace_bytes < w.total) return fail(SQFA_ERR_WORKSPACE, "workspace too small", hipSuccess);
  char* ws = static_cast<char*>(workspace);

  PairParams p;
  memset(&p, 0, sizeof(p));
  p.LT = ws + w.off_lt;
  p.Linv = ws + w.off_linv;
  p.W = pair_weights;
  p.EW = eig_weights;
  p.slab_grad = ws + w.off_slab;
  p.slab_loss = ws + w.off_loss;
  p.slab_flag = reinterpret_cast<int*>(ws + w.off_flag);
  p.row_start = reinterpret_cast<int*>(ws + w.off_rows);
  p.dist_out = dist_out;
  p.eig_out = eig_out;
  p.sweep_counter = options ? options->sweep_counter : nullptr;
  p.nA = nA;
  p.nB = nBeff;
  p.m = m;
  p.self_mode = self_mode ? 1 : 0;
  p.sqrt_mode = sqrt_mode ? 1 : 0;
  p.want_grad = gradA_out != nullptr ? 1 : 0;
  p.shard_index = shard_index;
  p.shard_count = shard_count;
  p.nbi = nbi;
  p.nbj = nbj;
  p.tj = tj;
  p.factor_mode = factor_mode;
  p.scale = scale;
  p.eps = eps;
  p.uniform_weight = uniform_weight;
  p.scale_f = (float)scale;
  p.eps_f = (float)eps;
  p.uniform_weight_f = (float)uniform_weight;

  // K0: factors
  // mean-metric factor pass (class_factor_mean_kernel): will K0b run, and on a size that has it?  Same rule as
  // launch_class_factors (the decision depends on (nA, nB, shard count, options) only: every shard of a job decides alike).
  const long pairs_per_shard = pair_count(nA, nB, shard_count);
  const bool want_mean = g.mean_metric && mean_mode > 0 && factor_mode >= 0 &&
                         (factor_mode > 0 || pairs_per_shard >= g.factor_min_pairs) && nA >= 2;
  double* mean_parts = reinterpret_cast<double*>(ws + w.off_mean);
  double* mean_linv = mean_parts + (size_t)kMeanParts * g.MR * g.MR;
  const int n_parts = nA < kMeanParts ? nA : kMeanParts;
  if (want_mean) {
    if (dtype == SQFA_F32) hipLaunchKernelGGL(mean_partial_kernel<float>, dim3(n_parts), dim3(256), 0, stream, static_cast<const float*>(A), nA, m, mean_parts);
    else hipLaunchKernelGGL(mean_partial_kernel<double>, dim3(n_parts), dim3(256), 0, stream, static_cast<const double*>(A), nA, m, mean_parts);
    p.mean_linv = mean_linv;
  }
  bool rows_done = false;
  auto launch_chol = [&](auto zero, const void* src, int n, void* lt, void* li) {
    using T = decltype(zero);
    const T* sp = static_cast<const T*>(src);
    T* ltp = static_cast<T*>(lt);
    T* lip = static_cast<T*>(li);
    int* rows = rows_done ? nullptr : p.row_start;  // the first prologue launch also writes the slab slot table
    const bool with_mean = want_mean && src == A && !rows_done;   // the A-side launch carries the mean block
    rows_done = true;
    const int blocks = n + (with_mean ? 1 : 0);
    const double* mp = with_mean ? mean_parts : nullptr;
    if (m <= 16) hipLaunchKernelGGL((cholesky_kernel<T, 16>), dim3(blocks), dim3(256), 0, stream, sp, m, g.MR, ltp, lip, rows, p, g.TI, n, mp, n_parts, mean_linv);
    else if (m <= 32) hipLaunchKernelGGL((cholesky_kernel<T, 32>), dim3(blocks), dim3(256), 0, stream, sp, m, g.MR, ltp, lip, rows, p, g.TI, n, mp, n_parts, mean_linv);
    else hipLaunchKernelGGL((cholesky_kernel<T, 64>), dim3(blocks), dim3(256), 0, stream, sp, m, g.MR, ltp, lip, rows, p, g.TI, n, mp, n_parts, mean_linv);
  };
  void* ws_lt = ws + w.off_lt;
  void* ws_li = ws + w.off_linv;
  if (dtype == SQFA_F32) {
    if (self_mode) {
      launch_chol(0.0f, A, nA, ws_lt, ws_li);
    } else {
      launch_chol(0.0f, A, nA, ws_lt, nullptr);
      launch_chol(0.0f, B, nB, nullptr, ws_li);
    }
  } else {
    if (self_mode) {
      launch_chol(0.0, A, nA, ws_lt, ws_li);
    } else {
      launch_chol(0.0, A, nA, ws_lt, nullptr);
      launch_chol(0.0, B, nB, nullptr, ws_li);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SQFA_ERR_LAUNCH, "cholesky_kernel", e);
  if (g.factor != nullptr) {  // K0b: orthogonalise the columns of each A-side factor (same stream: after the Cholesky launches)
    e = g.factor(p, stream);
    if (e != hipSuccess) return fail(SQFA_ERR_LAUNCH, "class_factor_kernel", e);
  }

  // K1: pair tiles
  EventPair ev{};
  bool prof = g_profile.load();
  if (prof) {  // event records do not belong in a captured graph: profile eager launches only
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) prof = false;
  }
  if (prof) {
    (void)hipEventCreate(&ev.a);
    (void)hipEventCreate(&ev.b);
    (void)hipEventRecord(ev.a, stream);
  }
  e = g.launch(p, stream);
  if (prof) {
    (void)hipEventRecord(ev.b, stream);
    std::lock_guard<std::mutex> lock(g_events_mutex);
    g_events.push_back(ev);
  }
  if (e != hipSuccess) return fail(SQFA_ERR_LAUNCH, "pair_tile_kernel", e);

  // K2: slab reduction
  const int n_cls = nA + (self_mode ? 0 : nB);
  const int k2_tri = g.MR * (g.MR + 1) / 2;
  const int k2_bpc = k2_tri <= 256 ? 1 : (k2_tri + 127) / 128;  // workgroups per class (finalize_kernel's EPB)
  if (dtype == SQFA_F32) {
    hipLaunchKernelGGL(finalize_kernel<float>, dim3(n_cls * k2_bpc + 1), dim3(SQFA_K2_THREADS), 0, stream, p, g.TI, tj, g.MR,
                       static_cast<float*>(gradA_out), static_cast<float*>(gradB_out),
                       static_cast<float*>(loss_out), nonfinite_out);
  } else {
    hipLaunchKernelGGL(finalize_kernel<double>, dim3(n_cls * k2_bpc + 1), dim3(SQFA_K2_THREADS), 0, stream, p, g.TI, tj, g.MR,
                       static_cast<double*>(gradA_out), static_cast<double*>(gradB_out),
                       static_cast<double*>(loss_out), nonfinite_out);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return fail(SQFA_ERR_LAUNCH, "finalize_kernel", e);
  return SQFA_OK;
}

size_t sqfa_airm_workspace_bytes_sharded(int nA, int nB, int m, int dtype, int shard_count, int geometry_policy) {
  int ti, tj, nbi, nbj, mr;
  if (shard_count < 1 || sqfa_airm_tiling(nA, nB, m, dtype, &ti, &tj, &nbi, &nbj, &mr) != SQFA_OK) return 0;
  Geometry g;
  find_geometry(m, dtype, pair_count(nA, nB, shard_count), &g, geometry_policy > 0 ? 1 : (geometry_policy < 0 ? -1 : 0));
  const int nBeff = nB == 0 ? nA : nB, self_mode = nB == 0 ? 1 : 0;
  return layout(nA, nBeff, g, dtype == SQFA_F32 ? 4 : 8, self_mode,
                choose_tile_width(nA, nBeff, g, self_mode, shard_count), shard_count).total;
}

int sqfa_airm_pairwise(const void* A, int nA, const void* B, int nB, int m, int dtype, double scale,
                       double eps, int sqrt_mode, const void* pair_weights, double uniform_weight,
                       int shard_index, int shard_count, void* loss_out, void* gradA_out,
                       void* gradB_out, void* dist_out, void* eig_out, int* nonfinite_out,
                       void* workspace, size_t workspace_bytes, void* stream_) {
  return pairwise_impl(A, nA, B, nB, m, dtype, scale, eps, sqrt_mode, pair_weights, uniform_weight, shard_index,
                       shard_count, loss_out, gradA_out, gradB_out, dist_out, eig_out, nonfinite_out, workspace,
                       workspace_bytes, stream_, nullptr, nullptr);
}

int sqfa_airm_pairwise_opt(const void* A, int nA, const void* B, int nB, int m, int dtype, double scale,
                           double eps, int sqrt_mode, const void* pair_weights, double uniform_weight,
                           int shard_index, int shard_count, void* loss_out, void* gradA_out,
                           void* gradB_out, void* dist_out, void* eig_out, int* nonfinite_out,
                           void* workspace, size_t workspace_bytes, void* stream_, const sqfa_airm_options* options) {
  return pairwise_impl(A, nA, B, nB, m, dtype, scale, eps, sqrt_mode, pair_weights, uniform_weight, shard_index,
                       shard_count, loss_out, gradA_out, gradB_out, dist_out, eig_out, nonfinite_out, workspace,
                       workspace_bytes, stream_, nullptr, options);
}

int sqfa_airm_eigenvalues_backward(const void* A, int nA, const void* B, int nB, int m, int dtype,
                                   const void* eig_weights, void* gradA_out, void* gradB_out,
                                   void* workspace, size_t workspace_bytes, void* stream_,
                                   const sqfa_airm_options* options) {
  if (eig_weights == nullptr || gradA_out == nullptr) {
    g_last_error[0] = 0;
    return fail(SQFA_ERR_BAD_ARGUMENT, "eig_weights / gradA_out", hipSuccess);
  }
  return pairwise_impl(A, nA, B, nB, m, dtype, 1.0, 0.0, 0, nullptr, 0.0, 0, 1, nullptr, gradA_out, gradB_out,
                       nullptr, nullptr, nullptr, workspace, workspace_bytes, stream_, eig_weights, options);
}

size_t sqfa_spd_function_workspace_bytes(int n, int m, int dtype) {
  Geometry g;
  if (n < 1 || m < 1 || (dtype != SQFA_F32 && dtype != SQFA_F64) || !find_geometry(m, dtype, -1, &g, -1)) return 0;
  return align_up((size_t)n * g.MR * g.MR * (dtype == SQFA_F32 ? 4 : 8));
}

int sqfa_spd_function(const void* S, int n, int m, int dtype, int kind, void* F_out, double* U_out, double* lam_out,
                      void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  g_last_error[0] = 0;
  if (S == nullptr || n < 1 || m < 1 || U_out == nullptr || lam_out == nullptr || workspace == nullptr)
    return fail(SQFA_ERR_BAD_ARGUMENT, "null/size argument", hipSuccess);
  if (dtype != SQFA_F32 && dtype != SQFA_F64) return fail(SQFA_ERR_BAD_ARGUMENT, "dtype", hipSuccess);
  if (kind != SQFA_SPD_LOG && kind != SQFA_SPD_SQRT && kind != SQFA_SPD_INV_SQRT) return fail(SQFA_ERR_BAD_ARGUMENT, "kind", hipSuccess);
  Geometry g;
  if (!find_geometry(m, dtype, -1, &g, -1) || g.eig == nullptr) return fail(SQFA_ERR_UNSUPPORTED_M, "matrix size not supported", hipSuccess);
  if (workspace_bytes < sqfa_spd_function_workspace_bytes(n, m, dtype)) return fail(SQFA_ERR_WORKSPACE, "workspace too small", hipSuccess);
  PairParams p;
  memset(&p, 0, sizeof(p));
  // Cholesky factor of every class, columns contiguous, identity padded to the size class (K0; double inside)
  auto chol = [&](auto zero) {
    using T = decltype(zero);
    const T* sp = static_cast<const T*>(S);
    T* ltp = static_cast<T*>(workspace);
    if (m <= 16) hipLaunchKernelGGL((cholesky_kernel<T, 16>), dim3(n), dim3(256), 0, stream, sp, m, g.MR, ltp, (T*)nullptr, (int*)nullptr, p, g.TI);
    else if (m <= 32) hipLaunchKernelGGL((cholesky_kernel<T, 32>), dim3(n), dim3(256), 0, stream, sp, m, g.MR, ltp, (T*)nullptr, (int*)nullptr, p, g.TI);
    else hipLaunchKernelGGL((cholesky_kernel<T, 64>), dim3(n), dim3(256), 0, stream, sp, m, g.MR, ltp, (T*)nullptr, (int*)nullptr, p, g.TI);
  };
  if (dtype == SQFA_F32) chol(0.0f); else chol(0.0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SQFA_ERR_LAUNCH, "cholesky_kernel", e);
  e = g.eig(workspace, n, m, U_out, lam_out, stream);
  if (e != hipSuccess) return fail(SQFA_ERR_LAUNCH, "class_eig_kernel", e);
  if (F_out != nullptr) {
    const size_t lds = ((size_t)m * m + m) * sizeof(double);
    if (dtype == SQFA_F32) hipLaunchKernelGGL(spd_function_kernel<float>, dim3(n), dim3(256), lds, stream, U_out, lam_out, m, kind, static_cast<float*>(F_out));
    else hipLaunchKernelGGL(spd_function_kernel<double>, dim3(n), dim3(256), lds, stream, U_out, lam_out, m, kind, static_cast<double*>(F_out));
    e = hipGetLastError();
    if (e != hipSuccess) return fail(SQFA_ERR_LAUNCH, "spd_function_kernel", e);
  }
  return SQFA_OK;
}

int sqfa_spd_function_backward(const double* U, const double* lam, const void* G, int n, int m, int dtype, int kind,
                               void* gradS_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  g_last_error[0] = 0;
  if (U == nullptr || lam == nullptr || G == nullptr || gradS_out == nullptr || n < 1 || m < 1)
    return fail(SQFA_ERR_BAD_ARGUMENT, "null/size argument", hipSuccess);
  if (dtype != SQFA_F32 && dtype != SQFA_F64) return fail(SQFA_ERR_BAD_ARGUMENT, "dtype", hipSuccess);
  if (kind != SQFA_SPD_LOG && kind != SQFA_SPD_SQRT && kind != SQFA_SPD_INV_SQRT) return fail(SQFA_ERR_BAD_ARGUMENT, "kind", hipSuccess);
  if (m > max_dim()) return fail(SQFA_ERR_UNSUPPORTED_M, "matrix size not supported", hipSuccess);
  const size_t lds = ((size_t)3 * m * m + m) * sizeof(double);   // 96.5 KB at m = 64
  if (dtype == SQFA_F32) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spd_function_backward_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(spd_function_backward_kernel<float>, dim3(n), dim3(256), lds, stream, U, lam, static_cast<const float*>(G), m, kind, static_cast<float*>(gradS_out));
  } else {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spd_function_backward_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(spd_function_backward_kernel<double>, dim3(n), dim3(256), lds, stream, U, lam, static_cast<const double*>(G), m, kind, static_cast<double*>(gradS_out));
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SQFA_ERR_LAUNCH, "spd_function_backward_kernel", e);
  return SQFA_OK;
}

}  // extern "C"
