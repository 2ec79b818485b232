// lbfgs_kernels.hip -- the L-BFGS search direction d = -H g in compact form, for optimizer state that
// lives on the device (parameters > 8192 elements: the c3, c4, c5 shapes).
//
// The reference optimises with torch.optim.LBFGS (src/sqfa/_optim.py:78-82), whose two-loop recursion is
// ~4 x history tiny vector operations per iteration; sqfa_amd/_lbfgs.py restates it as two triangular
// solves and four (history x n) matrix-vector products -- still ~35 torch launches, index uploads and a
// slot gather per iteration, 0.4 ms of host time per closure at c5.  Here the same arithmetic is six
// launches and no host-device traffic (round 2, late: the dot products split over ~768 workgroups with 16-byte
// loads, the triangle of the solves staged in LDS, independent accumulators in the combinations -- the first
// version's 6 launches took 230 us at h = 100, n = 49152, a quarter of the c5 fit's GPU time):
//
//   b0  = -S g                       (k dot products)            lb_dots
//   al  = triu(SY)^-1 b0             (one workgroup)             lb_solve (upper)
//   r0  = H (-g - Y^T al)            (n threads)                 lb_combine
//   yr  = Y r0                       (k dot products)            lb_dots
//   c   = tril(SY^T)^-1 (diag(SY) al - yr)                       lb_solve (lower)
//   d   = r0 + S^T c                 (n threads)                 lb_combine
//
// S, Y are (h, n) ring buffers of steps / gradient differences, SY[i][j] = s_i . y_j; `slots` lists the
// ring rows in chronological order (k <= h <= 128 valid pairs).  Fixed summation orders: reproducible.
#include <hip/hip_runtime.h>

#include "../../include/sqfa_hip.h"

namespace sqfa {

constexpr int LB_MAX_HISTORY = 128;
constexpr int LB_MAX_PARTS = 16;     // a dot product is split over at most this many workgroups
constexpr int LB_PART_MIN = 2048;    // elements per part at least
struct Slots { int v[LB_MAX_HISTORY]; };

// parts a length-n dot product is split into when `rows` of them are evaluated at once: ~768 workgroups in
// flight (a grid of `rows` <= 128 workgroups left most of the 256 CUs idle: 51 us per launch at n = 49152)
static int lb_parts(int rows, int n) {
  int p = (768 + rows - 1) / rows;
  const int cap = (n + LB_PART_MIN - 1) / LB_PART_MIN;
  if (p > cap) p = cap;
  if (p > LB_MAX_PARTS) p = LB_MAX_PARTS;
  return p < 1 ? 1 : p;
}

template <typename T> __device__ __forceinline__ T lb_block_sum(T v, T* s_red) {
  const int tid = threadIdx.x;
  s_red[tid] = v;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) s_red[tid] += s_red[tid + st];
    __syncthreads();
  }
  return s_red[0];
}

// part[i][p] = sum over the p-th chunk of  M[row_i] . v   (row_i = sl.v[i], or i itself when identity != 0;
// row `swap_row` is read from `swap_src` instead: the push uses that for the row being replaced).  The
// consumer adds the parts of a row in index order: fixed summation order, reproducible.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void lb_dots(const T* __restrict__ M, int n, Slots sl, int identity, const T* __restrict__ v,
                                               T* __restrict__ part, int parts, int chunk, int swap_row,
                                               const T* __restrict__ swap_src) {
  __shared__ T s_red[256];
  struct alignas(4 * sizeof(T)) Vec4 { T v[4]; };
  const int i = blockIdx.x, p = blockIdx.y;
  const int row = identity ? i : sl.v[i];
  const T* m = (row == swap_row && swap_src != nullptr) ? swap_src : M + (size_t)row * n;
  const int lo = p * chunk, hi = min(n, lo + chunk);
  T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
  if constexpr (VEC) {  // chunk and n are multiples of 4, rows 16-byte (f32) / 32-byte (f64) aligned
    for (int e = lo + 4 * threadIdx.x; e < hi; e += 1024) {
      const Vec4 a = *reinterpret_cast<const Vec4*>(m + e), b = *reinterpret_cast<const Vec4*>(v + e);
      a0 += a.v[0] * b.v[0];
      a1 += a.v[1] * b.v[1];
      a2 += a.v[2] * b.v[2];
      a3 += a.v[3] * b.v[3];
    }
  } else {
    for (int e = lo + threadIdx.x; e < hi; e += 256) a0 += m[e] * v[e];
  }
  const T tot = lb_block_sum((a0 + a1) + (a2 + a3), s_red);
  if (threadIdx.x == 0) part[(size_t)i * parts + p] = tot;
}

__device__ __forceinline__ float lb_read_lane(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double lb_read_lane(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

template <typename T> __device__ __forceinline__ T lb_sum_parts(const T* __restrict__ part, int i, int parts) {
  T t = T(0);
  for (int p = 0; p < parts; ++p) t += part[(size_t)i * parts + p];
  return t;
}

// x <- T^-1 x for the k x k triangular matrix T_ij = SY[sl_i][sl_j] (upper, j >= i) or SY[sl_j][sl_i]
// (lower, j <= i).  upper: x = rhs_scale * (sum of the dot-product parts); lower: x = diag(SY) al - (sum of the parts).
// The triangle is first gathered into LDS (packed, s_m[tri(i,j)], dynamic shared memory of k(k+1)/2 elements)
// when `in_lds`: the k dependent steps then cost two barriers each instead of an L2 round trip (35 -> ~6 us at k = 100).
template <typename T, bool lower, bool in_lds>
__global__ __launch_bounds__(256) void lb_solve(const T* __restrict__ SY, int h, Slots sl, int k, T* __restrict__ x,
                                                const T* __restrict__ al, const T* __restrict__ part, int parts,
                                                T rhs_scale) {
  extern __shared__ __align__(16) unsigned char s_raw[];
  __shared__ T s_x[LB_MAX_HISTORY], s_rd[LB_MAX_HISTORY];
  __shared__ int s_inv[LB_MAX_HISTORY];
  T* s_m = reinterpret_cast<T*>(s_raw);
  const int tid = threadIdx.x;
  const int me = tid < k ? sl.v[tid] : 0;
  if (tid < k) {
    const T dot = lb_sum_parts(part, tid, parts);
    s_x[tid] = lower ? SY[(size_t)me * h + me] * al[tid] - dot : rhs_scale * dot;
  }
  // ring row -> chronological index (or -1)
  if (tid < LB_MAX_HISTORY) s_inv[tid] = -1;
  __syncthreads();
  if (tid < k) s_inv[me] = tid;
  __syncthreads();
  if (in_lds) {
    // SY is read whole, coalesced, with simple independent loads (a gather of the k(k+1)/2 wanted entries with
    // per-entry index arithmetic serialised its loads: 20 us of the kernel's 32); entry (a, b), a <= b
    // chronological, lands packed by the larger index first at b(b+1)/2 + a
    const int pj = tid & (LB_MAX_HISTORY - 1), cj = pj < h ? s_inv[pj] : -1;
#pragma unroll 8
    for (int pi = tid / LB_MAX_HISTORY; pi < h; pi += 256 / LB_MAX_HISTORY) {
      const T v = pj < h ? SY[(size_t)pi * h + pj] : T(0);
      const int ci = s_inv[pi];
      if (ci >= 0 && cj >= ci) s_m[cj * (cj + 1) / 2 + ci] = v;
    }
  }
  __syncthreads();
  auto entry = [&](int small, int large) -> T {  // SY[s_small][s_large]
    return in_lds ? s_m[large * (large + 1) / 2 + small] : SY[(size_t)sl.v[small] * h + sl.v[large]];
  };
  // the k dependent steps run in ONE wave (lane l owns rows l and l + 64, values in registers, the pivot
  // travels by v_readlane): no workgroup barriers on the critical path; a step is readlane -> mul -> fma
  if (tid < k) s_rd[tid] = T(1) / entry(tid, tid);
  __syncthreads();
  if (tid >= 64) return;
  T x0 = tid < k ? s_x[tid] : T(0), x1 = tid + 64 < k ? s_x[tid + 64] : T(0);
  const int r0 = tid, r1 = tid + 64;
  const int c0 = min(r0, k - 1), c1 = min(r1, k - 1);  // clamped rows: any in-range entry, masked below
  auto sym = [&](int a, int b) -> T { return a < b ? entry(a, b) : entry(b, a); };
  // four steps at a time: their matrix entries and reciprocal pivots do not depend on x, so the 12 LDS reads
  // go out together and ONE wait covers four dependent readlane -> mul -> fma steps (one step at a time:
  // ~600 cycles per step, 22 of the kernel's 33 us at k = 100)
  const int first = lower ? 0 : k - 1, dir = lower ? 1 : -1;
  for (int st = 0; st < k; st += 4) {
    T e0[4], e1[4], rd[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = first + dir * min(st + u, k - 1);  // clamped: the surplus steps of the last group are skipped below
      e0[u] = sym(c0, i);
      e1[u] = sym(c1, i);
      rd[u] = s_rd[i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (st + u < k) {  // uniform
        const int i = first + dir * (st + u);
        const T piv = lb_read_lane(i < 64 ? x0 : x1, i & 63) * rd[u];
        const bool below0 = lower ? (r0 > i && r0 < k) : r0 < i, below1 = lower ? (r1 > i && r1 < k) : r1 < i;
        x0 = r0 == i ? piv : (below0 ? x0 - e0[u] * piv : x0);   // upper: U[r][i] = SY[s_r][s_i]; lower: L[r][i] = SY[s_i][s_r]
        x1 = r1 == i ? piv : (below1 ? x1 - e1[u] * piv : x1);
      }
    }
  }
  if (r0 < k) x[r0] = x0;
  if (r1 < k) x[r1] = x1;
}

// out[e] = post * (base_scale * base[e] + sign * sum_i coef[i] * M[sl_i][e]); the k row reads of a thread are
// independent: four accumulators keep that many loads in flight (one dependent chain: 28 us at k = 100, n = 49152)
template <typename T>
__global__ __launch_bounds__(256) void lb_combine(const T* __restrict__ M, int n, Slots sl, int k, const T* __restrict__ coef,
                                                  T sign, const T* __restrict__ base, T base_scale,
                                                  const T* __restrict__ post, T* __restrict__ out) {
  __shared__ T s_coef[LB_MAX_HISTORY];
  __shared__ int s_row[LB_MAX_HISTORY];
  if (threadIdx.x < k) {
    s_coef[threadIdx.x] = coef[threadIdx.x];
    s_row[threadIdx.x] = sl.v[threadIdx.x];
  }
  __syncthreads();
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
  int i = 0;
  for (; i + 4 <= k; i += 4) {
    a0 += s_coef[i] * M[(size_t)s_row[i] * n + e];
    a1 += s_coef[i + 1] * M[(size_t)s_row[i + 1] * n + e];
    a2 += s_coef[i + 2] * M[(size_t)s_row[i + 2] * n + e];
    a3 += s_coef[i + 3] * M[(size_t)s_row[i + 3] * n + e];
  }
  for (; i < k; ++i) a0 += s_coef[i] * M[(size_t)s_row[i] * n + e];
  const T p = post != nullptr ? post[0] : T(1);
  out[e] = p * (base_scale * base[e] + sign * ((a0 + a1) + (a2 + a3)));
}

// the push's last launch: rows `slot` of S and Y, and (last workgroup) the finished row and column `slot` of SY
template <typename T>
__global__ __launch_bounds__(256) void lb_store_rows(T* __restrict__ S, T* __restrict__ Y, T* __restrict__ SY, int h, int n,
                                                     int slot, const T* __restrict__ s, const T* __restrict__ y,
                                                     const T* __restrict__ part_row, const T* __restrict__ part_col, int parts) {
  if (blockIdx.x == gridDim.x - 1) {
    const int tid = threadIdx.x;
    if (tid < h) SY[(size_t)slot * h + tid] = lb_sum_parts(part_row, tid, parts);                                // y_j . s
    else if (tid >= LB_MAX_HISTORY && tid - LB_MAX_HISTORY < h)
      SY[(size_t)(tid - LB_MAX_HISTORY) * h + slot] = lb_sum_parts(part_col, tid - LB_MAX_HISTORY, parts);       // s_i . y
    return;
  }
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  S[(size_t)slot * n + e] = s[e];
  Y[(size_t)slot * n + e] = y[e];
}

// The element-wise part of an iteration and its decision scalars in two launches (torch: sub, mul, abs, max, abs,
// max, dot, dot, div, stack -- a dozen launches): y = g - g_prev, s = t d, and
//   out = [max|g|, max|s|, y.s, y.y, y.s / y.y]
// (the last one is torch's H_diag for the next direction, valid when the pair is accepted).  Block partials are
// combined in index order by the second launch: reproducible.
constexpr int LB_STAT_BLOCKS = 256;
// max that PROPAGATES NaN like torch's abs().max() (fmax drops it: an all-NaN gradient would read as max|g| = 0 and
// end the step as "converged" where torch.optim.LBFGS and the reference carry the NaN on)
template <typename T> __device__ __forceinline__ T lb_nan_max(T a, T b) { return (a != a || b != b) ? (a + b) : fmax(a, b); }
template <typename T>
__global__ __launch_bounds__(256) void lb_step_stats(const T* __restrict__ g, const T* __restrict__ g_prev, const T* __restrict__ d,
                                                     T t, int n, T* __restrict__ y, T* __restrict__ sv, T* __restrict__ part) {
  __shared__ T s_red[256];
  T gmax = T(0), smax = T(0), ys = T(0), yy = T(0);
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const T ge = g[e], ye = ge - g_prev[e], se = t * d[e];
    y[e] = ye;
    sv[e] = se;
    gmax = lb_nan_max(gmax, fabs(ge));
    smax = lb_nan_max(smax, fabs(se));
    ys += ye * se;
    yy += ye * ye;
  }
  const T ys_b = lb_block_sum(ys, s_red);
  __syncthreads();
  const T yy_b = lb_block_sum(yy, s_red);
  __syncthreads();
  // maxima: same tree with max
  auto block_max = [&](T v) {
    s_red[threadIdx.x] = v;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (threadIdx.x < st) s_red[threadIdx.x] = lb_nan_max(s_red[threadIdx.x], s_red[threadIdx.x + st]);
      __syncthreads();
    }
    const T r = s_red[0];
    __syncthreads();
    return r;
  };
  const T gm_b = block_max(gmax), sm_b = block_max(smax);
  if (threadIdx.x == 0) {
    part[4 * blockIdx.x + 0] = gm_b;
    part[4 * blockIdx.x + 1] = sm_b;
    part[4 * blockIdx.x + 2] = ys_b;
    part[4 * blockIdx.x + 3] = yy_b;
  }
}
template <typename T>
__global__ __launch_bounds__(64) void lb_step_stats_finish(const T* __restrict__ part, int blocks, T* __restrict__ out) {
  if (threadIdx.x != 0) return;
  T gmax = T(0), smax = T(0), ys = T(0), yy = T(0);
  for (int b = 0; b < blocks; ++b) {
    gmax = lb_nan_max(gmax, part[4 * b + 0]);
    smax = lb_nan_max(smax, part[4 * b + 1]);
    ys += part[4 * b + 2];
    yy += part[4 * b + 3];
  }
  out[0] = gmax;
  out[1] = smax;
  out[2] = ys;
  out[3] = yy;
  out[4] = ys / yy;
}

template <typename T>
static int step_stats_impl(const T* g, const T* g_prev, const T* d, double t, int n, T* y, T* s, T* out, T* work, hipStream_t stream) {
  int blocks = (n + 1023) / 1024;  // >= 4 elements per thread
  if (blocks > LB_STAT_BLOCKS) blocks = LB_STAT_BLOCKS;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(lb_step_stats<T>, dim3(blocks), dim3(256), 0, stream, g, g_prev, d, (T)t, n, y, s, work);
  hipLaunchKernelGGL(lb_step_stats_finish<T>, dim3(1), dim3(64), 0, stream, (const T*)work, blocks, out);
  return hipGetLastError() == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}

template <typename T> static bool lb_vec_ok(int n, int chunk, const void* a, const void* b, const void* c) {
  const size_t al = 4 * sizeof(T);
  return (n % 4) == 0 && (chunk % 4) == 0 && (reinterpret_cast<size_t>(a) % al) == 0 && (reinterpret_cast<size_t>(b) % al) == 0 &&
         (c == nullptr || (reinterpret_cast<size_t>(c) % al) == 0);
}

template <typename T>
static void launch_dots(const T* M, int n, const Slots& sl, int rows, int identity, const T* v, T* part, int parts,
                        int swap_row, const T* swap_src, hipStream_t stream) {
  int chunk = (n + parts - 1) / parts;
  chunk = (chunk + 3) / 4 * 4;
  const dim3 grid(rows, parts, 1);
  if (lb_vec_ok<T>(n, chunk, M, v, swap_src))
    hipLaunchKernelGGL((lb_dots<T, true>), grid, dim3(256), 0, stream, M, n, sl, identity, v, part, parts, chunk, swap_row, swap_src);
  else
    hipLaunchKernelGGL((lb_dots<T, false>), grid, dim3(256), 0, stream, M, n, sl, identity, v, part, parts, chunk, swap_row, swap_src);
}

template <typename T>
static void launch_solve(const T* SY, int h, const Slots& sl, int k, T* x, int lower, const T* al, const T* part, int parts,
                         T rhs_scale, hipStream_t stream) {
  const size_t tri_bytes = (size_t)k * (k + 1) / 2 * sizeof(T);
  const int in_lds = tri_bytes <= 60 * 1024;  // float64 histories above 123 pairs read SY from L2
  if (in_lds && lower)
    hipLaunchKernelGGL((lb_solve<T, true, true>), dim3(1), dim3(256), tri_bytes, stream, SY, h, sl, k, x, al, part, parts, rhs_scale);
  else if (in_lds)
    hipLaunchKernelGGL((lb_solve<T, false, true>), dim3(1), dim3(256), tri_bytes, stream, SY, h, sl, k, x, al, part, parts, rhs_scale);
  else if (lower)
    hipLaunchKernelGGL((lb_solve<T, true, false>), dim3(1), dim3(256), 0, stream, SY, h, sl, k, x, al, part, parts, rhs_scale);
  else
    hipLaunchKernelGGL((lb_solve<T, false, false>), dim3(1), dim3(256), 0, stream, SY, h, sl, k, x, al, part, parts, rhs_scale);
}

// work: [al (h) | c (h) | part (h * LB_MAX_PARTS) | part2 (h * LB_MAX_PARTS) | r0 (n)]
template <typename T>
static int direction_impl(const T* S, const T* Y, const T* SY, int h, int n, const int* slots, int k, const T* g,
                          const T* H, T* d, T* work, hipStream_t stream) {
  Slots sl;
  for (int i = 0; i < LB_MAX_HISTORY; ++i) sl.v[i] = i < k ? slots[i] : 0;
  T* al = work;
  T* c = work + h;
  T* part = work + 2 * h;
  T* r0 = work + 2 * h + 2 * (size_t)h * LB_MAX_PARTS;
  const int nb = (n + 255) / 256;
  const int parts = lb_parts(k, n);
  launch_dots(S, n, sl, k, 0, g, part, parts, -1, (const T*)nullptr, stream);
  launch_solve(SY, h, sl, k, al, 0, (const T*)nullptr, part, parts, T(-1), stream);          // al = triu^-1 (-S g)
  hipLaunchKernelGGL(lb_combine<T>, dim3(nb), dim3(256), 0, stream, Y, n, sl, k, al, T(-1), g, T(-1), H, r0);
  launch_dots(Y, n, sl, k, 0, (const T*)r0, part, parts, -1, (const T*)nullptr, stream);
  launch_solve(SY, h, sl, k, c, 1, (const T*)al, part, parts, T(1), stream);                 // c = tril^-1 (diag al - Y r0)
  hipLaunchKernelGGL(lb_combine<T>, dim3(nb), dim3(256), 0, stream, S, n, sl, k, c, T(1), r0, T(1), (const T*)nullptr, d);
  return hipGetLastError() == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}

template <typename T>
static int push_impl(T* S, T* Y, T* SY, int h, int n, int slot, const T* s, const T* y, T* work, hipStream_t stream) {
  Slots sl{};
  // SY[slot][j] = y_j . s  and  SY[i][slot] = s_i . y  for every ring row (unused rows hold zeros or
  // stale pairs that the direction never selects); the row being replaced is read from (s, y) directly
  T* part_row = work + 2 * h;
  T* part_col = part_row + (size_t)h * LB_MAX_PARTS;
  const int parts = lb_parts(h, n);
  launch_dots((const T*)Y, n, sl, h, 1, s, part_row, parts, slot, y, stream);
  launch_dots((const T*)S, n, sl, h, 1, y, part_col, parts, slot, s, stream);
  hipLaunchKernelGGL(lb_store_rows<T>, dim3((n + 255) / 256 + 1), dim3(256), 0, stream, S, Y, SY, h, n, slot, s, y,
                     (const T*)part_row, (const T*)part_col, parts);
  return hipGetLastError() == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}

}  // namespace sqfa

using namespace sqfa;

extern "C" int sqfa_lbfgs_max_history(void) { return LB_MAX_HISTORY; }

extern "C" size_t sqfa_lbfgs_work_elems(int h, int n) {
  if (h < 1 || h > LB_MAX_HISTORY || n < 1) return 0;
  const size_t direction = 2 * (size_t)h + 2 * (size_t)h * LB_MAX_PARTS + (size_t)n, stats = 4 * (size_t)LB_STAT_BLOCKS;
  return direction > stats ? direction : stats;
}

extern "C" int sqfa_lbfgs_push(void* S, void* Y, void* SY, int h, int n, int slot, const void* s, const void* y, void* work,
                               int dtype, void* stream_) {
  if (S == nullptr || Y == nullptr || SY == nullptr || s == nullptr || y == nullptr || work == nullptr || h < 1 ||
      h > LB_MAX_HISTORY || n < 1 || slot < 0 || slot >= h)
    return SQFA_ERR_BAD_ARGUMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (dtype == SQFA_F32)
    return push_impl(static_cast<float*>(S), static_cast<float*>(Y), static_cast<float*>(SY), h, n, slot,
                     static_cast<const float*>(s), static_cast<const float*>(y), static_cast<float*>(work), stream);
  if (dtype == SQFA_F64)
    return push_impl(static_cast<double*>(S), static_cast<double*>(Y), static_cast<double*>(SY), h, n, slot,
                     static_cast<const double*>(s), static_cast<const double*>(y), static_cast<double*>(work), stream);
  return SQFA_ERR_BAD_ARGUMENT;
}

extern "C" int sqfa_lbfgs_step_stats(const void* g, const void* g_prev, const void* d, double t, int n, void* y_out,
                                     void* s_out, void* scalars_out, void* work, int dtype, void* stream_) {
  if (g == nullptr || g_prev == nullptr || d == nullptr || y_out == nullptr || s_out == nullptr || scalars_out == nullptr ||
      work == nullptr || n < 1)
    return SQFA_ERR_BAD_ARGUMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (dtype == SQFA_F32)
    return step_stats_impl(static_cast<const float*>(g), static_cast<const float*>(g_prev), static_cast<const float*>(d), t, n,
                           static_cast<float*>(y_out), static_cast<float*>(s_out), static_cast<float*>(scalars_out),
                           static_cast<float*>(work), stream);
  if (dtype == SQFA_F64)
    return step_stats_impl(static_cast<const double*>(g), static_cast<const double*>(g_prev), static_cast<const double*>(d), t, n,
                           static_cast<double*>(y_out), static_cast<double*>(s_out), static_cast<double*>(scalars_out),
                           static_cast<double*>(work), stream);
  return SQFA_ERR_BAD_ARGUMENT;
}

extern "C" int sqfa_lbfgs_direction(const void* S, const void* Y, const void* SY, int h, int n, const int* slots, int k,
                                    const void* g, const void* H_diag, void* d_out, void* work, int dtype, void* stream_) {
  if (S == nullptr || Y == nullptr || SY == nullptr || slots == nullptr || g == nullptr || d_out == nullptr ||
      work == nullptr || h < 1 || h > LB_MAX_HISTORY || n < 1 || k < 1 || k > h)
    return SQFA_ERR_BAD_ARGUMENT;
  for (int i = 0; i < k; ++i)
    if (slots[i] < 0 || slots[i] >= h) return SQFA_ERR_BAD_ARGUMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (dtype == SQFA_F32)
    return direction_impl(static_cast<const float*>(S), static_cast<const float*>(Y), static_cast<const float*>(SY), h, n,
                          slots, k, static_cast<const float*>(g), static_cast<const float*>(H_diag),
                          static_cast<float*>(d_out), static_cast<float*>(work), stream);
  if (dtype == SQFA_F64)
    return direction_impl(static_cast<const double*>(S), static_cast<const double*>(Y), static_cast<const double*>(SY), h,
                          n, slots, k, static_cast<const double*>(g), static_cast<const double*>(H_diag),
                          static_cast<double*>(d_out), static_cast<double*>(work), stream);
  return SQFA_ERR_BAD_ARGUMENT;
}
