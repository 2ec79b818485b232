// lbfgs_kernels.hip -- the L-BFGS search direction d = -H g in compact form, for optimizer state that
// lives on the device (parameters > 8192 elements: the c3, c4, c5 shapes).
//
// The reference optimises with torch.optim.LBFGS (src/sqfa/_optim.py:78-82), whose two-loop recursion is
// ~4 x history tiny vector operations per iteration; sqfa_amd/_lbfgs.py restates it as two triangular
// solves and four (history x n) matrix-vector products -- still ~35 torch launches, index uploads and a
// slot gather per iteration, 0.4 ms of host time per closure at c5.  Here the same arithmetic is six
// launches and no host-device traffic:
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
struct Slots { int v[LB_MAX_HISTORY]; };

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

// out[i] = scale * M[row_i] . v   (row_i = sl.v[i], or i itself when identity != 0; rows `swap_row`
// are read from `swap_src` instead: the push kernel uses that for the row being replaced)
template <typename T>
__global__ __launch_bounds__(256) void lb_dots(const T* __restrict__ M, int n, Slots sl, int identity, const T* __restrict__ v,
                                               T scale, T* __restrict__ out, int out_stride, int swap_row,
                                               const T* __restrict__ swap_src) {
  __shared__ T s_red[256];
  const int i = blockIdx.x;
  const int row = identity ? i : sl.v[i];
  const T* m = (row == swap_row && swap_src != nullptr) ? swap_src : M + (size_t)row * n;
  T acc = T(0);
  for (int e = threadIdx.x; e < n; e += 256) acc += m[e] * v[e];
  const T tot = lb_block_sum(acc, s_red);
  if (threadIdx.x == 0) out[(size_t)i * out_stride] = scale * tot;
}

// x <- T^-1 x for the k x k triangular matrix T_ij = SY[sl_i][sl_j] (upper, j >= i) or SY[sl_j][sl_i]
// (lower, j <= i).  lower: the right-hand side is first formed as diag(SY) al - yr.
template <typename T>
__global__ __launch_bounds__(LB_MAX_HISTORY) void lb_solve(const T* __restrict__ SY, int h, Slots sl, int k, T* __restrict__ x,
                                                           int lower, const T* __restrict__ al, const T* __restrict__ yr) {
  __shared__ T s_x[LB_MAX_HISTORY];
  const int tid = threadIdx.x;
  const int me = tid < k ? sl.v[tid] : 0;
  if (tid < k) s_x[tid] = lower ? SY[(size_t)me * h + me] * al[tid] - yr[tid] : x[tid];
  __syncthreads();
  if (!lower) {
    for (int i = k - 1; i >= 0; --i) {
      const int si = sl.v[i];
      if (tid == i) s_x[i] = s_x[i] / SY[(size_t)si * h + si];
      __syncthreads();
      if (tid < i) s_x[tid] -= SY[(size_t)me * h + si] * s_x[i];   // U[tid][i] = SY[s_tid][s_i]
      __syncthreads();
    }
  } else {
    for (int i = 0; i < k; ++i) {
      const int si = sl.v[i];
      if (tid == i) s_x[i] = s_x[i] / SY[(size_t)si * h + si];
      __syncthreads();
      if (tid > i && tid < k) s_x[tid] -= SY[(size_t)si * h + me] * s_x[i];  // L[tid][i] = SY[s_i][s_tid]
      __syncthreads();
    }
  }
  if (tid < k) x[tid] = s_x[tid];
}

// out[e] = post * (base_scale * base[e] + sign * sum_i coef[i] * M[sl_i][e])
template <typename T>
__global__ __launch_bounds__(256) void lb_combine(const T* __restrict__ M, int n, Slots sl, int k, const T* __restrict__ coef,
                                                  T sign, const T* __restrict__ base, T base_scale,
                                                  const T* __restrict__ post, T* __restrict__ out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  T acc = T(0);
  for (int i = 0; i < k; ++i) acc += coef[i] * M[(size_t)sl.v[i] * n + e];
  const T p = post != nullptr ? post[0] : T(1);
  out[e] = p * (base_scale * base[e] + sign * acc);
}

template <typename T>
__global__ __launch_bounds__(256) void lb_store_rows(T* __restrict__ S, T* __restrict__ Y, int n, int slot,
                                                     const T* __restrict__ s, const T* __restrict__ y) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  S[(size_t)slot * n + e] = s[e];
  Y[(size_t)slot * n + e] = y[e];
}

template <typename T>
static int direction_impl(const T* S, const T* Y, const T* SY, int h, int n, const int* slots, int k, const T* g,
                          const T* H, T* d, T* work, hipStream_t stream) {
  Slots sl;
  for (int i = 0; i < LB_MAX_HISTORY; ++i) sl.v[i] = i < k ? slots[i] : 0;
  T* al = work;           // k
  T* yr = work + h;       // k
  T* c = work + 2 * h;    // k
  T* r0 = work + 3 * h;   // n
  const int nb = (n + 255) / 256;
  hipLaunchKernelGGL(lb_dots<T>, dim3(k), dim3(256), 0, stream, S, n, sl, 0, g, T(-1), al, 1, -1, (const T*)nullptr);
  hipLaunchKernelGGL(lb_solve<T>, dim3(1), dim3(LB_MAX_HISTORY), 0, stream, SY, h, sl, k, al, 0, (const T*)nullptr, (const T*)nullptr);
  hipLaunchKernelGGL(lb_combine<T>, dim3(nb), dim3(256), 0, stream, Y, n, sl, k, al, T(-1), g, T(-1), H, r0);
  hipLaunchKernelGGL(lb_dots<T>, dim3(k), dim3(256), 0, stream, Y, n, sl, 0, r0, T(1), yr, 1, -1, (const T*)nullptr);
  hipLaunchKernelGGL(lb_solve<T>, dim3(1), dim3(LB_MAX_HISTORY), 0, stream, SY, h, sl, k, c, 1, al, yr);
  hipLaunchKernelGGL(lb_combine<T>, dim3(nb), dim3(256), 0, stream, S, n, sl, k, c, T(1), r0, T(1), (const T*)nullptr, d);
  return hipGetLastError() == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}

template <typename T>
static int push_impl(T* S, T* Y, T* SY, int h, int n, int slot, const T* s, const T* y, hipStream_t stream) {
  Slots sl{};
  // SY[slot][j] = y_j . s  and  SY[i][slot] = s_i . y  for every ring row (unused rows hold zeros or
  // stale pairs that the direction never selects); the row being replaced is read from (s, y) directly
  hipLaunchKernelGGL(lb_dots<T>, dim3(h), dim3(256), 0, stream, (const T*)Y, n, sl, 1, s, T(1), SY + (size_t)slot * h, 1, slot, y);
  hipLaunchKernelGGL(lb_dots<T>, dim3(h), dim3(256), 0, stream, (const T*)S, n, sl, 1, y, T(1), SY + slot, h, slot, s);
  hipLaunchKernelGGL(lb_store_rows<T>, dim3((n + 255) / 256), dim3(256), 0, stream, S, Y, n, slot, s, y);
  return hipGetLastError() == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}

}  // namespace sqfa

using namespace sqfa;

extern "C" int sqfa_lbfgs_max_history(void) { return LB_MAX_HISTORY; }

extern "C" int sqfa_lbfgs_push(void* S, void* Y, void* SY, int h, int n, int slot, const void* s, const void* y, int dtype,
                               void* stream_) {
  if (S == nullptr || Y == nullptr || SY == nullptr || s == nullptr || y == nullptr || h < 1 || h > LB_MAX_HISTORY ||
      n < 1 || slot < 0 || slot >= h)
    return SQFA_ERR_BAD_ARGUMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (dtype == SQFA_F32)
    return push_impl(static_cast<float*>(S), static_cast<float*>(Y), static_cast<float*>(SY), h, n, slot,
                     static_cast<const float*>(s), static_cast<const float*>(y), stream);
  if (dtype == SQFA_F64)
    return push_impl(static_cast<double*>(S), static_cast<double*>(Y), static_cast<double*>(SY), h, n, slot,
                     static_cast<const double*>(s), static_cast<const double*>(y), stream);
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
