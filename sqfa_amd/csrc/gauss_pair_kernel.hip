// gauss_pair_kernel.hip -- per-pair Gaussian terms behind the reference's "other" distance_fun
// operators (src/sqfa/distances.py:240-432): for every pair (i, j) of Gaussians
//     Sbar = (Sigma_i + Sigma_j) / 2 = R R^T,   delta = mu_i - mu_j
//     Q_ij  = delta^T Sbar^-1 delta     (mahalanobis_sq; /8 in bhattacharyya; acosh(1+Q/4) in fisher_rao_same_cov)
//     LD_ij = log det Sbar              (bhattacharyya / hellinger)
// and, given upstream gradients gQ, gLD (nA,nB), the gradient of sum_ij (gQ_ij Q_ij + gLD_ij LD_ij)
// with respect to the A side:
//     d/dmu_i    = sum_j gQ_ij 2 s_ij,                          s = Sbar^-1 delta
//     d/dSigma_i = sum_j 1/2 (-gQ_ij s s^T + gLD_ij Sbar^-1)
// The reference forms the (nA,nB,K,K) tensor of mean covariances and calls batched inv / logdet on
// it (1 GB at C=1000, K=16); here nothing larger than (nA,nB) is ever stored.
//
// Mapping: one workgroup per A class i; a lane group of G = pow2 >= m lanes (inside one wave) owns
// one pair at a time and walks j = group, group + NG, ...; lane r of the group owns row r.  The
// matrices live in LDS (run-time m, no register indexing): Cholesky (right-looking), forward
// substitution for Q, back substitution for s, R^-1 column by column for Sbar^-1.  Each group
// keeps its own gradient accumulator in LDS; groups are combined in a fixed order at the end
// (bitwise reproducible, no atomics).  The B-side gradient is the same launch with the roles
// swapped (the caller transposes gQ / gLD).
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../include/sqfa_hip.h"

namespace sqfa {

template <typename T> __device__ __forceinline__ T g_rsqrt(T x);
template <> __device__ __forceinline__ float g_rsqrt<float>(float x) { return 1.0f / sqrtf(x); }
template <> __device__ __forceinline__ double g_rsqrt<double>(double x) { return 1.0 / sqrt(x); }
__device__ __forceinline__ float g_log(float x) { return logf(x); }
__device__ __forceinline__ double g_log(double x) { return log(x); }

// LDS of one wave is processed in program order; this only stops the compiler from moving LDS
// accesses across the hand-off between lanes of a group (all inside one wave)
__device__ __forceinline__ void group_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int G, typename T> __device__ __forceinline__ T group_total(T v) {
#pragma unroll
  for (int d = 1; d < G; d <<= 1) v += __shfl_xor(v, d, 64);
  return v;
}

struct GaussParams {
  const void *muA, *covA, *muB, *covB;
  const void *gQ, *gLD;      // (nA,nB) or nullptr
  void *Q, *LD;              // (nA,nB) or nullptr
  void *gmuA, *gcovA;        // (nA,m), (nA,m,m) or nullptr
  int nA, nB, m, ng;         // ng: lane groups per workgroup
};

template <typename T, int G>
__global__ void gauss_pair_kernel(const GaussParams p) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  constexpr int LD_ = G + 1;                 // row pitch
  constexpr int MAT = G * LD_;
  constexpr int PER_GROUP = 3 * MAT + 2 * G;  // a, w, acc, s, accmu
  const int tid = threadIdx.x;
  const int grp = tid / G, r = tid % G;
  const int m = p.m;
  const bool live = grp < p.ng;   // a block is padded to whole waves: surplus groups own no LDS and no pair
  const bool active = live && r < m;
  const int i = blockIdx.x;
  T* a = smem + (size_t)(live ? grp : 0) * PER_GROUP;
  T* w = a + MAT;
  T* acc = w + MAT;
  T* sv = acc + MAT;
  T* accmu = sv + G;
  const T* muA = static_cast<const T*>(p.muA);
  const T* covA = static_cast<const T*>(p.covA);
  const T* muB = static_cast<const T*>(p.muB);
  const T* covB = static_cast<const T*>(p.covB);
  const T* gQ = static_cast<const T*>(p.gQ);
  const T* gLD = static_cast<const T*>(p.gLD);
  const bool want_grad = p.gcovA != nullptr;
  const int lane_base = (tid & 63) & ~(G - 1);  // first lane of my group inside the wave

  if (want_grad && live) {
    for (int c = 0; c < G; ++c) acc[r * LD_ + c] = T(0);
  }
  T mu_acc = T(0);
  const T mu_i = active ? muA[(size_t)i * m + r] : T(0);
  const T* ci = covA + (size_t)i * m * m;

  for (int j = live ? grp : p.nB; j < p.nB; j += p.ng) {
    const T* cj = covB + (size_t)j * m * m;
    if (active) {
      for (int c = 0; c <= r; ++c) a[r * LD_ + c] = T(0.5) * (ci[r * m + c] + cj[r * m + c]);
    }
    T d = active ? mu_i - muB[(size_t)j * m + r] : T(0);
    group_sync();
    // ---- Cholesky Sbar = R R^T, in place (lower) --------------------------------------
    for (int k = 0; k < m; ++k) {
      const T piv = a[k * LD_ + k];
      const T rs = g_rsqrt<T>(piv);   // non-positive pivot -> NaN everywhere downstream, never a fault
      T l = T(0);
      if (active && r >= k) {
        l = a[r * LD_ + k] * rs;
        a[r * LD_ + k] = l;
      }
      group_sync();
      if (active && r > k) {
        for (int c = k + 1; c <= r; ++c) a[r * LD_ + c] -= l * a[c * LD_ + k];
      }
      group_sync();
    }
    const T diag = active ? a[r * LD_ + r] : T(1);
    const T rdiag = T(1) / diag;
    const T ld = T(2) * group_total<G>(active ? g_log(diag) : T(0));
    // ---- z = R^-1 delta (column oriented) ----------------------------------------------
    T z = T(0);
    for (int k = 0; k < m; ++k) {
      const T zk = __shfl(d * rdiag, lane_base + k, 64);
      if (r == k) z = zk;
      if (active && r > k) d -= a[r * LD_ + k] * zk;
    }
    const T q = group_total<G>(z * z);
    if (r == 0) {
      if (p.Q != nullptr) static_cast<T*>(p.Q)[(size_t)i * p.nB + j] = q;
      if (p.LD != nullptr) static_cast<T*>(p.LD)[(size_t)i * p.nB + j] = ld;
    }
    if (!want_grad) {
      group_sync();
      continue;
    }
    const T gq = gQ != nullptr ? gQ[(size_t)i * p.nB + j] : T(0);
    const T gl = gLD != nullptr ? gLD[(size_t)i * p.nB + j] : T(0);
    // ---- s = R^-T z (column oriented, from the last row up) -----------------------------
    T t = z, s = T(0);
    for (int k = m - 1; k >= 0; --k) {
      const T sk = __shfl(t * rdiag, lane_base + k, 64);
      if (r == k) s = sk;
      if (active && r < k) t -= a[k * LD_ + r] * sk;
    }
    sv[r] = s;
    mu_acc += T(2) * gq * s;
    // ---- W = R^-1: lane c owns column c ---------------------------------------------------
    if (gLD != nullptr) {
      if (active) {
        w[r * LD_ + r] = rdiag;
        for (int rr = r + 1; rr < m; ++rr) {
          T sum = T(0);
          for (int k = r; k < rr; ++k) sum += a[rr * LD_ + k] * w[k * LD_ + r];
          w[rr * LD_ + r] = -sum / a[rr * LD_ + rr];
        }
      }
    }
    group_sync();
    // ---- accumulate 1/2 (-gq s s^T + gl Sbar^-1), lower triangle, row r ---------------------
    if (active) {
      for (int c = 0; c <= r; ++c) {
        T v = -gq * s * sv[c];
        if (gLD != nullptr) {
          T inv = T(0);
          for (int k = r; k < m; ++k) inv += w[k * LD_ + r] * w[k * LD_ + c];
          v += gl * inv;
        }
        acc[r * LD_ + c] += T(0.5) * v;
      }
    }
    group_sync();
  }

  if (!want_grad) return;
  if (live) accmu[r] = mu_acc;
  __syncthreads();
  // combine the groups in a fixed order; write the full symmetric gradient
  T* gcov = static_cast<T*>(p.gcovA) + (size_t)i * m * m;
  for (int e = tid; e < m * m; e += blockDim.x) {
    const int rr = e / m, cc = e % m;
    const int hi = rr > cc ? rr : cc, lo = rr > cc ? cc : rr;
    T sum = T(0);
    for (int g2 = 0; g2 < p.ng; ++g2) sum += smem[(size_t)g2 * PER_GROUP + 2 * MAT + hi * LD_ + lo];
    gcov[e] = sum;
  }
  if (p.gmuA != nullptr) {
    T* gmu = static_cast<T*>(p.gmuA) + (size_t)i * m;
    for (int e = tid; e < m; e += blockDim.x) {
      T sum = T(0);
      for (int g2 = 0; g2 < p.ng; ++g2) sum += smem[(size_t)g2 * PER_GROUP + 3 * MAT + G + e];
      gmu[e] = sum;
    }
  }
}

template <typename T, int G>
static hipError_t launch_gauss(const GaussParams& p0, hipStream_t stream) {
  GaussParams p = p0;
  const size_t per_group = (size_t)(3 * G * (G + 1) + 2 * G) * sizeof(T);
  int ng = 256 / G;
  const size_t budget = 60 * 1024;
  while (ng > 1 && ng * per_group > budget) ng /= 2;
  if (ng > p.nB) {  // fewer pairs per class than groups: do not leave whole waves idle
    int need = 1;
    while (need < p.nB) need <<= 1;
    ng = need < ng ? need : ng;
    if (ng < 1) ng = 1;
  }
  p.ng = ng;
  const size_t lds = ng * per_group;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gauss_pair_kernel<T, G>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  int threads = ng * G;
  if (threads < 64) threads = 64;  // whole waves (extra lanes form idle groups: grp >= ng never owns a pair)
  hipLaunchKernelGGL((gauss_pair_kernel<T, G>), dim3(p.nA), dim3(threads), lds, stream, p);
  return hipGetLastError();
}

}  // namespace sqfa

using namespace sqfa;

extern "C" int sqfa_gauss_pair_terms(const void* muA, const void* covA, int nA, const void* muB, const void* covB,
                                     int nB, int m, int dtype, const void* gQ, const void* gLD, void* Q_out,
                                     void* LD_out, void* gmuA_out, void* gcovA_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (muA == nullptr || covA == nullptr || muB == nullptr || covB == nullptr || nA < 1 || nB < 1 || m < 1)
    return SQFA_ERR_BAD_ARGUMENT;
  if (dtype != SQFA_F32 && dtype != SQFA_F64) return SQFA_ERR_BAD_ARGUMENT;
  if (m > 64) return SQFA_ERR_UNSUPPORTED_M;
  if ((gmuA_out != nullptr) != (gcovA_out != nullptr)) return SQFA_ERR_BAD_ARGUMENT;
  if (gcovA_out != nullptr && gQ == nullptr && gLD == nullptr) return SQFA_ERR_BAD_ARGUMENT;
  GaussParams p{muA, covA, muB, covB, gQ, gLD, Q_out, LD_out, gmuA_out, gcovA_out, nA, nB, m, 1};
  hipError_t e;
#define SQFA_GAUSS_DISPATCH(T)                                         \
  if (m <= 4) e = launch_gauss<T, 4>(p, stream);                       \
  else if (m <= 8) e = launch_gauss<T, 8>(p, stream);                  \
  else if (m <= 16) e = launch_gauss<T, 16>(p, stream);                \
  else if (m <= 32) e = launch_gauss<T, 32>(p, stream);                \
  else e = launch_gauss<T, 64>(p, stream);
  if (dtype == SQFA_F32) {
    SQFA_GAUSS_DISPATCH(float)
  } else {
    SQFA_GAUSS_DISPATCH(double)
  }
  return e == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}
