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
#include "pair_kernel.hpp"   // transposing tree reductions (tree_reduce_blocks), wave_sum

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


// ---------------------------------------------------------------------------------------------------------------
// Register-resident variant (round 3) for small matrices: ONE PAIR PER LANE.  The LDS kernel above walks a 16 x 16
// factorisation through ~50 dependent LDS round trips with 16 lanes per pair: 4.9 ms for the 10^6 ordered pairs of
// C=1000, K=16 (forward + backward, float32), two orders of magnitude above its arithmetic.  Here a lane holds the lower
// triangle of Sbar (M(M+1)/2 registers, M = 4 / 8 / 12 / 16, identity padded), factorises it in place with every loop
// unrolled (no cross-lane traffic at all), inverts the factor in place (W = R^-1, then Sbar^-1 = W^T W), and the 64
// pairs of a wave -- same class i, 64 consecutive classes j -- reduce their gradient contributions with the
// transposing tree reduction of the pair kernel (lane l finishes entries 64 I + l) into a per-wave LDS accumulator.
// One workgroup (4 waves) per class i walks over all j; the four accumulators are added in wave order at the end:
// bitwise reproducible, no atomics.  float32 up to m = 16 (136 + ~40 registers), float64 up to m = 8.
// entry IDX of a lane's gradient contribution, formed on the fly from Sbar^-1 (lower triangle, in `a`) and s (in `d`):
// IDX < TRI: d/dSigma entry; TRI <= IDX < TRI + M: d/dmu entry
template <typename T, int M> struct GradEntries {
  static constexpr int TRI = M * (M + 1) / 2;
  const T (&a)[TRI];
  const T (&d)[M];
  T gq, gl;
  bool with_inverse;
  template <int IDX> __device__ __forceinline__ T get() const {
    if constexpr (IDX < TRI) {
      constexpr int r = tri_row(IDX), c = IDX - r * (r + 1) / 2;
      const T outer = gq * d[r] * d[c];
      return T(0.5) * (with_inverse ? gl * a[IDX] - outer : -outer);
    } else if constexpr (IDX < TRI + M) {
      return T(2) * gq * d[IDX - TRI];
    } else {
      return T(0);
    }
  }
};

// (two workgroups per CU = 256 VGPRs: without the bound the scheduler batches all 2 x M(M+1)/2 loads of a round ahead of
// the arithmetic and takes 313 registers at M=16)
// EXACT: m == M (no identity padding: no selects, rows of M elements, 16-byte row loads when M % 4 == 0 (float32) / M % 2
// == 0 (float64) and the matrices are 16-byte aligned, which the launcher checks)
template <typename T, int M, bool EXACT>
__global__ __launch_bounds__(256, (M * (M + 1) / 2) * (int)(sizeof(T) / 4) <= 40 ? 4 : ((M * (M + 1) / 2) * (int)(sizeof(T) / 4) <= 140 ? 2 : 1))
void gauss_pair_reg_kernel(const GaussParams p) {
  constexpr int TRI = M * (M + 1) / 2, NACC = TRI + M;
  __shared__ T s_acc[4][NACC | 1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = blockIdx.x, m = p.m;
  const T* muA = static_cast<const T*>(p.muA);
  const T* covA = static_cast<const T*>(p.covA);
  const T* muB = static_cast<const T*>(p.muB);
  const T* covB = static_cast<const T*>(p.covB);
  const T* gQ = static_cast<const T*>(p.gQ);
  const T* gLD = static_cast<const T*>(p.gLD);
  const bool want_grad = p.gcovA != nullptr;
  if (want_grad) {
    for (int k = lane; k < NACC; k += 64) s_acc[wave][k] = T(0);   // wave-private: no block barrier needed
  }
  const T* ci = covA + (size_t)i * m * m;
  auto at = [](int r, int c) constexpr { return r * (r + 1) / 2 + c; };  // c <= r

  const int rounds = (p.nB + 255) / 256;
  for (int it = 0; it < rounds; ++it) {
    const int j = it * 256 + tid;
    const bool valid = j < p.nB;
    const T* cj = covB + (size_t)(valid ? j : 0) * m * m;
    // Sigma_i is the same in every round: hidden from loop-invariant code motion, or its 136 values would be kept in
    // registers across the whole loop (321 VGPRs and spills at M=16 instead of ~200)
    const T* cir = ci;
    asm volatile("" : "+s"(cir));
    T a[TRI], d[M];
    // ---- Sbar = (Sigma_i + Sigma_j)/2 and its Cholesky factor R (lower, in place), ROW BY ROW: row r needs the finished
    //      rows above it and its own entries only, so the loads of a row sit next to their use (live set: the triangle
    //      built so far) ----
    T rdiag[M];
    T ld = T(0);
    constexpr int VW = 16 / (int)sizeof(T);
    struct alignas(16) RowVec { T v[VW]; };
#pragma unroll
    for (int r = 0; r < M; ++r) {
      T row[M];
      if constexpr (EXACT) {
        d[r] = muA[(size_t)i * M + r] - muB[(size_t)(valid ? j : 0) * M + r];
        if constexpr (M % VW == 0) {
#pragma unroll
          for (int c0 = 0; c0 <= r; c0 += VW) {
            const RowVec vi = *reinterpret_cast<const RowVec*>(cir + r * M + c0);
            const RowVec vj = *reinterpret_cast<const RowVec*>(cj + r * M + c0);
#pragma unroll
            for (int e = 0; e < VW; ++e) row[(c0 + e < M) ? c0 + e : 0] = T(0.5) * (vi.v[e] + vj.v[e]);
          }
        } else {
#pragma unroll
          for (int c = 0; c <= r; ++c) row[c] = T(0.5) * (cir[r * M + c] + cj[r * M + c]);
        }
      } else {
        const bool rr = r < m;
        const int rc = rr ? r : m - 1;   // clamped addresses, selected values: no branches
        const T dv = muA[(size_t)i * m + rc] - muB[(size_t)(valid ? j : 0) * m + rc];
        d[r] = rr ? dv : T(0);
#pragma unroll
        for (int c = 0; c <= r; ++c) {
          const int cc = c < m ? c : m - 1;
          const T v = T(0.5) * (cir[rc * m + cc] + cj[rc * m + cc]);
          row[c] = (rr && c < m) ? v : (r == c ? T(1) : T(0));
        }
      }
      // R[r][c] = (Sbar[r][c] - sum_{k<c} R[r][k] R[c][k]) / R[c][c];  R[r][r] = sqrt(Sbar[r][r] - sum_{k<r} R[r][k]^2)
#pragma unroll
      for (int c = 0; c < r; ++c) {
        T t = row[c];
#pragma unroll
        for (int k = 0; k < c; ++k) t -= row[k] * a[at(c, k)];
        row[c] = t * rdiag[c];
      }
      T piv = row[r];
#pragma unroll
      for (int k = 0; k < r; ++k) piv -= row[k] * row[k];
      const T rs = g_rsqrt<T>(piv);   // non-positive pivot -> NaN downstream, never a fault
      rdiag[r] = rs;
      const T dk = piv * rs;
      ld += g_log(dk);
#pragma unroll
      for (int c = 0; c < r; ++c) a[at(r, c)] = row[c];
      a[at(r, r)] = dk;
    }
    ld *= T(2);
    // ---- z = R^-1 delta, Q = |z|^2 ----
    T q = T(0);
#pragma unroll
    for (int k = 0; k < M; ++k) {
      T t = d[k];
#pragma unroll
      for (int c = 0; c < k; ++c) t -= a[at(k, c)] * d[c];
      d[k] = t * rdiag[k];   // d becomes z
      q += d[k] * d[k];
    }
    if (valid) {
      if (p.Q != nullptr) static_cast<T*>(p.Q)[(size_t)i * p.nB + j] = q;
      if (p.LD != nullptr) static_cast<T*>(p.LD)[(size_t)i * p.nB + j] = ld;
    }
    if (!want_grad) continue;
    const T gq = (valid && gQ != nullptr) ? gQ[(size_t)i * p.nB + j] : T(0);
    const T gl = (valid && gLD != nullptr) ? gLD[(size_t)i * p.nB + j] : T(0);
    // ---- s = R^-T z (in place, from the last row up) ----
#pragma unroll
    for (int k = M - 1; k >= 0; --k) {
      T t = d[k];
#pragma unroll
      for (int r = k + 1; r < M; ++r) t -= a[at(r, k)] * d[r];
      d[k] = t * rdiag[k];   // d becomes s
    }
    if (gLD != nullptr) {
      // ---- W = R^-1 in place, column by column: column c of W needs R[r][k] for k >= c only, so it may overwrite
      //      column c of R as soon as it is complete ----
#pragma unroll
      for (int c = 0; c < M; ++c) {
        T wc[M];
        wc[c] = rdiag[c];
#pragma unroll
        for (int r = c + 1; r < M; ++r) {
          T sum = T(0);
#pragma unroll
          for (int k = c; k < r; ++k) sum += a[at(r, k)] * wc[k];
          wc[r] = -sum * rdiag[r];
        }
#pragma unroll
        for (int r = c; r < M; ++r) a[at(r, c)] = wc[r];
      }
      // ---- Sbar^-1 = W^T W in place, row by row: row r needs the rows k >= r of W, so it may overwrite row r ----
#pragma unroll
      for (int r = 0; r < M; ++r) {
        T row[M];
#pragma unroll
        for (int c = 0; c <= r; ++c) {
          T inv = T(0);
#pragma unroll
          for (int k = r; k < M; ++k) inv += a[at(k, r)] * a[at(k, c)];
          row[c] = inv;
        }
#pragma unroll
        for (int c = 0; c <= r; ++c) a[at(r, c)] = row[c];
      }
    }
    // ---- 1/2 (gl Sbar^-1 - gq s s^T) and 2 gq s, summed over the 64 pairs of the wave: lane l finishes entries 64 I + l ----
    {
      const GradEntries<T, M> prod{a, d, gq, gl, gLD != nullptr};
      T* acc = s_acc[wave];
      tree_reduce_blocks<6, 0, (NACC + 63) / 64, NACC, T>(prod, lane, [&](int idx, T v) { acc[idx] += v; });
    }
  }
  if (!want_grad) return;
  __syncthreads();
  T* gcov = static_cast<T*>(p.gcovA) + (size_t)i * m * m;
  for (int e = tid; e < m * m; e += 256) {
    const int rr = e / m, cc = e % m;
    const int hi = rr > cc ? rr : cc, lo = rr > cc ? cc : rr;
    const int idx = hi * (hi + 1) / 2 + lo;
    gcov[e] = ((s_acc[0][idx] + s_acc[1][idx]) + s_acc[2][idx]) + s_acc[3][idx];
  }
  if (p.gmuA != nullptr) {
    T* gmu = static_cast<T*>(p.gmuA) + (size_t)i * m;
    for (int e = tid; e < m; e += 256)
      gmu[e] = ((s_acc[0][TRI + e] + s_acc[1][TRI + e]) + s_acc[2][TRI + e]) + s_acc[3][TRI + e];
  }
}

template <typename T, int M>
static hipError_t launch_gauss_reg(const GaussParams& p, hipStream_t stream) {
  const bool aligned = (reinterpret_cast<size_t>(p.covA) % 16) == 0 && (reinterpret_cast<size_t>(p.covB) % 16) == 0;
  if (p.m == M && aligned)
    hipLaunchKernelGGL((gauss_pair_reg_kernel<T, M, true>), dim3(p.nA), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((gauss_pair_reg_kernel<T, M, false>), dim3(p.nA), dim3(256), 0, stream, p);
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
#ifndef SQFA_GAUSS_REG
#define SQFA_GAUSS_REG 1   // 0: always the LDS kernel (development A/B)
#endif
#ifndef SQFA_GAUSS_REG_F64_MAX
#define SQFA_GAUSS_REG_F64_MAX 16   // float64 m > 8: the triangle needs 156 / 272 registers -- one wave per SIMD, part of it in AGPRs
#endif
  // small matrices: one pair per lane, everything in registers (float32 up to 16, float64 up to 8)
  if (SQFA_GAUSS_REG && dtype == SQFA_F32 && m <= 16) {
    if (m <= 4) e = launch_gauss_reg<float, 4>(p, stream);
    else if (m <= 8) e = launch_gauss_reg<float, 8>(p, stream);
    else if (m <= 12) e = launch_gauss_reg<float, 12>(p, stream);
    else e = launch_gauss_reg<float, 16>(p, stream);
    return e == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
  }
  if (SQFA_GAUSS_REG && dtype == SQFA_F64 && m <= SQFA_GAUSS_REG_F64_MAX) {
    if (m <= 4) e = launch_gauss_reg<double, 4>(p, stream);
    else if (m <= 8) e = launch_gauss_reg<double, 8>(p, stream);
    else if (m <= 12) e = launch_gauss_reg<double, 12>(p, stream);
    else e = launch_gauss_reg<double, 16>(p, stream);
    return e == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
  }
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
