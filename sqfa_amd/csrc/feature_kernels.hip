// feature_kernels.hip -- the two small products around the streaming projection (project_kernel.hip):
//
//   forward   S_c = F T_c                         (K x K per class; T_c = Psi_c F^T, (D,K) row-major)
//   backward  dL/dF = sum_c (G_c + G_c^T) T_c^T   (K x D), as per-class-group partial sums
//
// Both read T (C,D,K) exactly once (50 MB at c3) and are HBM/latency-bound; in torch they were a
// batched GEMM, a batched GEMM writing a (C,K,D) intermediate, an elementwise add and a
// reduction over classes (~170 us per closure at c3, 10 % of it).  Exact-f32 / f64 MFMA
// 16x16x4, operand layouts in proj_traits.hpp.  Reference: the einsum of conjugate_matrix
// (src/sqfa/linalg.py:41) and its autograd backward, as used by transform_scatters
// (src/sqfa/model.py:172-188).
#include <hip/hip_runtime.h>

#include "../../include/sqfa_hip.h"
#include "proj_traits.hpp"

namespace sqfa {

// ---- forward: one workgroup per class; wave (w, part) owns output rows [16 w, 16 w + 16) and every
// SPLIT-th contraction step (the kernel is latency-bound: T_c comes from HBM, 196 dependent steps
// in one wave would leave 4 loads in flight); the SPLIT partial tiles are summed through LDS.
// step s contracts d = 4s .. 4s+3:  A[i][k] = F[16 w + i][4 s + k],  B[k][j] = T_c[4 s + k][16 nb + j]
// (for K = 16 the 64 lanes of a wave read 256 contiguous bytes of T per step).
// FV: a lane reads its four F elements of FOUR consecutive steps as one 16-byte (float64: 32-byte) load: the group
// of steps 4S .. 4S+3 then contracts d = 16 S + 4 k + e in MFMA e (k = the lane's quarter) instead of d = 4 s + k,
// same sums in a different order.  The kernel is bound by the number of load instructions (the 16-row x 16-byte
// footprint of the scalar F load costs as much as a T load): 8 -> 5 per four steps at K = 16.
template <typename T, int NB, int SPLIT, bool FV>
__global__ __launch_bounds__(64 * NB * SPLIT) void feature_scatters_kernel(const T* __restrict__ F,
                                                                           const T* __restrict__ Tm,
                                                                           T* __restrict__ S, int D, int K, T noise,
                                                                           const T* __restrict__ means, int ld) {
  using Tr = ProjTraits<T>;
  using Acc = typename Tr::Acc;
  __shared__ T s_part[SPLIT][NB][NB][4][64];
  const int c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = wave % NB, part = wave / NB;
  const int r16 = lane & 15, q = lane >> 4;
  const int row = 16 * w + r16;
  const T* __restrict__ f = F + (size_t)(row < K ? row : 0) * D + q;
  const T* __restrict__ t = Tm + (size_t)c * D * K + (size_t)q * K + r16;
  Acc acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) acc[nb][reg] = T(0);
  const int steps = D / 4;
  int first_scalar_step = part;
  if constexpr (FV) {
    struct alignas(4 * sizeof(T)) Vec4 { T v[4]; };
    const int groups = steps / 4;
    const T* __restrict__ frow = F + (size_t)(row < K ? row : 0) * D + 4 * q;
#pragma unroll 2
    for (int S = part; S < groups; S += SPLIT) {
      Vec4 a4 = {{T(0), T(0), T(0), T(0)}};
      if (row < K) a4 = *reinterpret_cast<const Vec4*>(frow + 16 * S);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const T b = (16 * nb + r16 < K) ? t[(size_t)(16 * S + 3 * q + e) * K + 16 * nb] : T(0);  // row 16 S + 4 q + e (t holds + q)
          acc[nb] = Tr::mfma(a4.v[e], b, acc[nb]);
        }
      }
    }
    first_scalar_step = 4 * groups + part;  // the D % 16 tail, one step at a time
  }
#pragma unroll 8
  for (int s = first_scalar_step; s < steps; s += SPLIT) {
    const T a = row < K ? f[4 * s] : T(0);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const T b = (16 * nb + r16 < K) ? t[(size_t)4 * s * K + 16 * nb] : T(0);
      acc[nb] = Tr::mfma(a, b, acc[nb]);
    }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) s_part[part][w][nb][reg][lane] = acc[nb][reg];
  __syncthreads();
  if (part != 0) return;
  // epilogue (fused closure glue): + noise * I (reference src/sqfa/model.py:537-538) and, when the
  // projected means m_c (K) are given, the Calvo-Oller embedding [[S + m m^T, m], [m^T, 1]]
  // (src/sqfa/distances.py:141-174) written with row pitch ld = K + 1
  T* out = S + (size_t)c * ld * ld;
  const T* mc = means != nullptr ? means + (size_t)c * K : nullptr;
  if (mc != nullptr && w == 0) {
    for (int a = lane; a < K; a += 64) {
      out[(size_t)a * ld + K] = mc[a];
      out[(size_t)K * ld + a] = mc[a];
    }
    if (lane == 0) out[(size_t)K * ld + K] = T(1);
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      T v = T(0);
#pragma unroll
      for (int p2 = 0; p2 < SPLIT; ++p2) v += s_part[p2][w][nb][reg][lane];  // fixed order
      const int a_row = 16 * w + Tr::acc_row(q, reg), b_col = 16 * nb + r16;
      if (a_row < K && b_col < K) {
        if (a_row == b_col) v += noise;
        if (mc != nullptr) v += mc[a_row] * mc[b_col];
        out[(size_t)a_row * ld + b_col] = v;
      }
    }
  }
}

// ---- backward: one wave per (16-column block of d, class group) -----------------------------
// P_g[a][d] = sum_{c in group g} sum_b (G_c[a][b] + G_c[b][a]) T_c[d][b].  Per class and chunk of 16 b's
// a lane loads 4 consecutive b's of its T row (the 16 rows of the block are 16*K contiguous
// elements) and of its G row; MFMA step e contracts b = 16 bc + 4 q + e on both operands:
//   A[i][k] = sym[16 na + i][16 bc + 4 k + e],  B[k][j] = T_c[d0 + j][16 bc + 4 k + e].
// Classes are visited in a fixed order and the groups are summed in a fixed order by the
// caller: bitwise reproducible.
// TV: K % 4 == 0, a lane's four T elements are one 16-byte load.  SYM: the caller guarantees G_c = G_c^T (the
// pair kernels' dL/dS is written symmetric), so G_c + G_c^T = 2 G_c read along rows only -- no transposed
// (uncoalesced) second read; GV: and its rows are 16-byte aligned (ldg % 4 == 0).  The kernel is bound by the
// number of load instructions per class (round 2, c4 shape: 40 dword loads per class and wave, 365 us for
// 262 MB of T): 6 with TV+SYM+GV.
template <typename T, int NB, bool TV, bool SYM, bool GV>
__global__ __launch_bounds__(64) void feature_backward_kernel(const T* __restrict__ G, const T* __restrict__ Tm,
                                                              T* __restrict__ P, int C, int D, int K, int n_groups,
                                                              int ldg) {
  using Tr = ProjTraits<T>;
  using Acc = typename Tr::Acc;
  struct alignas(4 * sizeof(T)) Vec4 { T v[4]; };
  const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
  const int d = 16 * blockIdx.x + r16, g = blockIdx.y;
  const bool d_ok = d < D;
  Acc acc[NB];
#pragma unroll
  for (int na = 0; na < NB; ++na)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) acc[na][reg] = T(0);
  for (int c = g; c < C; c += n_groups) {
    const T* __restrict__ tc = Tm + ((size_t)c * D + (d_ok ? d : 0)) * K;
    const T* __restrict__ gc = G + (size_t)c * ldg * ldg;  // (ldg, ldg) per class: K, or K + 1 for a gradient wrt the embedding
    // all loads of the class first (independent), then the MFMAs
    T tv[NB][4], sv[NB][NB][4];
#pragma unroll
    for (int bc = 0; bc < NB; ++bc) {
      const int b0 = 16 * bc + 4 * q;  // my four contraction indices b0 .. b0+3 (each guarded: any K)
      if constexpr (TV) {
        Vec4 t4 = {{T(0), T(0), T(0), T(0)}};
        if (d_ok && b0 < K) t4 = *reinterpret_cast<const Vec4*>(tc + b0);
#pragma unroll
        for (int e = 0; e < 4; ++e) tv[bc][e] = t4.v[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) tv[bc][e] = (d_ok && b0 + e < K) ? tc[b0 + e] : T(0);
      }
#pragma unroll
      for (int na = 0; na < NB; ++na) {
        const int a = 16 * na + r16;
        if constexpr (SYM && GV) {
          Vec4 g4 = {{T(0), T(0), T(0), T(0)}};
          if (a < K && b0 < K) g4 = *reinterpret_cast<const Vec4*>(gc + (size_t)a * ldg + b0);  // K % 4 == 0 with GV
#pragma unroll
          for (int e = 0; e < 4; ++e) sv[bc][na][e] = T(2) * g4.v[e];
        } else if constexpr (SYM) {
#pragma unroll
          for (int e = 0; e < 4; ++e) sv[bc][na][e] = (a < K && b0 + e < K) ? T(2) * gc[(size_t)a * ldg + b0 + e] : T(0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            sv[bc][na][e] = (a < K && b0 + e < K) ? gc[(size_t)a * ldg + b0 + e] + gc[(size_t)(b0 + e) * ldg + a] : T(0);
        }
      }
    }
#pragma unroll
    for (int bc = 0; bc < NB; ++bc)
#pragma unroll
      for (int na = 0; na < NB; ++na)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[na] = Tr::mfma(sv[bc][na][e], tv[bc][e], acc[na]);
  }
  T* out = P + (size_t)g * K * D;
#pragma unroll
  for (int na = 0; na < NB; ++na) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int a = 16 * na + Tr::acc_row(q, reg);
      if (a < K && d_ok) out[(size_t)a * D + d] = acc[na][reg];
    }
  }
}

template <typename T>
static void launch_forward(const T* f, const T* t, T* s, int C, int D, int K, T noise, const T* means, int ld,
                           hipStream_t stream) {
  // (float64 too since round 3: no product code path is kept only to match a chaotic trajectory golden any more)
  const bool fv = (reinterpret_cast<size_t>(f) % (4 * sizeof(T))) == 0 && D >= 16;  // D % 4 == 0: every row is aligned then
#define SQFA_FWD(NB_, SPLIT_, THREADS_)                                                                                     \
  if (fv)                                                                                                                   \
    hipLaunchKernelGGL((feature_scatters_kernel<T, NB_, SPLIT_, true>), dim3(C), dim3(THREADS_), 0, stream, f, t, s, D, K, \
                       noise, means, ld);                                                                                   \
  else                                                                                                                      \
    hipLaunchKernelGGL((feature_scatters_kernel<T, NB_, SPLIT_, false>), dim3(C), dim3(THREADS_), 0, stream, f, t, s, D, K, \
                       noise, means, ld);
  switch ((K + 15) / 16) {
    case 1: SQFA_FWD(1, 8, 512) break;
    case 2: SQFA_FWD(2, 4, 512) break;
    case 3: SQFA_FWD(3, 2, 384) break;
    default: SQFA_FWD(4, 2, 512) break;
  }
#undef SQFA_FWD
}

template <typename T, bool TV, bool SYM, bool GV>
static void launch_backward_v(const T* g, const T* t, T* p, int C, int D, int K, int n_groups, int ldg, hipStream_t stream) {
  const dim3 grid((D + 15) / 16, n_groups, 1), block(64);
  switch ((K + 15) / 16) {
    case 1: hipLaunchKernelGGL((feature_backward_kernel<T, 1, TV, SYM, GV>), grid, block, 0, stream, g, t, p, C, D, K, n_groups, ldg); break;
    case 2: hipLaunchKernelGGL((feature_backward_kernel<T, 2, TV, SYM, GV>), grid, block, 0, stream, g, t, p, C, D, K, n_groups, ldg); break;
    case 3: hipLaunchKernelGGL((feature_backward_kernel<T, 3, TV, SYM, GV>), grid, block, 0, stream, g, t, p, C, D, K, n_groups, ldg); break;
    default: hipLaunchKernelGGL((feature_backward_kernel<T, 4, TV, SYM, GV>), grid, block, 0, stream, g, t, p, C, D, K, n_groups, ldg); break;
  }
}

template <typename T>
static void launch_backward(const T* g, const T* t, T* p, int C, int D, int K, int n_groups, int ldg, bool symmetric,
                            hipStream_t stream) {
  constexpr size_t VB = 4 * sizeof(T);
  const bool tvec = (K % 4) == 0 && (reinterpret_cast<size_t>(t) % VB) == 0;
  const bool gvec = symmetric && tvec && (ldg % 4) == 0 && (reinterpret_cast<size_t>(g) % VB) == 0;
  if (tvec && gvec) launch_backward_v<T, true, true, true>(g, t, p, C, D, K, n_groups, ldg, stream);
  else if (tvec && symmetric) launch_backward_v<T, true, true, false>(g, t, p, C, D, K, n_groups, ldg, stream);
  else if (tvec) launch_backward_v<T, true, false, false>(g, t, p, C, D, K, n_groups, ldg, stream);
  else if (symmetric) launch_backward_v<T, false, true, false>(g, t, p, C, D, K, n_groups, ldg, stream);
  else launch_backward_v<T, false, false, false>(g, t, p, C, D, K, n_groups, ldg, stream);
}

static bool shape_ok(int K, int D, int C, int dtype) {
  return K >= 1 && C >= 1 && D >= 4 && (dtype == SQFA_F32 || dtype == SQFA_F64);
}

}  // namespace sqfa

extern "C" int sqfa_feature_scatters_ex(const void* F, int K, int D, const void* T, int C, int dtype, double noise,
                                        const void* means_f, void* S_out, void* stream_) {
  using namespace sqfa;
  if (F == nullptr || T == nullptr || S_out == nullptr || !shape_ok(K, D, C, dtype)) return SQFA_ERR_BAD_ARGUMENT;
  if ((D % 4) != 0 || K > 64) return SQFA_ERR_UNSUPPORTED_M;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int ld = means_f != nullptr ? K + 1 : K;
  if (dtype == SQFA_F32)
    launch_forward(static_cast<const float*>(F), static_cast<const float*>(T), static_cast<float*>(S_out), C, D, K,
                   (float)noise, static_cast<const float*>(means_f), ld, stream);
  else
    launch_forward(static_cast<const double*>(F), static_cast<const double*>(T), static_cast<double*>(S_out), C, D, K,
                   noise, static_cast<const double*>(means_f), ld, stream);
  return hipGetLastError() == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}

extern "C" int sqfa_feature_scatters(const void* F, int K, int D, const void* T, int C, int dtype, void* S_out,
                                     void* stream_) {
  return sqfa_feature_scatters_ex(F, K, D, T, C, dtype, 0.0, nullptr, S_out, stream_);
}

extern "C" int sqfa_feature_scatters_backward_ex(const void* G, int ldg, const void* T, int C, int D, int K, int dtype,
                                                 int n_groups, int g_symmetric, void* partial_out, void* stream_) {
  using namespace sqfa;
  if (G == nullptr || T == nullptr || partial_out == nullptr || !shape_ok(K, D, C, dtype) || n_groups < 1 || ldg < K)
    return SQFA_ERR_BAD_ARGUMENT;
  if (K > 64) return SQFA_ERR_UNSUPPORTED_M;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (dtype == SQFA_F32)
    launch_backward(static_cast<const float*>(G), static_cast<const float*>(T), static_cast<float*>(partial_out), C, D, K,
                    n_groups, ldg, g_symmetric != 0, stream);
  else
    launch_backward(static_cast<const double*>(G), static_cast<const double*>(T), static_cast<double*>(partial_out), C, D,
                    K, n_groups, ldg, g_symmetric != 0, stream);
  return hipGetLastError() == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}

extern "C" int sqfa_feature_scatters_backward(const void* G, const void* T, int C, int D, int K, int dtype, int n_groups,
                                              void* partial_out, void* stream_) {
  return sqfa_feature_scatters_backward_ex(G, K, T, C, D, K, dtype, n_groups, 0, partial_out, stream_);
}
