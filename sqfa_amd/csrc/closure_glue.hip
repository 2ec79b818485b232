// closure_glue.hip -- the small elementwise/reduction steps of one closure evaluation around the three
// big kernels (projection, pair tiles, feature backward), each as ONE launch instead of the 3-8
// tiny torch kernels autograd issues for them:
//
//   sqfa_sphere_forward    F = X / ||X||_row                      (reference Sphere.forward, src/sqfa/constraints.py:37)
//   sqfa_sphere_backward   dL/dX = gloss * (gF - F (F . gF)) / ||X||, with gF = sum of the per-group
//                          partial sums of sqfa_feature_scatters_backward (+ an optional extra (K,D)
//                          term, the means path of SQFA) -- i.e. the class reduction, the gradient
//                          of the normalisation and the multiplication by the incoming loss gradient
//   sqfa_embed_backward_means   gradient wrt the projected means m_c of the Calvo-Oller embedding
//                          E = [[S + m m^T, m], [m^T, 1]] (src/sqfa/distances.py:141-174):
//                          g_m = (G + G^T) m + gE[:K,K] + gE[K,:K],  G = gE[:K,:K]
//
// One workgroup per filter row / class; fixed-order reductions (bitwise reproducible).
#include <hip/hip_runtime.h>

#include "../../include/sqfa_hip.h"

namespace sqfa {

template <typename T> __device__ __forceinline__ T block_sum_256(T v, T* s_red) {
  const int tid = threadIdx.x;
  s_red[tid] = v;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) s_red[tid] += s_red[tid + st];
    __syncthreads();
  }
  const T out = s_red[0];
  __syncthreads();
  return out;
}

template <typename T> __device__ __forceinline__ T g_sqrt(T x);
template <> __device__ __forceinline__ float g_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double g_sqrt<double>(double x) { return sqrt(x); }

template <typename T>
__global__ __launch_bounds__(256) void sphere_forward_kernel(const T* __restrict__ X, T* __restrict__ F,
                                                             T* __restrict__ norms, int D) {
  __shared__ T s_red[256];
  const int k = blockIdx.x, tid = threadIdx.x;
  const T* x = X + (size_t)k * D;
  T ss = T(0);
  for (int d = tid; d < D; d += 256) ss += x[d] * x[d];
  const T nrm = g_sqrt<T>(block_sum_256(ss, s_red));
  for (int d = tid; d < D; d += 256) F[(size_t)k * D + d] = x[d] / nrm;
  if (tid == 0) norms[k] = nrm;
}

// norms == nullptr: no constraint (Identity parametrization): dL/dX = gloss * gF
template <typename T>
__global__ __launch_bounds__(256) void sphere_backward_kernel(const T* __restrict__ X, const T* __restrict__ norms,
                                                              const T* __restrict__ partials, int n_groups,
                                                              const T* __restrict__ extra, const T* __restrict__ gloss,
                                                              T* __restrict__ out, int K, int D) {
  __shared__ T s_red[256];
  const int k = blockIdx.x, tid = threadIdx.x;
  const T scale = gloss != nullptr ? gloss[0] : T(1);
  const T* x = X + (size_t)k * D;
  const T nrm = norms != nullptr ? norms[k] : T(1);
  T dot = T(0);
  // pass 1: gF (kept in `out`) and F . gF
  for (int d = tid; d < D; d += 256) {
    T g = extra != nullptr ? extra[(size_t)k * D + d] : T(0);
    // four interleaved partial sums (fixed association: reproducible) keep the loads independent: the
    // kernel has only K workgroups and is bound by the latency of these L2 reads
    T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
    const T* pp = partials + (size_t)k * D + d;
    const size_t stride = (size_t)K * D;
    int q = 0;
#pragma unroll 2
    for (; q + 4 <= n_groups; q += 4) {
      a0 += pp[(size_t)q * stride];
      a1 += pp[(size_t)(q + 1) * stride];
      a2 += pp[(size_t)(q + 2) * stride];
      a3 += pp[(size_t)(q + 3) * stride];
    }
    for (; q < n_groups; ++q) a0 += pp[(size_t)q * stride];
    g += (a0 + a1) + (a2 + a3);
    out[(size_t)k * D + d] = g;
    dot += (x[d] / nrm) * g;
  }
  if (norms == nullptr) {
    for (int d = tid; d < D; d += 256) out[(size_t)k * D + d] *= scale;
    return;
  }
  dot = block_sum_256(dot, s_red);
  for (int d = tid; d < D; d += 256) {
    const T g = out[(size_t)k * D + d];  // written by this thread above
    out[(size_t)k * D + d] = scale * (g - (x[d] / nrm) * dot) / nrm;
  }
}

template <typename T>
__global__ __launch_bounds__(64) void embed_backward_means_kernel(const T* __restrict__ gE, const T* __restrict__ m,
                                                                  T* __restrict__ gm, int K) {
  const int c = blockIdx.x, ld = K + 1;
  const T* g = gE + (size_t)c * ld * ld;
  const T* mc = m + (size_t)c * K;
  for (int a = threadIdx.x; a < K; a += 64) {
    T acc = g[(size_t)a * ld + K] + g[(size_t)K * ld + a];
    for (int b = 0; b < K; ++b) acc += (g[(size_t)a * ld + b] + g[(size_t)b * ld + a]) * mc[b];
    gm[(size_t)c * K + a] = acc;
  }
}

}  // namespace sqfa

using namespace sqfa;

extern "C" int sqfa_sphere_forward(const void* X, int K, int D, int dtype, void* F_out, void* norms_out, void* stream_) {
  if (X == nullptr || F_out == nullptr || norms_out == nullptr || K < 1 || D < 1) return SQFA_ERR_BAD_ARGUMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (dtype == SQFA_F32)
    hipLaunchKernelGGL(sphere_forward_kernel<float>, dim3(K), dim3(256), 0, stream, static_cast<const float*>(X),
                       static_cast<float*>(F_out), static_cast<float*>(norms_out), D);
  else if (dtype == SQFA_F64)
    hipLaunchKernelGGL(sphere_forward_kernel<double>, dim3(K), dim3(256), 0, stream, static_cast<const double*>(X),
                       static_cast<double*>(F_out), static_cast<double*>(norms_out), D);
  else
    return SQFA_ERR_BAD_ARGUMENT;
  return hipGetLastError() == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}

extern "C" int sqfa_sphere_backward(const void* X, const void* norms, int K, int D, int dtype, const void* partials,
                                    int n_groups, const void* extra, const void* gloss, void* grad_out, void* stream_) {
  if (X == nullptr || grad_out == nullptr || K < 1 || D < 1 || n_groups < 0 || (n_groups > 0 && partials == nullptr))
    return SQFA_ERR_BAD_ARGUMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (dtype == SQFA_F32)
    hipLaunchKernelGGL(sphere_backward_kernel<float>, dim3(K), dim3(256), 0, stream, static_cast<const float*>(X),
                       static_cast<const float*>(norms), static_cast<const float*>(partials), n_groups,
                       static_cast<const float*>(extra), static_cast<const float*>(gloss), static_cast<float*>(grad_out), K, D);
  else if (dtype == SQFA_F64)
    hipLaunchKernelGGL(sphere_backward_kernel<double>, dim3(K), dim3(256), 0, stream, static_cast<const double*>(X),
                       static_cast<const double*>(norms), static_cast<const double*>(partials), n_groups,
                       static_cast<const double*>(extra), static_cast<const double*>(gloss), static_cast<double*>(grad_out), K, D);
  else
    return SQFA_ERR_BAD_ARGUMENT;
  return hipGetLastError() == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}

extern "C" int sqfa_embed_backward_means(const void* gE, const void* means_f, int C, int K, int dtype, void* gm_out,
                                         void* stream_) {
  if (gE == nullptr || means_f == nullptr || gm_out == nullptr || C < 1 || K < 1) return SQFA_ERR_BAD_ARGUMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (dtype == SQFA_F32)
    hipLaunchKernelGGL(embed_backward_means_kernel<float>, dim3(C), dim3(64), 0, stream, static_cast<const float*>(gE),
                       static_cast<const float*>(means_f), static_cast<float*>(gm_out), K);
  else if (dtype == SQFA_F64)
    hipLaunchKernelGGL(embed_backward_means_kernel<double>, dim3(C), dim3(64), 0, stream, static_cast<const double*>(gE),
                       static_cast<const double*>(means_f), static_cast<double*>(gm_out), K);
  else
    return SQFA_ERR_BAD_ARGUMENT;
  return hipGetLastError() == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}
