// proj_traits.hpp -- per-dtype pieces shared by the projection kernels: the 16-byte load (VW
// elements), the exact MFMA (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64) and its C/D row map.
// Operand layout of both instructions: A[i][k] from lane (i = l & 15, k = l >> 4), B[k][j] from
// lane (j = l & 15, k = l >> 4); D[acc_row(l >> 4, reg)][l & 15] in register `reg` of lane l.
#pragma once
#include <hip/hip_runtime.h>

namespace sqfa {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f64x2 = __attribute__((ext_vector_type(2))) double;
using f64x4 = __attribute__((ext_vector_type(4))) double;

// per-dtype pieces: the 16-byte load (VW elements), the exact MFMA, and the C/D row map
template <typename T> struct ProjTraits;
template <> struct ProjTraits<float> {
  using Vec = f32x4;
  using Acc = f32x4;
  static constexpr int VW = 4;
  static __device__ __forceinline__ Acc mfma(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int acc_row(int q, int reg) { return 4 * q + reg; }  // v_mfma_f32_16x16x4_f32
};
template <> struct ProjTraits<double> {
  using Vec = f64x2;
  using Acc = f64x4;
  static constexpr int VW = 2;
  static __device__ __forceinline__ Acc mfma(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int acc_row(int q, int reg) { return q + 4 * reg; }  // v_mfma_f64_16x16x4_f64
};

}  // namespace sqfa
