// project_kernel.hip -- T_c = Psi_c F^T for all classes: the HBM-bound half of the projection
// S_c = F Psi_c F^T of the class scatter matrices into feature space (reference:
// conjugate_matrix, src/sqfa/linalg.py:19-45, as called by transform_scatters,
// src/sqfa/model.py:172-188).  Psi (C,D,D) is streamed from HBM exactly once; the small
// products S_c = F T_c and dL/dF = sum_c (G_c + G_c^T) T_c^T only touch T (C,D,K) and stay in
// torch (sqfa_amd/_native.py:ProjectScatters).
//
// Roofline: HBM.  Algorithmic traffic 4*C*D^2 bytes (+4*C*D*K written); arithmetic
// intensity K/2 flop/byte < 19.7 flop/byte ridge for every K <= 64, so the exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32, same peak as the f32 VALU) has 3x headroom at K = 16.
//
// Decomposition (T^T = F Psi, using Psi = Psi^T): a wave owns a stripe of 64 consecutive
// columns d of one class and all K (padded to 16*NB) filters; it walks DOWN the rows k of Psi.
// Per step every lane loads ONE float4: lane l reads Psi[k = 4s + (l>>4)][d0 + 4*(l&15) .. +3],
// so a wave reads 4 rows x 256 contiguous bytes and the 4 waves of a workgroup (adjacent
// stripes) read 4 rows x 1 KiB -- long contiguous runs whatever the row pitch (a 16-row x 64-B
// footprint camped on a few HBM channels when D*4 is a multiple of 4 KiB).  Component j of
// that float4 is the B operand of MFMA j (output columns d0 + 4*i + j), the A operand is
// F[n = l&15][k = 4s + (l>>4)], one ds_read_b32 from the F chunk staged in LDS as [k][n].
// The C/D layout (rows n = 4*(l>>4)+reg) makes the four accumulator registers of one MFMA a
// contiguous float4 of T[d][4q..4q+3].
#include <hip/hip_runtime.h>

#include <utility>
#include <vector>

#include "../../include/sqfa_hip.h"

bool sqfa_profile_enabled();                                            // sqfa_api.hip
std::vector<std::pair<hipEvent_t, hipEvent_t>>& sqfa_project_events();  // sqfa_api.hip

namespace sqfa {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NB, int KC>
__global__ __launch_bounds__(256) void project_kernel(const float* __restrict__ F, const float* __restrict__ Psi,
                                                      float* __restrict__ T, int C, int D, int K) {
  __shared__ float s_f[2][KC][16 * NB];  // F chunk, double buffered: [buf][k][n]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r16 = lane & 15, q = lane >> 4;
  const int c = blockIdx.y;
  const int nstripes = (D + 63) / 64;
  const int stripe = blockIdx.x * 4 + wave;
  const bool active = stripe < nstripes;  // idle waves still help staging F
  int dcol = stripe * 64 + 4 * r16;
  if (dcol > D - 4) dcol = D - 4;          // clamped columns are computed and thrown away
  const float* __restrict__ pc = Psi + (size_t)c * D * D + dcol;
  const int nchunks = (D + KC - 1) / KC;

  auto stage = [&](int chunk, int buf) {
    for (int e = tid; e < KC * 16 * NB; e += 256) {
      const int n = e % (16 * NB), kk = e / (16 * NB);
      const int k = chunk * KC + kk;
      s_f[buf][kk][n] = (n < K && k < D) ? F[(size_t)n * D + k] : 0.f;
    }
  };

  f32x4 acc[NB][4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[nb][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  __syncthreads();
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int buf = chunk & 1;
    if (chunk + 1 < nchunks) stage(chunk + 1, buf ^ 1);
    if (active) {
      const int kbase = chunk * KC;
#pragma unroll 8
      for (int s = 0; s < KC / 4; ++s) {
        int k = kbase + 4 * s + q;
        if (k > D - 1) k = D - 1;  // past the end: F is zero there, any finite row will do
        const f32x4 b = *reinterpret_cast<const f32x4*>(pc + (size_t)k * D);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float a = s_f[buf][4 * s + q][nb * 16 + r16];
          acc[nb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.x, acc[nb][0], 0, 0, 0);
          acc[nb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.y, acc[nb][1], 0, 0, 0);
          acc[nb][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.z, acc[nb][2], 0, 0, 0);
          acc[nb][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.w, acc[nb][3], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  if (!active) return;
  // acc[nb][j][reg] = T^T[n = 16 nb + 4q + reg][d = stripe*64 + 4*r16 + j]
  const int d0 = stripe * 64 + 4 * r16;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n0 = nb * 16 + 4 * q;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int d = d0 + j;
      if (d < D) {
        float* out = T + ((size_t)c * D + d) * K + n0;
        if (n0 + 3 < K && (K % 4) == 0) {
          *reinterpret_cast<f32x4*>(out) = acc[nb][j];
        } else {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg)
            if (n0 + reg < K) out[reg] = acc[nb][j][reg];
        }
      }
    }
  }
}

}  // namespace sqfa

extern "C" int sqfa_project_scatters(const void* F, int K, int D, const void* Psi, int C, int dtype, void* T_out,
                                     void* stream_) {
  using namespace sqfa;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (F == nullptr || Psi == nullptr || T_out == nullptr || K < 1 || D < 4 || C < 1) return SQFA_ERR_BAD_ARGUMENT;
  if (dtype != SQFA_F32 || (D % 4) != 0 || K > 64 || K > D) return SQFA_ERR_UNSUPPORTED_M;
  const dim3 grid(((D + 63) / 64 + 3) / 4, C, 1), block(256);
  const float* f = static_cast<const float*>(F);
  const float* p = static_cast<const float*>(Psi);
  float* t = static_cast<float*>(T_out);
  const int nb = (K + 15) / 16;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool prof = sqfa_profile_enabled();
  if (prof) {
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, stream);
  }
  switch (nb) {
    case 1: hipLaunchKernelGGL((project_kernel<1, 128>), grid, block, 0, stream, f, p, t, C, D, K); break;
    case 2: hipLaunchKernelGGL((project_kernel<2, 64>), grid, block, 0, stream, f, p, t, C, D, K); break;
    case 3: hipLaunchKernelGGL((project_kernel<3, 32>), grid, block, 0, stream, f, p, t, C, D, K); break;
    default: hipLaunchKernelGGL((project_kernel<4, 32>), grid, block, 0, stream, f, p, t, C, D, K); break;
  }
  if (prof) {
    (void)hipEventRecord(e1, stream);
    sqfa_project_events().emplace_back(e0, e1);
  }
  return hipGetLastError() == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}
