// RECORDED EXPERIMENT (round 3), not part of the build: the shipped projection kernel plus ALIGN2 -- separate, 128-byte
// aligned column partitions (and accumulators) for even and odd rows when rows are an odd multiple of 64 bytes
// (float32, D % 32 == 16: D=784), exchanged through LDS in the epilogue.  Correct; L2 requests -10 %, L2 misses -3 %
// (gpurun_out/r3/pmc_proj, rocprofv3 --pmc TCC_REQ_sum / TCC_MISS_sum), but 0.462 vs 0.447 ms at D=784 and equal at D=816 /
// 1008: the 11 % deficit of D % 32 == 16 against its neighbours (D=768 6.2, 784 5.5, 800 6.2, 816 5.6, 832 6.1 TB/s) is
// not the line straddling of the wave segments.
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
// so a wave reads 4 rows x 256 contiguous bytes and the WAVES (4/8/16) waves of a workgroup
// (adjacent stripes) read 4 rows x WAVES*256 B -- long contiguous runs whatever the row pitch (a 16-row x 64-B
// footprint camped on a few HBM channels when D*4 is a multiple of 4 KiB).  Component j of
// that float4 is the B operand of MFMA j (output columns d0 + 4*i + j), the A operand is
// F[n = l&15][k = 4s + (l>>4)], one ds_read_b32 from the F chunk staged in LDS as [k][n].
// float64 uses v_mfma_f64_16x16x4_f64 with 2 columns per lane (same 16-byte loads) and that
// instruction's own C/D row map (row = (l>>4) + 4*reg).
//
// Round 3, rows that are an odd multiple of 64 bytes long (float32, D % 32 == 16: D = 784, the MNIST-sized c3):
// every second row starts in the middle of a 128-byte line, so the 256-byte segments of the odd rows straddle
// three lines instead of two (measured with this kernel at K=16, C=1000: D=768 6.19, D=784 5.76, D=800 6.23 TB/s;
// a ragged last stripe costs nothing: D=800 is 12.5 stripes; PMC over-fetch at D=784 11.5 %).  ALIGN2 gives the
// odd rows their own column partition, shifted by 16 columns (= 64 bytes), and their own accumulators: even rows
// are read in stripes [64 s, 64 s + 64), odd rows in stripes [64 s - 16, 64 s + 48) -- every segment of every row
// is then line-aligned.  The two partial results of a column live in different lanes (and, at stripe borders, in
// different waves of the workgroup): the odd-row part travels through LDS once per class, in the epilogue.
// A barrier-free variant (every wave its own stream, F straight from L2, no LDS) was built and measured first:
// slower everywhere (c3 0.515-0.557 vs 0.465 ms, c4 3.48 vs 2.98) -- the barriers keep the waves of a workgroup
// on the same rows, and HBM rewards that (kept as tools/experiments/project_kernel_barrierfree.hip).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <mutex>

#include <utility>
#include <vector>

#include "../../include/sqfa_hip.h"
#include "proj_traits.hpp"

bool sqfa_profile_enabled();                                            // sqfa_api.hip
std::vector<std::pair<hipEvent_t, hipEvent_t>>& sqfa_project_events();  // sqfa_api.hip
std::mutex& sqfa_project_events_mutex();                                 // sqfa_api.hip

#ifndef SQFA_PROJ_VEC_STORE
#define SQFA_PROJ_VEC_STORE 1
#endif

namespace sqfa {

// ---- ALIGN2: separate, line-aligned column partitions for even and odd rows (see the header) ----------------
// One workgroup holds every stripe of a class (the launcher guarantees nstripes <= WAVES), float32 only.
extern __shared__ __attribute__((aligned(16))) unsigned char sqfa_proj_dyn_lds[];
template <typename T, int NB, int KC, int WAVES>
__device__ __forceinline__ void project_body_align2(const T* __restrict__ F, const T* __restrict__ Psi,
                                                    T* __restrict__ Tout, int C, int D, int K) {
  using Tr = ProjTraits<T>;
  using Vec = typename Tr::Vec;
  using Acc = typename Tr::Acc;
  static_assert(sizeof(T) == 4 && Tr::VW == 4, "ALIGN2 is the float32 path");
  static_assert(KC % 8 == 0, "rows are processed in blocks of 8");
  constexpr int VW = 4, SW = 64, KP = 16 * NB;
  // one dynamic LDS region, used twice: during the stream the F chunks, double buffered ([buf][k][n]; inside every
  // block of 8 rows the 4 even rows come first, then the 4 odd rows); in the epilogue the odd-row partial results of
  // the class, [D][KP] (the launcher sizes the region for the larger of the two)
  T (*s_f)[KC][KP] = reinterpret_cast<T (*)[KC][KP]>(sqfa_proj_dyn_lds);
  T* s_x = reinterpret_cast<T*>(sqfa_proj_dyn_lds);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r16 = lane & 15, q = lane >> 4;
  const int c = blockIdx.y;
  constexpr int NT = 64 * WAVES;
  const int nstripes = (D + SW - 1) / SW;
  const bool active = wave < nstripes;      // idle waves still help staging F
  const int de = wave * SW + VW * r16;      // my even-row columns de .. de+3
  const int dodd = de - 16;                 // my odd-row columns
  const bool e_ok = active && de < D, o_ok = active && dodd >= 0 && dodd < D;
  const T* __restrict__ base = Psi + (size_t)c * D * D;
  const T* __restrict__ pe = base + (e_ok ? de : 0);     // lanes without columns repeat a valid address; results discarded
  const T* __restrict__ po = base + (o_ok ? dodd : 0);
  const int nchunks = (D + KC - 1) / KC;

  auto stage = [&](int chunk, int buf) {
    for (int e = tid; e < KC * KP; e += NT) {
      const int n = e % KP, kk = e / KP;
      const int k = chunk * KC + kk;
      const int slot = (kk & ~7) + ((kk & 1) << 2) + ((kk & 7) >> 1);
      s_f[buf][slot][n] = (n < K && k < D) ? F[(size_t)n * D + k] : T(0);
    }
  };

  Acc acc_e[NB][VW], acc_o[NB][VW];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int j = 0; j < VW; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) acc_e[nb][j][reg] = acc_o[nb][j][reg] = T(0);

  stage(0, 0);
  __syncthreads();
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int buf = chunk & 1;
    if (chunk + 1 < nchunks) stage(chunk + 1, buf ^ 1);
    if (active) {
      const int kbase = chunk * KC;
#pragma unroll 2
      for (int t = 0; t < KC / 8; ++t) {
        int ke = kbase + 8 * t + 2 * q, ko = ke + 1;
        if (ke > D - 1) ke = D - 1;  // past the end: F is zero there, any valid row will do
        if (ko > D - 1) ko = D - 1;
#ifdef SQFA_PROJ_PLAIN_LOADS
        const Vec be = *reinterpret_cast<const Vec*>(pe + (size_t)ke * D);
        const Vec bo = *reinterpret_cast<const Vec*>(po + (size_t)ko * D);
#else
        const Vec be = __builtin_nontemporal_load(reinterpret_cast<const Vec*>(pe + (size_t)ke * D));
        const Vec bo = __builtin_nontemporal_load(reinterpret_cast<const Vec*>(po + (size_t)ko * D));
#endif
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const T ae = s_f[buf][8 * t + q][nb * 16 + r16];
          const T ao = s_f[buf][8 * t + 4 + q][nb * 16 + r16];
#pragma unroll
          for (int j = 0; j < VW; ++j) {
            acc_e[nb][j] = Tr::mfma(ae, be[j], acc_e[nb][j]);
            acc_o[nb][j] = Tr::mfma(ao, bo[j], acc_o[nb][j]);
          }
        }
      }
    }
    __syncthreads();
  }
  // acc_x[nb][j][reg] = partial T^T[n = 16 nb + 4 q + reg][d = dx + j]; a lane's four registers are T[d][4q .. 4q+3]
  struct alignas(16) V4 { T v[4]; };
  if (o_ok) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int j = 0; j < VW; ++j)
        *reinterpret_cast<V4*>(s_x + (size_t)(dodd + j) * KP + nb * 16 + 4 * q) =
            V4{{acc_o[nb][j][0], acc_o[nb][j][1], acc_o[nb][j][2], acc_o[nb][j][3]}};
  }
  __syncthreads();
  if (!e_ok) return;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const int d = de + j;   // < D: D % 4 == 0
      const V4 o = *reinterpret_cast<const V4*>(s_x + (size_t)d * KP + nb * 16 + 4 * q);
      T* out = Tout + ((size_t)c * D + d) * K;
      const int n0 = nb * 16 + 4 * q;
      V4 r = {{acc_e[nb][j][0] + o.v[0], acc_e[nb][j][1] + o.v[1], acc_e[nb][j][2] + o.v[2], acc_e[nb][j][3] + o.v[3]}};
      if ((K & 3) == 0 && (reinterpret_cast<size_t>(Tout) & 15) == 0) {
        if (n0 < K) *reinterpret_cast<V4*>(out + n0) = r;
      } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          if (n0 + reg < K) out[n0 + reg] = r.v[reg];
      }
    }
  }
}

// (ALIGN2: two accumulator sets; the register budget of TWO 1024-thread workgroups per CU -- 64 VGPRs -- is requested
// explicitly: at 68 the kernel ran one workgroup per CU and lost 10 %)
template <typename T, int NB, int KC, int WAVES, bool ALIGN2 = false>
__global__ __launch_bounds__(64 * WAVES, (ALIGN2 && NB == 1) ? 8 : 1) void project_kernel(const T* __restrict__ F, const T* __restrict__ Psi,
                                                      T* __restrict__ Tout, int C, int D, int K) {
  if constexpr (ALIGN2) {
    project_body_align2<T, NB, KC, WAVES>(F, Psi, Tout, C, D, K);
    return;
  }
  using Tr = ProjTraits<T>;
  using Vec = typename Tr::Vec;
  using Acc = typename Tr::Acc;
  constexpr int VW = Tr::VW, SW = 16 * VW;  // stripe width (columns per wave)
  __shared__ T s_f[2][KC][16 * NB];  // F chunk, double buffered: [buf][k][n]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r16 = lane & 15, q = lane >> 4;
  const int c = blockIdx.y;
  const int nstripes = (D + SW - 1) / SW;
  constexpr int NT = 64 * WAVES;
  const int stripe = blockIdx.x * WAVES + wave;
  const bool active = stripe < nstripes;  // idle waves still help staging F
  int dcol = stripe * SW + VW * r16;
  if (dcol > D - VW) dcol = D - VW;        // clamped columns are computed and thrown away
  const T* __restrict__ pc = Psi + (size_t)c * D * D + dcol;
  const int nchunks = (D + KC - 1) / KC;

  auto stage = [&](int chunk, int buf) {
    for (int e = tid; e < KC * 16 * NB; e += NT) {
      const int n = e % (16 * NB), kk = e / (16 * NB);
      const int k = chunk * KC + kk;
      s_f[buf][kk][n] = (n < K && k < D) ? F[(size_t)n * D + k] : T(0);
    }
  };

  Acc acc[NB][VW];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int j = 0; j < VW; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) acc[nb][j][reg] = T(0);

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
#ifdef SQFA_PROJ_PLAIN_LOADS
        const Vec b = *reinterpret_cast<const Vec*>(pc + (size_t)k * D);
#else
        const Vec b = __builtin_nontemporal_load(reinterpret_cast<const Vec*>(pc + (size_t)k * D));
#endif
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const T a = s_f[buf][4 * s + q][nb * 16 + r16];
#pragma unroll
          for (int j = 0; j < VW; ++j) acc[nb][j] = Tr::mfma(a, b[j], acc[nb][j]);
        }
      }
    }
    __syncthreads();
  }
  if (!active) return;
  // acc[nb][j][reg] = T^T[n = 16 nb + acc_row(q, reg)][d = stripe*SW + VW*r16 + j]
  const int d0 = stripe * SW + VW * r16;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const int d = d0 + j;
      if (d < D) {
        T* out = Tout + ((size_t)c * D + d) * K;
        if constexpr (sizeof(T) == 4) {
          // float32: the four accumulator registers of a lane are T[d][4q .. 4q+3], one 16-byte store when K % 4 == 0
          // (T_out rows are then 16-byte aligned); 16 scattered 4-byte stores per lane otherwise
          const int n0 = nb * 16 + 4 * q;
          if ((K & 3) == 0 && (reinterpret_cast<size_t>(Tout) & 15) == 0 && SQFA_PROJ_VEC_STORE) {
            if (n0 < K) {
              struct alignas(16) V4 { T v[4]; };
              V4 o = {{acc[nb][j][0], acc[nb][j][1], acc[nb][j][2], acc[nb][j][3]}};
              *reinterpret_cast<V4*>(out + n0) = o;
            }
            continue;
          }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int n = nb * 16 + Tr::acc_row(q, reg);
          if (n < K) out[n] = acc[nb][j][reg];
        }
      }
    }
  }
}

template <typename T, int WV>
static void launch_project_w(const T* f, const T* p, T* t, int C, int D, int K, hipStream_t stream) {
  constexpr int SW = 16 * ProjTraits<T>::VW;
  const dim3 grid(((D + SW - 1) / SW + WV - 1) / WV, C, 1), block(64 * WV);
  const int nb = (K + 15) / 16;
  constexpr int S = sizeof(T) / 4;  // keep the LDS chunk at the float32 byte size
  switch (nb) {
    case 1: hipLaunchKernelGGL((project_kernel<T, 1, 128 / S, WV>), grid, block, 0, stream, f, p, t, C, D, K); break;
    case 2: hipLaunchKernelGGL((project_kernel<T, 2, 64 / S, WV>), grid, block, 0, stream, f, p, t, C, D, K); break;
    case 3: hipLaunchKernelGGL((project_kernel<T, 3, 32 / S, WV>), grid, block, 0, stream, f, p, t, C, D, K); break;
    default: hipLaunchKernelGGL((project_kernel<T, 4, 32 / S, WV>), grid, block, 0, stream, f, p, t, C, D, K); break;
  }
}

// Workgroup width: the waves of a workgroup read adjacent stripes, i.e. WV*256 contiguous bytes
// of 4 consecutive rows per step.  Measured on MI355X (tools/time_variants_proj.py): throughput
// grows with the contiguous run (1 wave 2.2 TB/s, 4 waves 3.8-6.1, 16 waves = whole 3 KiB rows
// at D=784: 4.6-5.2) until a class needs several workgroups anyway, where narrower ones balance better.
#ifndef SQFA_PROJ_ALIGN2
#define SQFA_PROJ_ALIGN2 1
#endif
// float32 rows of an odd multiple of 64 bytes, one workgroup per class, K <= 32, the odd-row exchange buffer fits in
// LDS next to a second workgroup, class matrices line-aligned: the two-partition kernel (see the header)
static bool launch_project_align2(const float* f, const float* p, float* t, int C, int D, int K, hipStream_t stream) {
  const int nstripes = (D + 63) / 64, nb = (K + 15) / 16;
  const size_t stage_bytes = 2 * (size_t)(nb == 1 ? 128 : 64) * 16 * nb * sizeof(float);
  const size_t xbytes = std::max((size_t)D * 16 * nb * sizeof(float), stage_bytes);
  if (!SQFA_PROJ_ALIGN2 || D % 32 != 16 || nstripes > 16 || nb > 2 || xbytes > 60 * 1024 ||
      (reinterpret_cast<size_t>(p) % 128) != 0)
    return false;
  const dim3 grid(1, C, 1), block(64 * 16);
  if (nb == 1) hipLaunchKernelGGL((project_kernel<float, 1, 128, 16, true>), grid, block, xbytes, stream, f, p, t, C, D, K);
  else hipLaunchKernelGGL((project_kernel<float, 2, 64, 16, true>), grid, block, xbytes, stream, f, p, t, C, D, K);
  return true;
}
static bool launch_project_align2(const double*, const double*, double*, int, int, int, hipStream_t) { return false; }

template <typename T>
static void launch_project(const T* f, const T* p, T* t, int C, int D, int K, hipStream_t stream) {
  constexpr int SW = 16 * ProjTraits<T>::VW;
  const int nstripes = (D + SW - 1) / SW;
  if (launch_project_align2(f, p, t, C, D, K, stream)) return;
#ifndef SQFA_PROJ_WV_SMALL
#define SQFA_PROJ_WV_SMALL 16
#endif
  if (nstripes <= 16) launch_project_w<T, SQFA_PROJ_WV_SMALL>(f, p, t, C, D, K, stream);
  else if (nstripes <= 32) launch_project_w<T, 8>(f, p, t, C, D, K, stream);
  else launch_project_w<T, 4>(f, p, t, C, D, K, stream);
}

}  // namespace sqfa

extern "C" int sqfa_project_scatters(const void* F, int K, int D, const void* Psi, int C, int dtype, void* T_out,
                                     void* stream_) {
  using namespace sqfa;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (F == nullptr || Psi == nullptr || T_out == nullptr || K < 1 || D < 4 || C < 1) return SQFA_ERR_BAD_ARGUMENT;
  if ((dtype != SQFA_F32 && dtype != SQFA_F64) || (D % 4) != 0 || K > 64 || K > D) return SQFA_ERR_UNSUPPORTED_M;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool prof = sqfa_profile_enabled();
  if (prof) {  // no event records inside a captured graph
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) prof = false;
  }
  if (prof) {
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, stream);
  }
  if (dtype == SQFA_F32)
    launch_project(static_cast<const float*>(F), static_cast<const float*>(Psi), static_cast<float*>(T_out), C, D, K, stream);
  else
    launch_project(static_cast<const double*>(F), static_cast<const double*>(Psi), static_cast<double*>(T_out), C, D, K, stream);
  if (prof) {
    (void)hipEventRecord(e1, stream);
    std::lock_guard<std::mutex> lock(sqfa_project_events_mutex());
    sqfa_project_events().emplace_back(e0, e1);
  }
  return hipGetLastError() == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}
