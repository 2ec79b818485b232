// RECORDED EXPERIMENT (round 3), not part of the build: the projection kernel with every wave its own stream -- no LDS,
// no barriers, F straight from L1/L2, register double buffer.  Correct (same results), slower than the shipped
// kernel on every shape measured in one session (MI355X, ms): c3 D=784 K=16 0.515-0.557 vs 0.465; c4 D=2048 K=32 3.48 vs
// 2.98; c5 D=3072 K=16 0.612 vs 0.580; D=1024 equal (0.630).  Deeper prefetch (2, 3 blocks) and exact 56-column stripes
// were slower still.  Reading: the barriers keep the waves of a workgroup on the same rows, and HBM rewards that.
// project_kernel.hip -- T_c = Psi_c F^T for all classes: the HBM-bound half of the projection
// S_c = F Psi_c F^T of the class scatter matrices into feature space (reference:
// conjugate_matrix, src/sqfa/linalg.py:19-45, as called by transform_scatters,
// src/sqfa/model.py:172-188).  Psi (C,D,D) is streamed from HBM exactly once; the small
// products S_c = F T_c and dL/dF = sum_c (G_c + G_c^T) T_c^T only touch T (C,D,K)
// (feature_kernels.hip).
//
// Roofline: HBM.  Algorithmic traffic 4*C*D^2 bytes (+4*C*D*K written); arithmetic
// intensity K/2 flop/byte < 19.7 flop/byte ridge for every K <= 64, so the exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32, same peak as the f32 VALU) has 3x headroom at K = 16.
//
// Round 3: every wave is its own stream -- no LDS, no barriers.  (Rounds 1-2 staged chunks of F in LDS behind a
// workgroup barrier per 128 rows: at D=784 a 16-wave workgroup drained its load pipeline seven times per class and
// three of its wave slots only staged F: 0.69 of 8 TB/s against 0.80 at D=3072.)
//
// Decomposition (T^T = F Psi, using Psi = Psi^T): a wave owns a stripe of LS*VW <= 64 consecutive columns d of one
// class (LS <= 16 lanes of each 16-lane group carry columns; the host picks the stripe count so that LS * stripes
// covers a row exactly when it can: D=784 -> 14 stripes of 56 columns instead of 12 1/4 of 64) and all K (padded to
// 16*NB) filters; it walks DOWN the rows of Psi in blocks of 16.  Per block a lane (r16 = l & 15, q = l >> 4)
// loads ONE 16-byte vector of F per 16 filters -- F[n = 16 nb + r16][k = 16 blk + 4 q + e], e = 0..3, the A operands of
// the block's four MFMA steps -- and four 16-byte vectors of Psi: at step e the rows k = 16 blk + 4 q + e of its
// columns, whose component j is the B operand of MFMA j (output columns d0 + VW*i + j).  The wave reads
// 4 x LS*16 contiguous bytes per step and the workgroup's waves (adjacent stripes) read whole rows.  F (K*D*4 bytes)
// comes from L1/L2.  The loads of block b+1 are issued before the MFMAs of block b (register double buffer).
// float64 uses v_mfma_f64_16x16x4_f64 with 2 columns per lane (same 16-byte loads) and that
// instruction's own C/D row map (row = (l>>4) + 4*reg).
#include <hip/hip_runtime.h>
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
#ifndef SQFA_PROJ_DEPTH
#define SQFA_PROJ_DEPTH 1   // blocks of 16 rows requested ahead of the one being multiplied
#endif

namespace sqfa {

// FV: F rows are 16-byte aligned (one vector load per lane and 16 filters); otherwise four scalar loads
template <typename T, int NB, int WAVES, bool FV>
__global__ __launch_bounds__(64 * WAVES) void project_kernel(const T* __restrict__ F, const T* __restrict__ Psi,
                                                      T* __restrict__ Tout, int C, int D, int K, int nstripes, int LS) {
  using Tr = ProjTraits<T>;
  using Vec = typename Tr::Vec;
  using Acc = typename Tr::Acc;
  constexpr int VW = Tr::VW;
  struct alignas(4 * sizeof(T)) F4 { T v[4]; };
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r16 = lane & 15, q = lane >> 4;
  const int c = blockIdx.y;
  const int stripe = blockIdx.x * WAVES + wave;
  if (stripe >= nstripes) return;  // no barriers below: surplus waves simply leave
  const int d0 = (stripe * LS + r16) * VW;
  const bool has_cols = r16 < LS && d0 < D;   // D % VW == 0: a lane's VW columns are all inside or all outside
  const int dcol = has_cols ? d0 : stripe * LS * VW;  // idle lanes repeat the stripe's first address (same cache line)
  const T* __restrict__ pc = Psi + (size_t)c * D * D + dcol;
  const T* fr[NB];
  bool fok[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = 16 * nb + r16;
    fok[nb] = n < K;
    fr[nb] = F + (size_t)(fok[nb] ? n : K - 1) * D;
  }
  const int nblk = (D + 15) / 16;

  struct Block { Vec b[4]; F4 a[NB]; };
  auto load_block = [&](int blk, Block& o) {
    const int k0 = 16 * blk + 4 * q;
    const bool in = k0 < D;               // D % 4 == 0: the four k of a lane are all inside or all outside
    const int kc = in ? k0 : D - 4;       // outside: any valid rows, their A operands are zero
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      F4 v;
      if constexpr (FV) {
        v = *reinterpret_cast<const F4*>(fr[nb] + kc);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v.v[e] = fr[nb][kc + e];
      }
      const bool use = in && fok[nb];
#pragma unroll
      for (int e = 0; e < 4; ++e) o.a[nb].v[e] = use ? v.v[e] : T(0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#ifdef SQFA_PROJ_PLAIN_LOADS
      o.b[e] = *reinterpret_cast<const Vec*>(pc + (size_t)(kc + e) * D);
#else
      o.b[e] = __builtin_nontemporal_load(reinterpret_cast<const Vec*>(pc + (size_t)(kc + e) * D));
#endif
    }
  };

  Acc acc[NB][VW];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int j = 0; j < VW; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) acc[nb][j][reg] = T(0);

  auto multiply = [&](const Block& o) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[nb][j] = Tr::mfma(o.a[nb].v[e], o.b[e][j], acc[nb][j]);
  };

  constexpr int DEPTH = SQFA_PROJ_DEPTH;
  Block ring[DEPTH + 1];
#pragma unroll
  for (int p = 0; p < DEPTH; ++p) load_block(p < nblk ? p : nblk - 1, ring[p]);   // (p >= nblk: loaded, never multiplied)
  int blk = 0;
  // ring slots rotate with compile-time indices: the body is unrolled over one full turn of the ring
  for (; blk + DEPTH + 1 <= nblk; blk += DEPTH + 1) {
#pragma unroll
    for (int u = 0; u <= DEPTH; ++u) {
      const int nxt = blk + u + DEPTH;
      load_block(nxt < nblk ? nxt : nblk - 1, ring[(u + DEPTH) % (DEPTH + 1)]);
      multiply(ring[u]);
    }
  }
  // remaining blocks (fewer than one turn): their data is already in the ring, slots 0 .. rem-1
  {
    const int rem = nblk - blk;
#pragma unroll
    for (int u = 0; u < DEPTH + 1; ++u) {
      if (u < rem) {
        if (u + DEPTH < rem) load_block(blk + u + DEPTH, ring[(u + DEPTH) % (DEPTH + 1)]);
        multiply(ring[u]);
      }
    }
  }
  if (!has_cols) return;
  // acc[nb][j][reg] = T^T[n = 16 nb + acc_row(q, reg)][d = d0 + j]
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

// Stripe geometry: a row holds D / VW lane-vectors; with at most 16 lanes per 16-lane group it needs at least
// ceil(D / (16 VW)) stripes.  A few more stripes are accepted when they divide the row exactly (no ragged last
// stripe: D=784, float32: 14 x 14 lanes instead of 12 x 16 + 4), at the price of idle MFMA columns.
static void stripe_geometry(int D, int VW, int* nstripes, int* LS) {
  const int vecs = D / VW, nmin = (vecs + 15) / 16;
  int n = nmin;
#ifndef SQFA_PROJ_EXACT_STRIPES
#define SQFA_PROJ_EXACT_STRIPES 1
#endif
  if (SQFA_PROJ_EXACT_STRIPES && vecs % nmin != 0) {
    for (int t = nmin + 1; t <= nmin + (nmin + 3) / 4; ++t) {
      if (vecs % t == 0) { n = t; break; }
    }
  }
  *nstripes = n;
  *LS = (vecs + n - 1) / n;
}

template <typename T, int WV>
static void launch_project_w(const T* f, const T* p, T* t, int C, int D, int K, hipStream_t stream) {
  int nstripes, LS;
  stripe_geometry(D, ProjTraits<T>::VW, &nstripes, &LS);
  const dim3 grid((nstripes + WV - 1) / WV, C, 1), block(64 * WV);
  const int nb = (K + 15) / 16;
  const bool fv = (reinterpret_cast<size_t>(f) % (4 * sizeof(T))) == 0;   // D % 4 == 0: every row of F is aligned then
#define SQFA_PROJ_LAUNCH(NB_)                                                                                                   \
  if (fv) hipLaunchKernelGGL((project_kernel<T, NB_, WV, true>), grid, block, 0, stream, f, p, t, C, D, K, nstripes, LS);    \
  else hipLaunchKernelGGL((project_kernel<T, NB_, WV, false>), grid, block, 0, stream, f, p, t, C, D, K, nstripes, LS);
  switch (nb) {
    case 1: SQFA_PROJ_LAUNCH(1) break;
    case 2: SQFA_PROJ_LAUNCH(2) break;
    case 3: SQFA_PROJ_LAUNCH(3) break;
    default: SQFA_PROJ_LAUNCH(4) break;
  }
#undef SQFA_PROJ_LAUNCH
}

// Workgroup width: the waves of a workgroup read adjacent stripes, i.e. up to WV*256 contiguous bytes of the same
// rows.  Measured on MI355X in rounds 1-2 (tools/time_variants_proj.py): throughput grows with the contiguous run
// until a class needs several workgroups anyway, where narrower ones balance better.
template <typename T>
static void launch_project(const T* f, const T* p, T* t, int C, int D, int K, hipStream_t stream) {
  int nstripes, LS;
  stripe_geometry(D, ProjTraits<T>::VW, &nstripes, &LS);
#ifndef SQFA_PROJ_WV_SMALL
#define SQFA_PROJ_WV_SMALL 16
#endif
  // (K > 48: four accumulator blocks need more than the 128 registers a 1024-thread workgroup leaves per lane)
  if (nstripes <= 16 && K <= 48) launch_project_w<T, SQFA_PROJ_WV_SMALL>(f, p, t, C, D, K, stream);
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
