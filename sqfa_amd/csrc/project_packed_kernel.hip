// project_packed_kernel.hip -- T_c = Psi_c F^T from BLOCK-TRIANGULAR PACKED statistics: the symmetric scatter matrices
// are stored (once per fit) as their lower block triangle and every closure streams 51-54 % of the bytes of the full
// (C,D,D) tensor (the reference's conjugate_matrix, src/sqfa/linalg.py:19-45 via transform_scatters,
// src/sqfa/model.py:172-188, reads the full tensor -- twice, counting its autograd backward).
//
// Packed layout (sqfa_pack_scatters), per class: row blocks rb = 0 .. D/16-1 of 16 rows; row block rb (rows R = 16 rb ..)
// holds, one after the other, the 16 x 64 tiles of the column stripes s = 0 .. R/64 (stripe s = columns 64 s .. 64 s + 63,
// zero padded past D), i.e. everything of those rows up to and including the 64-wide diagonal block; tile = 16 rows x 64
// floats, row-major (4 KiB).  A workgroup that walks down the rows reads ONE contiguous run of (R/64 + 1) x 4 KiB per step.
//   blocks before row block rb:  (g + 1)(2 g + rem),  g = rb / 4, rem = rb % 4      [1 block = 1024 floats]
//
// One workgroup per class, WAVES waves; wave w owns the stripes w, w + WAVES, ...  All waves walk the row blocks in
// lockstep (one barrier per step).  Tile (rb, s) feeds two exact-f32 MFMA products (v_mfma_f32_16x16x4_f32):
//   column output  T^T[n][d in stripe s] += sum_{r in rb} F[n][r] Psi[r][d]     accumulators live in registers for the whole
//                  walk (the current kernel's product); tile read as "map A": lane (q, i16) holds row 4i + q, cols 4 i16 + j
//   row output     T^T[n][r in rb] += sum_{d in stripe s} F[n][d] Psi[r][d]     only for tiles strictly left of the
//                  diagonal block (64 (s+1) <= R): the transposed half the full tensor would have supplied; the MFMA
//                  contracts over the lane >> 4 index, so the tile is needed a second time as "map B": lane (q, i16) holds
//                  row i16, cols 16 jj + 4 q + j -- transposed through a per-wave LDS scratch (a second global read of the
//                  same bytes cost 0.05 ms more at c3); F for the stripe's 64 columns is re-read per tile (L2 hits).  The 16 x 16 partial results of the contributing waves (w < R/64) are
//                  summed in LDS in wave order -- fixed association, bitwise reproducible -- by the wave that owns the
//                  stripe R/64 and added to ITS column-output accumulators (rows R.. are columns R.. of T^T).
// Every T element is stored exactly once, after the walk.  Algorithmic bytes per launch: 4 * C * packed_elems(D).
// The MFMA work is that of the full kernel (2 C K D^2 flop: symmetry saves bytes, not products): at K = 16 the exact-f32
// MFMA floor is 0.125 ms at c3 against 0.17 ms of HBM time for the packed bytes; at K = 32 (c4) the kernel is MFMA-bound.
#include <hip/hip_runtime.h>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/sqfa_hip.h"
#include "proj_traits.hpp"

bool sqfa_profile_enabled();                                            // sqfa_api.hip
std::vector<std::pair<hipEvent_t, hipEvent_t>>& sqfa_project_events();  // sqfa_api.hip
std::mutex& sqfa_project_events_mutex();                                 // sqfa_api.hip

namespace sqfa {

__host__ __device__ inline size_t packed_blocks_before(int rb) {
  const size_t g = (size_t)(rb >> 2), rem = (size_t)(rb & 3);
  return (g + 1) * (2 * g + rem);
}
__host__ __device__ inline size_t packed_class_elems(int D) { return packed_blocks_before(D / 16) * 1024; }

// Psi (C,D,D) row-major -> packed (C, packed_class_elems(D)); one thread per float4 of the output
__global__ __launch_bounds__(256) void pack_scatters_kernel(const float* __restrict__ Psi, float* __restrict__ out, int D,
                                                            size_t class_elems) {
  const int c = blockIdx.y;
  const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;      // float4 index inside the class
  if (v * 4 >= class_elems) return;
  const size_t blk = v >> 8;                                     // 1024 floats = 256 float4 per tile
  const int within = (int)(v & 255), row = within >> 4, col4 = (within & 15) * 4;
  // invert (g + 1)(2 g + rem) <= blk: g from the group boundary 2 g (g + 1) <= blk
  int g = (int)((__builtin_sqrt((double)(1 + 2 * blk)) - 1.0) * 0.5);
  while ((size_t)2 * (g + 1) * (g + 2) <= blk) ++g;
  while ((size_t)2 * g * (g + 1) > blk) --g;
  const int rest = (int)(blk - (size_t)2 * g * (g + 1));         // tiles before blk inside group g: rem (g + 1) + s
  const int rem = rest / (g + 1), s = rest % (g + 1);
  const int r = 16 * (4 * g + rem) + row, d = 64 * s + col4;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (d < D) o = *reinterpret_cast<const float4*>(Psi + ((size_t)c * D + r) * D + d);   // D % 16 == 0: whole float4 inside
  *reinterpret_cast<float4*>(out + (size_t)c * class_elems + v * 4) = o;
}

// (SQFA_PK_NO_ROWOUT / _NO_BARRIER / _NO_REDUCE / _FAKE_F / _FAKE_TR below are development switches for timing ablations --
// profiles/r4_projection_packed.txt; results are wrong with any of them)
template <int NB, int WAVES, int SPW, int MINW>
__global__ __launch_bounds__(64 * WAVES, MINW) void project_packed_kernel(const float* __restrict__ F, const float* __restrict__ Pk,
                                                                    float* __restrict__ Tout, int D, int K, size_t class_elems) {
  using Tr = ProjTraits<float>;
  using Acc = Tr::Acc;
  constexpr int NT = 64 * WAVES;
  __shared__ float s_f[2][16][16 * NB];            // F^T rows of the current / next row block: [buf][k][n]
  __shared__ float s_red[2][WAVES][NB][256];       // row-output partials per wave: [buf][wave][nb][r][n]
  __shared__ float s_tot[NB][256];                 // their sum (owner wave only)
  __shared__ float s_tr[WAVES][16 * 68];          // per-wave tile transposition (map A -> map B), row pitch 68 dwords: both the
                                                   // 16-byte writes (row 4 i + q, cols 4 i16) and reads (row i16, cols 16 jj + 4 q) are conflict-free
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int i16 = lane & 15, q = lane >> 4;
  const int c = blockIdx.x;
  const int NS = (D + 63) >> 6, NRB = D >> 4;
  const float* __restrict__ pk = Pk + (size_t)c * class_elems;

  auto stage = [&](int rb_, int buf) {             // F^T rows 16 rb_ .. + 15 (k fastest over the threads: 64-byte runs of F)
    for (int e = tid; e < 16 * 16 * NB; e += NT) {
      const int kk = e & 15, n = e >> 4;
      s_f[buf][kk][n] = (n < K) ? F[(size_t)n * D + 16 * rb_ + kk] : 0.f;
    }
  };

  // F for a stripe's columns (row-output A operand), F[16 nb + i16][64 s + 16 jj + 4 q .. + 3]: 16-byte loads repeated per tile
  // (L2 hits; keeping them resident costs SPW * NB * 16 registers, which is what decides how many classes a CU holds)
  auto load_f = [&](int s, int nb, int jj) {
    const int n = 16 * nb + i16, d = 64 * s + 16 * jj + 4 * q;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < K && d < D) v = *reinterpret_cast<const f32x4*>(F + (size_t)n * D + d);
    return v;
  };
  Acc colacc[SPW][NB][4];
#pragma unroll
  for (int u = 0; u < SPW; ++u)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) colacc[u][nb][jj][reg] = 0.f;

  // Tiles are processed as a software pipeline over the items (row block, my stripe slot u): the HBM read of item i + 1
  // ("map A", 4 KiB) is issued before the MFMAs of item i, into the other of two register sets (the row-block loop is
  // unrolled by two so that the set of an item is a compile-time constant); the second read of the same bytes ("map B",
  // L1 / L2 hits) and the F loads are issued at the start of the item itself and land behind its 16 column-output MFMAs.
  f32x4 ta[2][4];
  auto issue = [&](auto set_c, int rb, int u) {
    constexpr int set = decltype(set_c)::value;
    const int s = wave + u * WAVES, sR = rb >> 2;
    if (rb < NRB && s <= sR) {
      const float* __restrict__ tile = pk + (packed_blocks_before(rb) + (size_t)s) * 1024;
#pragma unroll
      for (int i = 0; i < 4; ++i) ta[set][i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(tile + (4 * i + q) * 64 + 4 * i16));
    }
  };
  Acc rowacc[NB];
  // One item.  Loads complete in issue order (s_waitcnt vmcnt counts them in order), so the loads an item needs SOON -- the
  // second read of its own tile and its F values, L1 / L2 hits -- are issued BEFORE the HBM prefetch of the next item: waiting
  // for them then leaves the four prefetch loads outstanding (issued after them, the prefetch would have to land first and
  // every tile would pay the whole HBM latency: 0.52 ms instead of 0.3x at c3).
  auto compute = [&](auto set_c, auto u_c, int rb, int next_rb, int next_u) {
    constexpr int set = decltype(set_c)::value, u = decltype(u_c)::value;
    const int s = wave + u * WAVES, sR = rb >> 2, rbuf = rb & 1;
    const bool active = s <= sR, left = s < sR;      // left: strictly left of the diagonal block, the transposed half as well
    f32x4 tb[4], fv[NB][4];
#ifndef SQFA_PK_NO_ROWOUT
    if (left) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
#ifdef SQFA_PK_FAKE_F
        for (int jj = 0; jj < 4; ++jj) fv[nb][jj] = f32x4{1.f, 2.f, 3.f, (float)jj};
#else
        for (int jj = 0; jj < 4; ++jj) fv[nb][jj] = load_f(s, nb, jj);
#endif
    }
#endif
    __builtin_amdgcn_sched_barrier(0);               // keep the F loads ahead of the prefetch in program order
    issue(std::integral_constant<int, set ^ 1>{}, next_rb, next_u);
    __builtin_amdgcn_sched_barrier(0);
#ifndef SQFA_PK_NO_ROWOUT
    if (left) {
      // the tile in the second lane map (row i16, cols 16 jj + 4 q ..) through this wave's LDS scratch: no second global read
      float* tr = s_tr[wave];
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(tr + (4 * i + q) * 68 + 4 * i16) = ta[set][i];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) tb[jj] = *reinterpret_cast<const f32x4*>(tr + i16 * 68 + 16 * jj + 4 * q);
#ifdef SQFA_PK_FAKE_TR
      for (int jj = 0; jj < 4; ++jj) tb[jj] = ta[set][jj];
#endif
    }
#endif
    if (active) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float fa = s_f[rbuf][4 * i + q][nb * 16 + i16];
#pragma unroll
          for (int j = 0; j < 4; ++j) colacc[u][nb][j] = Tr::mfma(fa, ta[set][i][j], colacc[u][nb][j]);
        }
      }
#ifndef SQFA_PK_NO_ROWOUT
      if (left) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int j = 0; j < 4; ++j) rowacc[nb] = Tr::mfma(fv[nb][jj][j], tb[jj][j], rowacc[nb]);
      }
#endif
    }
  };
  auto finish_block = [&](int rb) {
    // rowacc[nb][reg] = D'[n = 4 q + reg][r = i16]: stored [r][n] (one 16-byte write per lane, conflict-free)
    const int sR = rb >> 2, t = rb & 3, rbuf = rb & 1;
    const int contributors = sR < WAVES ? sR : WAVES;              // waves whose first stripe lies left of the diagonal block
    if (wave < contributors) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) *reinterpret_cast<f32x4*>(&s_red[rbuf][wave][nb][i16 * 16 + 4 * q]) = rowacc[nb];
    }
#ifndef SQFA_PK_NO_BARRIER      // development: timing ablations (results are wrong with any of them)
    __syncthreads();
#endif
#ifdef SQFA_PK_NO_REDUCE
    if (false) {
#else
    if (contributors > 0 && wave == sR % WAVES) {
#endif
      // the owner of stripe sR: sum in wave order, then add into its column-output accumulators
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        f32x4 tot = *reinterpret_cast<const f32x4*>(&s_red[rbuf][0][nb][4 * lane]);
        for (int w = 1; w < contributors; ++w) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&s_red[rbuf][w][nb][4 * lane]);
          tot += v;
        }
        *reinterpret_cast<f32x4*>(&s_tot[nb][4 * lane]) = tot;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int uo = sR / WAVES;                                   // wave-uniform slot of stripe sR in this wave
      if ((i16 >> 2) == t) {
        // my accumulator columns d = 64 sR + 4 i16 + j are the rows r = 4 (i16 - 4 t) + j of this row block
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&s_tot[nb][(4 * (i16 & 3) + j) * 16 + 4 * q]);
#pragma unroll
            for (int u = 0; u < SPW; ++u) {
              if (u == uo) colacc[u][nb][j] += v;
            }
          }
        }
      }
    }
  };

  stage(0, 0);
  issue(std::integral_constant<int, 0>{}, 0, 0);
  __syncthreads();
  auto half = [&](auto par_c, int rb) {             // one row block; par = rb & 1 at compile time
    constexpr int par = decltype(par_c)::value;
    if (rb >= NRB) return;                           // wave-uniform (odd number of row blocks)
    if (rb + 1 < NRB) stage(rb + 1, par ^ 1);        // next row block's F^T rows (visible after this block's barrier)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) rowacc[nb][reg] = 0.f;
    auto items = [&](auto self, auto u_c) {
      constexpr int U = decltype(u_c)::value, item = par * SPW + U, set = item & 1;
      // next item: (rb, U + 1), or the first slot of the next row block
      if constexpr (U + 1 < SPW) compute(std::integral_constant<int, set>{}, u_c, rb, rb, U + 1);
      else compute(std::integral_constant<int, set>{}, u_c, rb, rb + 1, 0);
      if constexpr (U + 1 < SPW) self(self, std::integral_constant<int, U + 1>{});
    };
    items(items, std::integral_constant<int, 0>{});
    finish_block(rb);
  };
  for (int rb = 0; rb < NRB; rb += 2) {
    half(std::integral_constant<int, 0>{}, rb);
    half(std::integral_constant<int, 1>{}, rb + 1);
  }
  // colacc[u][nb][j][reg] = T^T[n = 16 nb + 4 q + reg][d = 64 s + 4 i16 + j]
#pragma unroll
  for (int u = 0; u < SPW; ++u) {
    const int s = wave + u * WAVES;
    if (s >= NS) continue;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d = 64 * s + 4 * i16 + j, n0 = nb * 16 + 4 * q;
        if (d < D && n0 < K) {
          float* out = Tout + ((size_t)c * D + d) * K + n0;
          if ((K & 3) == 0) {
            *reinterpret_cast<f32x4*>(out) = colacc[u][nb][j];
          } else {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
              if (n0 + reg < K) out[reg] = colacc[u][nb][j][reg];
          }
        }
      }
    }
  }
}

template <int NB>
static bool launch_packed(const float* f, const float* pk, float* t, int C, int D, int K, hipStream_t stream) {
  const int NS = (D + 63) / 64;
  const size_t ce = packed_class_elems(D);
  const dim3 grid(C, 1, 1);
#define SQFA_PK(WV, SPW, MINW) hipLaunchKernelGGL((project_packed_kernel<NB, WV, SPW, MINW>), grid, dim3(64 * WV), 0, stream, f, pk, t, D, K, ce)
  // Workgroup shapes: 8 waves with a few stripes each, compiled for 4 waves per SIMD where the accumulators allow it (two
  // classes resident per CU fill the triangular activity profile of one class's walk), for 2 otherwise
  if constexpr (NB == 1) {
    if (NS <= 4) SQFA_PK(4, 1, 4);
    else if (NS <= 8) SQFA_PK(8, 1, 4);
#ifndef SQFA_PK_CFG16
#define SQFA_PK_CFG16 0
#endif
    else if (NS <= 16) { if (SQFA_PK_CFG16 == 0) SQFA_PK(8, 2, 4); else if (SQFA_PK_CFG16 == 1) SQFA_PK(16, 1, 4); else SQFA_PK(4, 4, 2); }
    else if (NS <= 32) SQFA_PK(8, 4, 2);
    else if (NS <= 48) SQFA_PK(8, 6, 2);
    else if (NS <= 64) SQFA_PK(8, 8, 1);
    else return false;
  } else if constexpr (NB == 2) {
    if (NS <= 4) SQFA_PK(4, 1, 4);
    else if (NS <= 8) SQFA_PK(8, 1, 4);
    else if (NS <= 16) SQFA_PK(8, 2, 2);
    else if (NS <= 32) SQFA_PK(8, 4, 2);
    else return false;                 // the accumulators of more stripes per wave do not fit: keep the full tensor
  } else {
    if (NS <= 8) SQFA_PK(8, 1, 2);
    else if (NS <= 16) SQFA_PK(8, 2, 2);
    else if (NS <= 32) SQFA_PK(8, 4, 1);
    else return false;
  }
#undef SQFA_PK
  return true;
}

}  // namespace sqfa

extern "C" size_t sqfa_packed_scatter_elems(int D) {
  if (D < 16 || (D % 16) != 0) return 0;
  return sqfa::packed_class_elems(D);
}

extern "C" int sqfa_pack_scatters(const void* Psi, int C, int D, int dtype, void* packed_out, void* stream_) {
  using namespace sqfa;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (Psi == nullptr || packed_out == nullptr || C < 1 || D < 1) return SQFA_ERR_BAD_ARGUMENT;
  if (dtype != SQFA_F32 || D < 16 || (D % 16) != 0 || D > 4096) return SQFA_ERR_UNSUPPORTED_M;
  const size_t ce = packed_class_elems(D);
  const dim3 grid((unsigned)((ce / 4 + 255) / 256), C, 1);
  hipLaunchKernelGGL(pack_scatters_kernel, grid, dim3(256), 0, stream, static_cast<const float*>(Psi), static_cast<float*>(packed_out), D, ce);
  return hipGetLastError() == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}

extern "C" int sqfa_project_scatters_packed(const void* F, int K, int D, const void* packed, int C, int dtype, void* T_out,
                                            void* stream_) {
  using namespace sqfa;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (F == nullptr || packed == nullptr || T_out == nullptr || K < 1 || D < 1 || C < 1) return SQFA_ERR_BAD_ARGUMENT;
  if (dtype != SQFA_F32 || D < 16 || (D % 16) != 0 || D > 4096 || K > 64 || K > D) return SQFA_ERR_UNSUPPORTED_M;
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
  const float* f = static_cast<const float*>(F);
  const float* pk = static_cast<const float*>(packed);
  float* t = static_cast<float*>(T_out);
  bool ok = false;
  switch ((K + 15) / 16) {
    case 1: ok = launch_packed<1>(f, pk, t, C, D, K, stream); break;
    case 2: ok = launch_packed<2>(f, pk, t, C, D, K, stream); break;
    case 3: ok = launch_packed<3>(f, pk, t, C, D, K, stream); break;
    default: ok = launch_packed<4>(f, pk, t, C, D, K, stream); break;
  }
  if (prof) {
    (void)hipEventRecord(e1, stream);
    std::lock_guard<std::mutex> lock(sqfa_project_events_mutex());
    sqfa_project_events().emplace_back(e0, e1);
  }
  if (!ok) return SQFA_ERR_UNSUPPORTED_M;
  return hipGetLastError() == hipSuccess ? SQFA_OK : SQFA_ERR_LAUNCH;
}
