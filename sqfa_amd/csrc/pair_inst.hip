// pair_inst.hip -- one explicit instantiation of the pair tile kernel per translation unit.
// Compiled once per configuration with -DSQFA_T=.. -DSQFA_TAG=.. -DSQFA_MR=.. -DSQFA_G=.. -DSQFA_CPL=.. -DSQFA_TJ=.. -DSQFA_WAVES=..
// (whole-column layout, pair_kernel.hpp) and, with -DSQFA_RS=16|32 in addition, for the 2-D layout of pair_kernel_2d.hpp
// (SQFA_G is then the number of COLUMN lanes).
#include <type_traits>

#include "configs.hpp"
#include "pair_kernel.hpp"

#define SQFA_CAT_(a, b, c) a##b##_##c
#define SQFA_CAT(a, b, c) SQFA_CAT_(a, b, c)

#if defined(SQFA_RS) && SQFA_RS != 0
#include "pair_kernel_2d.hpp"
#define SQFA_ROW2D_MATCHES(T, MR, GC, CPL, TJ, WV, RS) \
  || (std::is_same<T, SQFA_T>::value && MR == SQFA_MR && GC == SQFA_G && CPL == SQFA_CPL && TJ == SQFA_TJ && WV == SQFA_WAVES && RS == SQFA_RS)
static_assert(false SQFA_CONFIGS2D_F32(SQFA_ROW2D_MATCHES) SQFA_CONFIGS2D_F64(SQFA_ROW2D_MATCHES),
              "Makefile CONFIGS2D and configs.hpp disagree");
namespace sqfa {
hipError_t SQFA_CAT(launch_pair2d_, SQFA_TAG, SQFA_MR)(const PairParams& p, hipStream_t stream) {
  return launch_pair_tiles_2d<PairCfg2D<SQFA_T, SQFA_MR, SQFA_G, SQFA_CPL, SQFA_TJ, SQFA_WAVES, SQFA_RS>>(p, stream);
}
hipError_t SQFA_CAT(launch_factor2d_, SQFA_TAG, SQFA_MR)(const PairParams& p, hipStream_t stream) {
  return launch_class_factors<PairCfg2D<SQFA_T, SQFA_MR, SQFA_G, SQFA_CPL, SQFA_TJ, SQFA_WAVES, SQFA_RS>>(p, stream);
}
hipError_t SQFA_CAT(launch_classeig2d_, SQFA_TAG, SQFA_MR)(const void* LT, int n, int m, double* U, double* lam, hipStream_t stream) {
  return launch_class_eig<PairCfg2D<SQFA_T, SQFA_MR, SQFA_G, SQFA_CPL, SQFA_TJ, SQFA_WAVES, SQFA_RS>>(LT, n, m, U, lam, stream);
}
}  // namespace sqfa
#else
// the -D geometry of this translation unit must be a row of the table the API dispatches on
#define SQFA_ROW_MATCHES(T, MR, G, CPL, TJ, WV) \
  || (std::is_same<T, SQFA_T>::value && MR == SQFA_MR && G == SQFA_G && CPL == SQFA_CPL && TJ == SQFA_TJ && WV == SQFA_WAVES)
static_assert(false SQFA_CONFIGS_F32(SQFA_ROW_MATCHES) SQFA_CONFIGS_F64(SQFA_ROW_MATCHES) SQFA_CONFIGS_F32_SMALL(SQFA_ROW_MATCHES)
              SQFA_CONFIGS_F64_SMALL(SQFA_ROW_MATCHES),
              "Makefile CONFIGS / CONFIGS_SMALL and configs.hpp disagree");
namespace sqfa {
hipError_t SQFA_CAT(launch_pair_, SQFA_TAG, SQFA_MR)(const PairParams& p, hipStream_t stream) {
  return launch_pair_tiles<PairCfg<SQFA_T, SQFA_MR, SQFA_G, SQFA_CPL, SQFA_TJ, SQFA_WAVES>>(p, stream);
}
hipError_t SQFA_CAT(launch_factor_, SQFA_TAG, SQFA_MR)(const PairParams& p, hipStream_t stream) {
  return launch_class_factors<PairCfg<SQFA_T, SQFA_MR, SQFA_G, SQFA_CPL, SQFA_TJ, SQFA_WAVES>>(p, stream);
}
// per-class eigen-decomposition (spd_log / spd_sqrt): depends on (T, MR) only, not on the pair geometry of this row
hipError_t SQFA_CAT(launch_classeig_, SQFA_TAG, SQFA_MR)(const void* LT, int n, int m, double* U, double* lam, hipStream_t stream) {
  return launch_class_eig<PairCfg<SQFA_T, SQFA_MR, SQFA_G, SQFA_CPL, SQFA_TJ, SQFA_WAVES>>(LT, n, m, U, lam, stream);
}
}  // namespace sqfa
#endif
