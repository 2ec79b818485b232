// pair_inst.hip -- one explicit instantiation of the pair tile kernel per translation unit.
// Compiled once per configuration with -DSQFA_T=.. -DSQFA_TAG=.. -DSQFA_MR=.. -DSQFA_G=.. -DSQFA_CPL=.. -DSQFA_TJ=.. -DSQFA_WAVES=..
#include <type_traits>

#include "configs.hpp"
#include "pair_kernel.hpp"

// the -D geometry of this translation unit must be a row of the table the API dispatches on
#define SQFA_ROW_MATCHES(T, MR, G, CPL, TJ, WV) \
  || (std::is_same<T, SQFA_T>::value && MR == SQFA_MR && G == SQFA_G && CPL == SQFA_CPL && TJ == SQFA_TJ && WV == SQFA_WAVES)
static_assert(false SQFA_CONFIGS_F32(SQFA_ROW_MATCHES) SQFA_CONFIGS_F64(SQFA_ROW_MATCHES),
              "Makefile CONFIGS and configs.hpp disagree");

#define SQFA_CAT_(a, b, c) a##b##_##c
#define SQFA_CAT(a, b, c) SQFA_CAT_(a, b, c)

namespace sqfa {
hipError_t SQFA_CAT(launch_pair_, SQFA_TAG, SQFA_MR)(const PairParams& p, hipStream_t stream) {
  return launch_pair_tiles<PairCfg<SQFA_T, SQFA_MR, SQFA_G, SQFA_CPL, SQFA_TJ, SQFA_WAVES>>(p, stream);
}
}  // namespace sqfa
