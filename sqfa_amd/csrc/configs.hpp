// configs.hpp -- the (dtype, padded size) -> lane-group geometry table.
// X(T, MR, G, CPL, TJ, WAVES): matrices up to MR x MR are handled by G lanes per pair with
// CPL column slots per lane; a workgroup of WAVES wave64 owns a tile of (64/G) x TJ classes.
// A problem of size m runs on the smallest MR >= m (identity padding is exact: padded
// generalized eigenvalues are 1).  Keep in sync with the CONFIGS list in the Makefile.
#pragma once

#define SQFA_CONFIGS_F32(X)  \
  X(float, 4, 1, 4, 8, 4)    \
  X(float, 8, 1, 8, 8, 4)    \
  X(float, 12, 4, 3, 8, 4)   \
  X(float, 16, 4, 4, 8, 4)  \
  X(float, 17, 4, 5, 8, 4)  \
  X(float, 20, 4, 5, 8, 4)  \
  X(float, 24, 8, 3, 8, 4)   \
  X(float, 32, 8, 4, 4, 1)   \
  X(float, 33, 8, 5, 4, 1)   \
  X(float, 40, 8, 5, 4, 1)   \
  X(float, 48, 16, 3, 4, 1)  \
  X(float, 64, 32, 2, 4, 2)

#define SQFA_CONFIGS_F64(X)  \
  X(double, 4, 1, 4, 8, 4)   \
  X(double, 8, 2, 4, 8, 4)   \
  X(double, 12, 4, 3, 8, 4)  \
  X(double, 16, 8, 2, 8, 4)  \
  X(double, 17, 8, 3, 8, 4)  \
  X(double, 20, 8, 3, 8, 4)  \
  X(double, 24, 16, 2, 4, 1) \
  X(double, 32, 16, 2, 4, 1) \
  X(double, 33, 16, 3, 4, 1) \
  X(double, 48, 32, 2, 4, 1) \
  X(double, 64, 64, 1, 4, 2)
