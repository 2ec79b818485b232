// configs.hpp -- the (dtype, padded size) -> lane-group geometry table.
// X(T, MR, G, CPL, TJ, WAVES): matrices up to MR x MR are handled by G lanes per pair with
// CPL column slots per lane; a workgroup of WAVES wave64 owns a tile of (64/G) x TJ classes.
// A problem of size m runs on the smallest MR >= m (identity padding is exact: padded
// generalized eigenvalues are 1).  Keep in sync with the CONFIGS list in the Makefile.
// Every row is its own macro so that a development build can override one geometry from the command line
// (tools/build_variant.sh: -D'SQFA_ROW_F32_32(X)=X(float,32,16,2,4,1)').
#pragma once

#ifndef SQFA_ROW_F32_4
#define SQFA_ROW_F32_4(X) X(float, 4, 1, 4, 8, 4)
#endif
#ifndef SQFA_ROW_F32_8
#define SQFA_ROW_F32_8(X) X(float, 8, 1, 8, 8, 4)
#endif
#ifndef SQFA_ROW_F32_12
#define SQFA_ROW_F32_12(X) X(float, 12, 4, 3, 8, 4)
#endif
#ifndef SQFA_ROW_F32_16
#define SQFA_ROW_F32_16(X) X(float, 16, 4, 4, 8, 4)
#endif
#ifndef SQFA_ROW_F32_17
#define SQFA_ROW_F32_17(X) X(float, 17, 4, 5, 8, 4)
#endif
#ifndef SQFA_ROW_F32_20
#define SQFA_ROW_F32_20(X) X(float, 20, 4, 5, 8, 4)
#endif
#ifndef SQFA_ROW_F32_24
#define SQFA_ROW_F32_24(X) X(float, 24, 8, 3, 8, 4)
#endif
#ifndef SQFA_ROW_F32_32
#define SQFA_ROW_F32_32(X) X(float, 32, 8, 4, 4, 1)
#endif
#ifndef SQFA_ROW_F32_33
#define SQFA_ROW_F32_33(X) X(float, 33, 8, 5, 4, 1)
#endif
#ifndef SQFA_ROW_F32_40
#define SQFA_ROW_F32_40(X) X(float, 40, 8, 5, 4, 1)
#endif
#ifndef SQFA_ROW_F32_48
#define SQFA_ROW_F32_48(X) X(float, 48, 16, 3, 4, 1)
#endif
#ifndef SQFA_ROW_F32_64
#define SQFA_ROW_F32_64(X) X(float, 64, 32, 2, 4, 2)
#endif

#define SQFA_CONFIGS_F32(X) \
  SQFA_ROW_F32_4(X) \
  SQFA_ROW_F32_8(X) \
  SQFA_ROW_F32_12(X) \
  SQFA_ROW_F32_16(X) \
  SQFA_ROW_F32_17(X) \
  SQFA_ROW_F32_20(X) \
  SQFA_ROW_F32_24(X) \
  SQFA_ROW_F32_32(X) \
  SQFA_ROW_F32_33(X) \
  SQFA_ROW_F32_40(X) \
  SQFA_ROW_F32_48(X) \
  SQFA_ROW_F32_64(X)

#ifndef SQFA_ROW_F64_4
#define SQFA_ROW_F64_4(X) X(double, 4, 1, 4, 8, 4)
#endif
#ifndef SQFA_ROW_F64_8
#define SQFA_ROW_F64_8(X) X(double, 8, 2, 4, 8, 4)
#endif
#ifndef SQFA_ROW_F64_12
#define SQFA_ROW_F64_12(X) X(double, 12, 4, 3, 8, 4)
#endif
#ifndef SQFA_ROW_F64_16
#define SQFA_ROW_F64_16(X) X(double, 16, 8, 2, 8, 4)
#endif
#ifndef SQFA_ROW_F64_17
#define SQFA_ROW_F64_17(X) X(double, 17, 8, 3, 8, 4)
#endif
#ifndef SQFA_ROW_F64_20
#define SQFA_ROW_F64_20(X) X(double, 20, 8, 3, 8, 4)
#endif
#ifndef SQFA_ROW_F64_24
#define SQFA_ROW_F64_24(X) X(double, 24, 16, 2, 4, 1)
#endif
#ifndef SQFA_ROW_F64_32
#define SQFA_ROW_F64_32(X) X(double, 32, 16, 2, 4, 1)
#endif
#ifndef SQFA_ROW_F64_33
#define SQFA_ROW_F64_33(X) X(double, 33, 16, 3, 4, 1)
#endif
#ifndef SQFA_ROW_F64_48
#define SQFA_ROW_F64_48(X) X(double, 48, 32, 2, 4, 1)
#endif
#ifndef SQFA_ROW_F64_64
#define SQFA_ROW_F64_64(X) X(double, 64, 64, 1, 4, 2)
#endif

#define SQFA_CONFIGS_F64(X) \
  SQFA_ROW_F64_4(X) \
  SQFA_ROW_F64_8(X) \
  SQFA_ROW_F64_12(X) \
  SQFA_ROW_F64_16(X) \
  SQFA_ROW_F64_17(X) \
  SQFA_ROW_F64_20(X) \
  SQFA_ROW_F64_24(X) \
  SQFA_ROW_F64_32(X) \
  SQFA_ROW_F64_33(X) \
  SQFA_ROW_F64_48(X) \
  SQFA_ROW_F64_64(X)
