// rbf_device.h -- GPy RBF scaled squared distance, restated op by op (no FMA contraction) so that K_uu / K_uf
// round like numpy's   r2 = -2 x.z + (|x|^2 + |z|^2);  r = sqrt(clip(r2, 0, inf)) / l;  K = s2 * exp(-0.5 * r**2)
// (GPy 1.9.5 kern/stationary, reached from hetmogp/util.py:161,197).
#pragma once
#include "common.h"

template <int P>
__device__ __forceinline__ double rbf_r2(const double* x, double xsq, const double* z, double zsq, double ell) {
#pragma clang fp contract(off)
  double dot = x[0] * z[0];
#pragma unroll
  for (int p = 1; p < P; ++p) dot = dot + x[p] * z[p];
  double r2 = -2.0 * dot + (xsq + zsq);
  r2 = fmax(r2, 0.0);
  const double r = sqrt(r2) / ell;
  return r * r;
}
// cheaper scaled squared distance: clip(r2, 0) * (1/l^2), no sqrt / divide
template <int P>
__device__ __forceinline__ double rbf_r2_fast(const double* x, double xsq, const double* z, double zsq, double inv_l2) {
#pragma clang fp contract(off)
  double dot = x[0] * z[0];
#pragma unroll
  for (int p = 1; p < P; ++p) dot = dot + x[p] * z[p];
  const double r2 = -2.0 * dot + (xsq + zsq);
  return fmax(r2, 0.0) * inv_l2;
}
template <int P>
__device__ __forceinline__ double sumsq(const double* x) {
#pragma clang fp contract(off)
  double s = x[0] * x[0];
#pragma unroll
  for (int p = 1; p < P; ++p) s = s + x[p] * x[p];
  return s;
}

#define DISPATCH_P(P, CALL)                        \
  switch (P) {                                     \
    case 1: { constexpr int PP = 1; CALL; } break; \
    case 2: { constexpr int PP = 2; CALL; } break; \
    case 3: { constexpr int PP = 3; CALL; } break; \
    case 4: { constexpr int PP = 4; CALL; } break; \
    default: throw HipError{hipErrorInvalidValue, "input dimension P must be 1..4", __FILE__, __LINE__}; \
  }
