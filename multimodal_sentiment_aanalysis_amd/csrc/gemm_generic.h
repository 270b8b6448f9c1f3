// Fully general per-element operand loaders of the GEMM descriptor (every gather mode of gemm.h): the SIMT kernel's only
// loaders, and the slow path of the fp32-MFMA kernel for tiles its vector loaders do not cover (edges, odd strides).
#pragma once
#include "gemm.h"

__device__ __forceinline__ long simt_tap_src(const ConvGeom& g, int pix, int ky, int kx) {
  const int ghw = g.GH * g.GW;
  const int img = pix / ghw, rem = pix - img * ghw;
  const int y = rem / g.GW, x = rem - y * g.GW;
  int sy = y * g.mul + g.off + ky * g.kmul, sx = x * g.mul + g.offx + kx * g.kmul;
  if (g.div > 1) {
    if (sy < 0 || sx < 0 || sy % g.div || sx % g.div) return -1;
    sy /= g.div; sx /= g.div;
  }
  if (sy < 0 || sx < 0 || sy >= g.SH || sx >= g.SW) return -1;
  return (((long)img * g.SH + sy) * g.SW + sx) * g.src_pix_stride;
}

template <typename T>
__device__ __forceinline__ float simt_load_a(const GemmParams& p, int m, int k, int kend) {
  if (m >= p.M || k >= kend) return 0.f;
  const T* A = (const T*)p.A;
  if (p.gather == 1) {
    const int tap = k / p.g.cper, c = k - tap * p.g.cper;
    const int ky = tap / p.g.KW, kx = tap - ky * p.g.KW;
    const long s = simt_tap_src(p.g, m, ky, kx);
    return s < 0 ? 0.f : to_f32<T>(A[s + c]);
  }
  return to_f32<T>(p.a_kmajor ? A[(long)k * p.lda + m] : A[(long)m * p.lda + k]);
}

template <typename T>
__device__ __forceinline__ float simt_load_b(const GemmParams& p, int k, int n, int kend) {
  if (n >= p.N || k >= kend) return 0.f;
  const T* B = (const T*)p.B;
  if (p.gather == 2) {
    const int tap = n / p.g.cper, c = n - tap * p.g.cper;
    const int ky = tap / p.g.KW, kx = tap - ky * p.g.KW;
    const long s = simt_tap_src(p.g, k, ky, kx);
    return s < 0 ? 0.f : to_f32<T>(B[s + c]);
  }
  if (p.gather == 1 && p.b_kmajor) {
    const int tap = k / p.g.cper, c = k - tap * p.g.cper;
    const int ky = tap / p.g.KW, kx = tap - ky * p.g.KW;
    return to_f32<T>(B[(long)c * p.ldb + b_tap_offset(p, ky, kx) + n]);
  }
  return to_f32<T>(p.b_kmajor ? B[(long)k * p.ldb + n] : B[(long)n * p.ldb + k]);
}

