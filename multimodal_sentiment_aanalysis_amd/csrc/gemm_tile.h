// LDS tile layouts and fragment readers shared by the MFMA GEMM kernels (gemm_mfma.hip, gemm_mfma2.hip).
//
// k-contiguous tile [rows][64 k] (128-byte rows): 16-byte chunk kc of row r sits at chunk position kc ^ (r & 7), read
// with ds_read_b128. k-major tile [64 k][128 cols] (256-byte k-rows): 32-byte XOR swizzle, read with
// ds_read_b64_tr_b16 (the CDNA4 LDS transpose read), so NN / TN GEMMs need no transposed copies in HBM.
// Both images are lane-linear per LDS-DMA wave-instruction: the swizzle is applied to the per-lane SOURCE address.
#pragma once
#include "gemm.h"

__device__ __forceinline__ int kmajor_swz(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// LDS byte offset of 16-byte chunk kc (0..7) of row `row` in a k-contiguous tile [128][64]
__device__ __forceinline__ int kc_off(int row, int kc) { return row * 128 + ((kc ^ (row & 7)) << 4); }
// LDS byte offset of 16-byte chunk cc (0..15) of k-row `krow` in a k-major tile [64][128]
__device__ __forceinline__ int km_off(int krow, int cc) {
  return krow * 256 + ((((cc >> 1) ^ kmajor_swz(krow)) << 5) | ((cc & 1) << 4));
}

template <bool KM>
__device__ __forceinline__ bf16x8 read_frag(const unsigned char* tile, int base_rc, int kk, int lane) {
  const int r16 = lane & 15, g = lane >> 4;
  if constexpr (!KM) {
    const int row = base_rc + r16;
    return *(const bf16x8*)(tile + kc_off(row, kk * 4 + g));
  } else {
    const int q = r16 >> 2, pp = r16 & 3;
    const int col = base_rc + 4 * pp;
    const int cc = col >> 3;
    const int kb = kk * 32 + 8 * g;
    const int a0 = km_off(kb + q, cc) + ((pp & 1) << 3);
    const int a1 = km_off(kb + 4 + q, cc) + ((pp & 1) << 3);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + a1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
  }
}

struct RowPix {  // decomposed pixel of a gathered row
  int img, yb, xb;  // img < 0: row invalid; yb = y*mul + off, xb = x*mul + off
};

__device__ __forceinline__ RowPix decompose_pixel(const ConvGeom& g, int m, int limit) {
  RowPix r;
  if (m >= limit) {
    r.img = -1; r.yb = 0; r.xb = 0;
    return r;
  }
  const uint32_t img = fd_div((uint32_t)m, g.fd_ghw);
  const uint32_t rem = (uint32_t)m - img * (uint32_t)(g.GH * g.GW);
  const uint32_t y = fd_div(rem, g.fd_gw);
  const uint32_t x = rem - y * (uint32_t)g.GW;
  r.img = (int)img;
  r.yb = (int)y * g.mul + g.off;
  r.xb = (int)x * g.mul + g.offx;
  return r;
}

// element offset of the source pixel for (row pixel, tap), or -1 when the tap is invalid
__device__ __forceinline__ long tap_src(const ConvGeom& g, const RowPix& r, int ky, int kx) {
  if (r.img < 0) return -1;
  int sy = r.yb + ky * g.kmul, sx = r.xb + kx * g.kmul;
  if (g.div > 1) {
    if (sy < 0 || sx < 0) return -1;
    if (g.div == 2) {
      if ((sy | sx) & 1) return -1;
      sy >>= 1; sx >>= 1;
    } else {
      if (sy % g.div || sx % g.div) return -1;
      sy /= g.div; sx /= g.div;
    }
  }
  if ((unsigned)sy >= (unsigned)g.SH || (unsigned)sx >= (unsigned)g.SW) return -1;
  return (((long)r.img * g.SH + sy) * g.SW + sx) * g.src_pix_stride;
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

