// Layout / pooling kernels of the ResNet side (all HBM-bound, NHWC, 4-channel vector accesses):
// stem im2col straight from the NCHW fp32 image, 3x3/2 max-pool with saved argmax, global average pool,
// storage casts of the working weights, and small copies.
#include "common.h"
#include "gemm_epilogue.h"
#include "ops.h"

// col[m][k], m = (b, oy, ox), k = (ky*KW + kx)*Cin + ci for k < KH*KW*Cin, zero for the padding up to Kpad
template <typename T>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ img, T* __restrict__ col, int B, int Cin,
                                                          int H, int W, int OH, int OW, int KH, int KW, int stride, int pad,
                                                          int Kpad) {
  const int kch = Kpad / 4;
  const long total = (long)B * OH * OW * kch;
  const int Kreal = KH * KW * Cin;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / kch;
    const int k0 = (int)(i - m * kch) * 4;
    const int ox = (int)(m % OW);
    const long t = m / OW;
    const int oy = (int)(t % OH), b = (int)(t / OH);
    f32x4 v = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + e;
      if (k < Kreal) {
        const int ci = k % Cin, tap = k / Cin;
        const int kx = tap % KW, ky = tap / KW;
        const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[e] = img[(((long)b * Cin + ci) * H + iy) * W + ix];
      }
    }
    Vec4<T>::store(col + m * Kpad + k0, v);
  }
}

// the ResNet stem (7x7, 3 channels): compile-time divisors, 8 elements (16 bytes of bf16) per lane. The generic
// kernel spends ~100 VALU per 4 elements on runtime div/mod (0.29 ms for 308 MB = 1.1 TB/s).
template <typename T, int CIN, int KW_, int KH_>
__global__ __launch_bounds__(256) void stem_im2col_fixed_kernel(const float* __restrict__ img, T* __restrict__ col, int B,
                                                                int H, int W, int OH, int OW, int stride, int pad, int Kpad) {
  const int kch = Kpad / 8;
  const long total = (long)B * OH * OW * kch;
  constexpr int Kreal = KH_ * KW_ * CIN;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / kch;
    const int k0 = (int)(i - m * kch) * 8;
    const int ox = (int)(m % OW);
    const long t = m / OW;
    const int oy = (int)(t % OH), b = (int)(t / OH);
    const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
    const float* ib = img + (long)b * CIN * H * W;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + e;
      v[e] = 0.f;
      if (k < Kreal) {
        const int ci = k % CIN, tap = k / CIN;
        const int kx = tap % KW_, ky = tap / KW_;
        const int iy = iy0 + ky, ix = ix0 + kx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v[e] = ib[((long)ci * H + iy) * W + ix];
      }
    }
    T* dst = col + m * Kpad + k0;
    Vec4<T>::store(dst, f32x4{v[0], v[1], v[2], v[3]});
    Vec4<T>::store(dst + 4, f32x4{v[4], v[5], v[6], v[7]});
  }
}

// Same matrix through an LDS transpose. The chunk-per-lane forms read the image with every lane of a wave on a
// different (channel, tap): ~24 cache lines per load instruction, address-coalescing bound at 1.5-1.7 TB/s of output
// (180-200 us for the 308 MB matrix). Here a block owns 64 consecutive output pixels; for each k the 64 lanes of a wave
// read the same (channel, tap) of 64 neighbouring pixels (one or two image rows, stride-2 floats: 4-5 lines), drop the
// bf16 value into LDS [pixel][k], and the block then streams its 64 x Kpad rows — one contiguous 24 KB range of the
// matrix — out with 16-byte stores.
template <typename T, int CIN, int KW_, int KH_, int KPAD>
__global__ __launch_bounds__(256) void stem_im2col_lds_kernel(const float* __restrict__ img, T* __restrict__ col, int B, int H,
                                                              int W, int OH, int OW, int stride, int pad, FastDiv fd_ow,
                                                              FastDiv fd_oh) {
  constexpr int Kreal = KH_ * KW_ * CIN, PIX = 64, LDW = KPAD + 2;  // row pitch 97 dwords (bf16): conflict-free columns
  __shared__ T tile[PIX * LDW];
  const int npix = B * OH * OW;
  const int p = threadIdx.x & 63, kg = threadIdx.x >> 6;
  for (int m0 = blockIdx.x * PIX; m0 < npix; m0 += gridDim.x * PIX) {
    const int m = m0 + p;
    const bool pv = m < npix;
    const uint32_t t = fd_div((uint32_t)(pv ? m : 0), fd_ow);
    const int ox = (pv ? m : 0) - (int)t * OW;
    const uint32_t b = fd_div(t, fd_oh);
    const int oy = (int)t - (int)b * OH;
    const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
    const float* ib = img + (long)b * CIN * H * W;
#pragma unroll 4
    for (int k = kg; k < KPAD; k += 4) {
      const int ci = k % CIN, tap = k / CIN;
      const int kx = tap % KW_, ky = tap / KW_;
      const int iy = iy0 + ky, ix = ix0 + kx;
      const bool ok = pv && k < Kreal && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const float x = ib[ok ? (ci * H + iy) * W + ix : 0];
      tile[p * LDW + k] = from_f32<T>(ok ? x : 0.f);
    }
    __syncthreads();
    constexpr int CH = KPAD * (int)sizeof(T) / 16;  // 16-byte chunks per row
    for (int q = threadIdx.x; q < PIX * CH; q += 256) {
      const int row = q / CH, c = q - row * CH;
      if (m0 + row < npix) {
        const uint32_t* src = (const uint32_t*)(tile + row * LDW) + c * 4;
        const uint4 v = make_uint4(src[0], src[1], src[2], src[3]);
        *(uint4*)((char*)(col + (long)(m0 + row) * KPAD) + c * 16) = v;
      }
    }
    __syncthreads();
  }
}

int stem_im2col(int dtype, const float* img, void* col, int B, int Cin, int H, int W, int OH, int OW, int KH, int KW,
                int stride, int pad, int Kpad, hipStream_t st) {
  if (Kpad % 4) return MMSA_ERR_ARG;
  if (Cin == 3 && KH == 7 && KW == 7 && Kpad == 192 && dtype == MMSA_BF16 && (long)B * OH * OW < 0x7FFFFF00L) {
    const int grid = (int)min(((long)B * OH * OW + 63) / 64, 8192L);
    hipLaunchKernelGGL((stem_im2col_lds_kernel<bf16, 3, 7, 7, 192>), dim3(grid), dim3(256), 0, st, img, (bf16*)col, B, H, W, OH,
                       OW, stride, pad, make_fastdiv((uint32_t)OW), make_fastdiv((uint32_t)OH));
    MMSA_CHECK_LAUNCH();
    return MMSA_OK;
  }
  if (Cin == 3 && KH == 7 && KW == 7 && Kpad % 8 == 0) {
    const long total8 = (long)B * OH * OW * (Kpad / 8);
    const int grid8 = (int)min((total8 + 255) / 256, 16384L);
    if (dtype == MMSA_BF16)
      hipLaunchKernelGGL((stem_im2col_fixed_kernel<bf16, 3, 7, 7>), dim3(grid8), dim3(256), 0, st, img, (bf16*)col, B, H, W, OH,
                         OW, stride, pad, Kpad);
    else
      hipLaunchKernelGGL((stem_im2col_fixed_kernel<float, 3, 7, 7>), dim3(grid8), dim3(256), 0, st, img, (float*)col, B, H, W,
                         OH, OW, stride, pad, Kpad);
    MMSA_CHECK_LAUNCH();
    return MMSA_OK;
  }
  const long total = (long)B * OH * OW * (Kpad / 4);
  const int grid = (int)min((total + 255) / 256, 8192L);
  if (dtype == MMSA_BF16)
    hipLaunchKernelGGL(stem_im2col_kernel<bf16>, dim3(grid), dim3(256), 0, st, img, (bf16*)col, B, Cin, H, W, OH, OW, KH,
                       KW, stride, pad, Kpad);
  else
    hipLaunchKernelGGL(stem_im2col_kernel<float>, dim3(grid), dim3(256), 0, st, img, (float*)col, B, Cin, H, W, OH, OW, KH,
                       KW, stride, pad, Kpad);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// ------------------------------------------------------------------------------------------------ max pool 3x3/2 pad 1
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          unsigned char* __restrict__ idx, int B, int H, int W, int C, int OH,
                                                          int OW) {
  const int c4n = C / 4;
  const long total = (long)B * OH * OW * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const long m = i / c4n;
    const int ox = (int)(m % OW);
    const long t = m / OW;
    const int oy = (int)(t % OH), b = (int)(t / OH);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool first = true;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 + ky - 1;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 + kx - 1;
        if (ix < 0 || ix >= W) continue;
        const f32x4 v = Vec4<T>::load(x + (((long)b * H + iy) * W + ix) * C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (first || v[e] > best[e]) { best[e] = v[e]; bi[e] = ky * 3 + kx; }
        first = false;
      }
    }
    Vec4<T>::store(y + m * C + c, best);
    *(uchar4*)(idx + m * C + c) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                          T* __restrict__ dx, int B, int H, int W, int C, int OH, int OW) {
  const int c4n = C / 4;
  const long total = (long)B * H * W * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const long m = i / c4n;
    const int ix = (int)(m % W);
    const long t = m / W;
    const int iy = (int)(t % H), b = (int)(t / H);
    // the (at most 2 x 2) output windows [2o-1, 2o+1] that contain this pixel: o = i/2 and, for odd i, (i+1)/2.
    // Branch-free: all four candidates are loaded (clamped addresses) and the invalid ones masked, so the eight loads
    // are in flight together instead of one round trip per taken branch.
    f32x4 g = {0, 0, 0, 0};
    uchar4 id[4];
    f32x4 d[4];
    int tap[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oy = iy / 2 + (q >> 1), ox = ix / 2 + (q & 1);
      const bool ok = ((q >> 1) == 0 || (iy & 1)) && ((q & 1) == 0 || (ix & 1)) && oy < OH && ox < OW;
      const int oyc = ok ? oy : 0, oxc = ok ? ox : 0;
      const long om = ((long)b * OH + oyc) * OW + oxc;
      id[q] = *(const uchar4*)(idx + om * C + c);
      d[q] = Vec4<T>::load(dy + om * C + c);
      tap[q] = ok ? (iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1)) : -1;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (id[q].x == tap[q]) g[0] += d[q][0];
      if (id[q].y == tap[q]) g[1] += d[q][1];
      if (id[q].z == tap[q]) g[2] += d[q][2];
      if (id[q].w == tap[q]) g[3] += d[q][3];
    }
    Vec4<T>::store(dx + m * C + c, g);
  }
}

// bf16, C % 8 == 0 (round 4): 8 channels = 16 bytes per lane, 32-bit index arithmetic with exact fast divisions (the general
// kernels above spend ~200 instructions per element group on four 64-bit divisions and move 8 bytes per lane: the backward ran
// at 1.8 TB/s), every tap / window loaded unconditionally (clamped address, masked use) so that all loads of an element group
// are in flight together. Same selection rule as the general kernels: the first valid tap, then strictly greater.
__global__ __launch_bounds__(256) void maxpool_fwd8_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                           unsigned char* __restrict__ idx, int total, int H, int W, int C,
                                                           int OH, int OW, FastDiv fd_c8, FastDiv fd_ow, FastDiv fd_oh) {
  const int c8n = C >> 3;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const uint32_t m = fd_div((uint32_t)i, fd_c8);
    const int c = (i - (int)m * c8n) * 8;
    const uint32_t t = fd_div(m, fd_ow);
    const int ox = (int)m - (int)t * OW;
    const uint32_t b = fd_div(t, fd_oh);
    const int oy = (int)t - (int)b * OH;
    bf16x8 v[9];
    bool ok[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
        const bool in = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        ok[ky * 3 + kx] = in;
        const int iyc = in ? iy : 0, ixc = in ? ix : 0;
        v[ky * 3 + kx] = *(const bf16x8*)(x + ((long)((int)b * H + iyc) * W + ixc) * C + c);
      }
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    bool first = true;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)v[tp][e];
        const bool take = ok[tp] && (first || f > best[e]);
        best[e] = take ? f : best[e];
        bi[e] = take ? tp : bi[e];
      }
      first = first && !ok[tp];
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)best[e];
    *(bf16x8*)(y + (long)m * C + c) = o;
    const unsigned lo = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
    const unsigned hi = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
    *(uint2*)(idx + (long)m * C + c) = make_uint2(lo, hi);
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd8_kernel(const bf16* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                           bf16* __restrict__ dx, int total, int H, int W, int C, int OH, int OW,
                                                           FastDiv fd_c8, FastDiv fd_w, FastDiv fd_h) {
  const int c8n = C >> 3;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const uint32_t m = fd_div((uint32_t)i, fd_c8);
    const int c = (i - (int)m * c8n) * 8;
    const uint32_t t = fd_div(m, fd_w);
    const int ix = (int)m - (int)t * W;
    const uint32_t b = fd_div(t, fd_h);
    const int iy = (int)t - (int)b * H;
    bf16x8 d[4];
    uint2 id[4];
    int tap[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // the (at most 2 x 2) output windows that contain this pixel (maxpool_bwd_kernel)
      const int oy = (iy >> 1) + (q >> 1), ox = (ix >> 1) + (q & 1);
      const bool ok = ((q >> 1) == 0 || (iy & 1)) && ((q & 1) == 0 || (ix & 1)) && oy < OH && ox < OW;
      const int oyc = ok ? oy : 0, oxc = ok ? ox : 0;
      const long om = (long)((int)b * OH + oyc) * OW + oxc;
      id[q] = *(const uint2*)(idx + om * C + c);
      d[q] = *(const bf16x8*)(dy + om * C + c);
      tap[q] = ok ? (iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1)) : -1;
    }
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int sel = (int)(((e < 4 ? id[q].x : id[q].y) >> (8 * (e & 3))) & 0xFFu);
        g[e] += sel == tap[q] ? (float)d[q][e] : 0.f;
      }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)g[e];
    *(bf16x8*)(dx + (long)m * C + c) = o;
  }
}

int maxpool_fwd(int dtype, const void* x, void* y, unsigned char* idx, int B, int H, int W, int C, hipStream_t st) {
  if (C % 4) return MMSA_ERR_ARG;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  if (dtype == MMSA_BF16 && !(C % 8) && (long)B * H * W * C < 0x7FFFFF00L && !mmsa_disabled("pool8")) {
    const int total8 = B * OH * OW * (C / 8);
    hipLaunchKernelGGL(maxpool_fwd8_kernel, dim3(min((total8 + 255) / 256, 16384)), dim3(256), 0, st, (const bf16*)x, (bf16*)y, idx,
                       total8, H, W, C, OH, OW, make_fastdiv((uint32_t)(C / 8)), make_fastdiv((uint32_t)OW), make_fastdiv((uint32_t)OH));
    MMSA_CHECK_LAUNCH();
    return MMSA_OK;
  }
  const long total = (long)B * OH * OW * (C / 4);
  const int grid = (int)min((total + 255) / 256, 8192L);
  if (dtype == MMSA_BF16)
    hipLaunchKernelGGL(maxpool_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)x, (bf16*)y, idx, B, H, W, C, OH,
                       OW);
  else
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y, idx, B, H, W, C,
                       OH, OW);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

int maxpool_bwd(int dtype, const void* dy, const unsigned char* idx, void* dx, int B, int H, int W, int C, hipStream_t st) {
  if (C % 4) return MMSA_ERR_ARG;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  if (dtype == MMSA_BF16 && !(C % 8) && (long)B * H * W * C < 0x7FFFFF00L && !mmsa_disabled("pool8")) {
    const int total8 = B * H * W * (C / 8);
    hipLaunchKernelGGL(maxpool_bwd8_kernel, dim3(min((total8 + 255) / 256, 16384)), dim3(256), 0, st, (const bf16*)dy, idx, (bf16*)dx,
                       total8, H, W, C, OH, OW, make_fastdiv((uint32_t)(C / 8)), make_fastdiv((uint32_t)W), make_fastdiv((uint32_t)H));
    MMSA_CHECK_LAUNCH();
    return MMSA_OK;
  }
  const long total = (long)B * H * W * (C / 4);
  const int grid = (int)min((total + 255) / 256, 8192L);
  if (dtype == MMSA_BF16)
    hipLaunchKernelGGL(maxpool_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)dy, idx, (bf16*)dx, B, H, W, C,
                       OH, OW);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dy, idx, (float*)dx, B, H, W,
                       C, OH, OW);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// ------------------------------------------------------------------------------------------------ global average pool
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int HW, int C) {
  const int c4n = C / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * c4n) return;
  const int b = (int)(i / c4n), c = (int)(i % c4n) * 4;
  f32x4 s = {0, 0, 0, 0};
  for (int p = 0; p < HW; ++p) s += Vec4<T>::load(x + ((long)b * HW + p) * C + c);
  Vec4<T>::store(y + (long)b * C + c, s * (1.0f / (float)HW));
}
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int HW, int C) {
  const int c4n = C / 4;
  const long total = (long)B * HW * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const long bp = i / c4n;
    const int b = (int)(bp / HW);
    Vec4<T>::store(dx + bp * C + c, Vec4<T>::load(dy + (long)b * C + c) * (1.0f / (float)HW));
  }
}
int avgpool_fwd(int dtype, const void* x, void* y, int B, int HW, int C, hipStream_t st) {
  if (C % 4) return MMSA_ERR_ARG;
  const int grid = cdiv((long)B * (C / 4), 256);
  if (dtype == MMSA_BF16)
    hipLaunchKernelGGL(avgpool_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)x, (bf16*)y, B, HW, C);
  else
    hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y, B, HW, C);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
int avgpool_bwd(int dtype, const void* dy, void* dx, int B, int HW, int C, hipStream_t st) {
  if (C % 4) return MMSA_ERR_ARG;
  const long total = (long)B * HW * (C / 4);
  const int grid = (int)min((total + 255) / 256, 8192L);
  if (dtype == MMSA_BF16)
    hipLaunchKernelGGL(avgpool_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)dy, (bf16*)dx, B, HW, C);
  else
    hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dy, (float*)dx, B, HW, C);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// ------------------------------------------------------------------------------------------------ casts / copies
// dst[i] = (T) src[i]  (n % 4 == 0 handled by the tail loop)
template <typename T>
__global__ __launch_bounds__(256) void cast_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    Vec4<T>::store(dst + i * 4, *(const f32x4*)(src + i * 4));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] = from_f32<T>(src[n4 * 4 + threadIdx.x]);
}
int cast_f32(int dtype, const float* src, void* dst, long n, hipStream_t st) {
  if (n <= 0) return MMSA_OK;
  const int grid = (int)min((n / 4 + 255) / 256 + 1, 8192L);
  if (dtype == MMSA_BF16) hipLaunchKernelGGL(cast_f32_kernel<bf16>, dim3(grid), dim3(256), 0, st, src, (bf16*)dst, n);
  else hipLaunchKernelGGL(cast_f32_kernel<float>, dim3(grid), dim3(256), 0, st, src, (float*)dst, n);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// dst fp32 <- src bf16 (the reduced gradient payload of a data-parallel step coming back from the all-reduce, fused.py)
__global__ __launch_bounds__(256) void widen_bf16_kernel(const bf16* __restrict__ src, float* __restrict__ dst, long n) {
  const long n8 = n / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const bf16x8 v = *(const bf16x8*)(src + i * 8);
    *(f32x4*)(dst + i * 8) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    *(f32x4*)(dst + i * 8 + 4) = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[n8 * 8 + threadIdx.x] = (float)src[n8 * 8 + threadIdx.x];
}
int widen_bf16(const void* src, float* dst, long n, hipStream_t st) {
  if (n <= 0) return MMSA_OK;
  const int grid = (int)min((n / 8 + 255) / 256 + 1, 4096L);
  hipLaunchKernelGGL(widen_bf16_kernel, dim3(grid), dim3(256), 0, st, (const bf16*)src, dst, n);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// dst[r][0:cols_dst] = src[r][0:cols_src] zero-padded (stem weight [64][147] -> [64][Kpad]) in storage type T
template <typename T>
__global__ void pad_rows_kernel(const float* __restrict__ src, T* __restrict__ dst, int rows, int cols_src, int cols_dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols_dst) return;
  const int r = i / cols_dst, c = i - r * cols_dst;
  dst[i] = from_f32<T>(c < cols_src ? src[(long)r * cols_src + c] : 0.f);
}
int pad_rows(int dtype, const float* src, void* dst, int rows, int cols_src, int cols_dst, hipStream_t st) {
  const int grid = cdiv((long)rows * cols_dst, 256);
  if (dtype == MMSA_BF16)
    hipLaunchKernelGGL(pad_rows_kernel<bf16>, dim3(grid), dim3(256), 0, st, src, (bf16*)dst, rows, cols_src, cols_dst);
  else
    hipLaunchKernelGGL(pad_rows_kernel<float>, dim3(grid), dim3(256), 0, st, src, (float*)dst, rows, cols_src, cols_dst);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// dst[r][0:cols_dst] (+)= src[r][0:cols_dst] with src row stride cols_src >= cols_dst (stem wgrad un-padding)
__global__ void unpad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols_src,
                                  int cols_dst, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols_dst) return;
  const int r = i / cols_dst, c = i - r * cols_dst;
  const float v = src[(long)r * cols_src + c];
  dst[i] = accumulate ? dst[i] + v : v;
}
int unpad_rows(const float* src, float* dst, int rows, int cols_src, int cols_dst, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(unpad_rows_kernel, dim3(cdiv((long)rows * cols_dst, 256)), dim3(256), 0, st, src, dst, rows, cols_src,
                     cols_dst, accumulate);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
