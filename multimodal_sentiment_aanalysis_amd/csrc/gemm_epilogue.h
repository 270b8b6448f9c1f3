// Fused GEMM epilogue shared by the MFMA kernel, the SIMT kernel and the split-K reducer.
// Processes 4 consecutive columns n..n+3 of row m (N % 4 == 0 is required by the launchers).
#pragma once
#include "gemm.h"

template <typename T> struct Vec4;
template <> struct Vec4<float> {
  static __device__ __forceinline__ f32x4 load(const float* p) { return *(const f32x4*)p; }
  static __device__ __forceinline__ void store(float* p, f32x4 v) { *(f32x4*)p = v; }
};
template <> struct Vec4<bf16> {
  static __device__ __forceinline__ f32x4 load(const bf16* p) {
    bf16x4 t = *(const bf16x4*)p;
    f32x4 v = {(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
    return v;
  }
  static __device__ __forceinline__ void store(bf16* p, f32x4 v) {
    bf16x4 t = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *(bf16x4*)p = t;
  }
};

// `bias4` = the 4 bias values of columns n..n+3 already in registers (zeros when there is no bias): the MFMA kernel
// loads them once per column group instead of once per 16x16 tile.
template <typename T>
__device__ __forceinline__ void gemm_epilogue4b(const GemmParams& p, int m, int n, f32x4 v, f32x4 bias4) {
  v += bias4;
  if (p.C2 && p.c2_gelu_grad) {
    f32x4 d;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const GeluPair gp = gelu_erf_both(v[r]); v[r] = gp.y; d[r] = gp.dy; }
    Vec4<T>::store((T*)p.C2 + (long)m * p.ldc2 + n, d);
  } else {
    if (p.C2) Vec4<T>::store((T*)p.C2 + (long)m * p.ldc2 + n, v);
    if (p.act != MMSA_ACT_NONE && !p.act_after_add) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
    }
  }
  if (p.mul) {
    f32x4 x = Vec4<T>::load((const T*)p.mul + (long)m * p.ldmul + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= p.mul_is_factor ? x[r] : gelu_erf_grad(x[r]);
  }
  if (p.add) v += Vec4<T>::load((const T*)p.add + (long)m * p.ldadd + n);
  if (p.act != MMSA_ACT_NONE && p.act_after_add) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
  }
  if (p.out_f32) {
    float* c = (float*)p.C + (long)m * p.ldc + n;
    if (p.accumulate) v += *(const f32x4*)c;
    *(f32x4*)c = v;
  } else {
    Vec4<T>::store((T*)p.C + (long)m * p.ldc + n, v);
  }
}

template <typename T>
__device__ __forceinline__ void gemm_epilogue4(const GemmParams& p, int m, int n, f32x4 v) {
  if (p.bias) v += *(const f32x4*)(p.bias + n);
  if (p.C2 && p.c2_gelu_grad) {
    f32x4 d;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const GeluPair gp = gelu_erf_both(v[r]); v[r] = gp.y; d[r] = gp.dy; }
    Vec4<T>::store((T*)p.C2 + (long)m * p.ldc2 + n, d);
  } else {
    if (p.C2) Vec4<T>::store((T*)p.C2 + (long)m * p.ldc2 + n, v);
    if (p.act != MMSA_ACT_NONE && !p.act_after_add) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
    }
  }
  if (p.mul) {
    f32x4 x = Vec4<T>::load((const T*)p.mul + (long)m * p.ldmul + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= p.mul_is_factor ? x[r] : gelu_erf_grad(x[r]);
  }
  if (p.add) v += Vec4<T>::load((const T*)p.add + (long)m * p.ldadd + n);
  if (p.act != MMSA_ACT_NONE && p.act_after_add) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
  }
  if (p.out_f32) {
    float* c = (float*)p.C + (long)m * p.ldc + n;
    if (p.accumulate) v += *(const f32x4*)c;
    *(f32x4*)c = v;
  } else {
    Vec4<T>::store((T*)p.C + (long)m * p.ldc + n, v);
  }
}

// scalar form for shapes whose N (or a leading dimension) is not a multiple of 4 (the 3-class heads)
template <typename T>
__device__ __forceinline__ void gemm_epilogue1(const GemmParams& p, int m, int n, float v) {
  if (p.bias) v += p.bias[n];
  if (p.C2 && p.c2_gelu_grad) {
    const GeluPair gp = gelu_erf_both(v);
    v = gp.y;
    ((T*)p.C2)[(long)m * p.ldc2 + n] = from_f32<T>(gp.dy);
  } else {
    if (p.C2) ((T*)p.C2)[(long)m * p.ldc2 + n] = from_f32<T>(v);
    if (p.act != MMSA_ACT_NONE && !p.act_after_add) v = apply_act(v, p.act);
  }
  if (p.mul) {
    const float x = to_f32<T>(((const T*)p.mul)[(long)m * p.ldmul + n]);
    v *= p.mul_is_factor ? x : gelu_erf_grad(x);
  }
  if (p.add) v += to_f32<T>(((const T*)p.add)[(long)m * p.ldadd + n]);
  if (p.act != MMSA_ACT_NONE && p.act_after_add) v = apply_act(v, p.act);
  if (p.out_f32) {
    float* c = (float*)p.C + (long)m * p.ldc + n;
    if (p.accumulate) v += *c;
    *c = v;
  } else {
    ((T*)p.C)[(long)m * p.ldc + n] = from_f32<T>(v);
  }
}

// Split-K second pass: sums the fp32 partial slabs and applies the epilogue. 256 threads = (256 / L) float4 outputs x
// L slab lanes: lane l adds slabs l, l+L, ... in order, then the lanes are added in order through LDS (fixed
// summation tree -> bitwise reproducible). L > 1 is picked by the launcher when the output is so small that one
// thread per output would leave the chip empty and walk hundreds of slabs serially (stage-1 weight gradients:
// 64x64 outputs x 256 slabs took 75 us on 4 workgroups).
template <typename T>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmParams p, int L) {
  __shared__ f32x4 red[256];
  // timing stamps from every 16th block only: 2 x 2048 same-address atomics serialise in L2 and cost the launch ~29 us
  // (3136x512x2048 split 2: 26 us by HIP events, 55 us by its own stamps)
  const bool stamped = (blockIdx.x & 15) == 0 || blockIdx.x == gridDim.x - 1;
  if (stamped) stamp_begin(p.stamp);
  const long total4 = (long)p.M * p.N / 4;
  const int opb = 256 / L;
  const int l = threadIdx.x / opb, o = threadIdx.x - l * opb;
  const long slab = (long)p.M * p.N;
  for (long base = (long)blockIdx.x * opb; base < total4; base += (long)gridDim.x * opb) {
    const long i = base + o;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < total4)
      for (int s = l; s < p.split_k; s += L) v += *(const f32x4*)(p.ws + (long)s * slab + i * 4);
    if (L > 1) {
      __syncthreads();
      red[threadIdx.x] = v;
      __syncthreads();
      if (l == 0)
        for (int j = 1; j < L; ++j) v += red[j * opb + o];
    }
    if (l == 0 && i < total4) {
      const long e = i * 4;
      const int m = (int)(e / p.N), n = (int)(e - (long)m * p.N);
      gemm_epilogue4<T>(p, m, n, v);
    }
  }
  if (stamped) stamp_end(p.stamp);
}

template <typename T>
static inline void launch_splitk_reduce(const GemmParams& p, hipStream_t st) {
  const long total4 = (long)p.M * p.N / 4;
  int L = 1;
  while (L < 64 && L * 2 <= p.split_k && (total4 * L + 255) / 256 < 512) L *= 2;
  long blocks = (total4 * L + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel<T>, dim3((int)blocks), dim3(256), 0, st, p, L);
}
