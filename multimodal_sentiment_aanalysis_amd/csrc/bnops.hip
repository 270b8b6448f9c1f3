// BatchNorm over the rows of an [M][C] matrix (NHWC activations: M = B*H*W; fusion head: M = B), training
// statistics, fused apply (+ residual, + ReLU/GELU) and backward. HBM-bound: every kernel streams rows with
// 4-element vector accesses; a block's 256 threads are laid out as (C/4 column threads) x (row lanes) so that
// even C = 64 keeps all lanes busy on fully contiguous rows. Reductions are two-stage and order-fixed
// (per-chunk partials -> finalize in double), hence bitwise reproducible.
//
// Semantics follow nn.BatchNorm1d/2d as the reference uses them (MultimodalModel.py:181,186,194,380,419,423;
// eps 1e-5, momentum 0.1): normalise with the biased batch variance, update running_var with the unbiased one.
#include <stdlib.h>
#include "common.h"
#include "gemm_epilogue.h"
#include "ops.h"

#define BN_CHUNKS 1024

// 8 consecutive channels per lane: 16 bytes of bf16 (one dwordx4 per lane, 1 KiB per wave instruction) or 32 bytes of
// fp32. The 8-byte form (4 channels per lane) ran the streaming kernels at 2.6-3.3 TB/s; see profiles/.
struct f32x8 { f32x4 lo, hi; };
template <typename T> struct Vec8;
template <> struct Vec8<float> {
  static __device__ __forceinline__ f32x8 load(const float* p) { return f32x8{*(const f32x4*)p, *(const f32x4*)(p + 4)}; }
  static __device__ __forceinline__ void store(float* p, f32x8 v) { *(f32x4*)p = v.lo; *(f32x4*)(p + 4) = v.hi; }
};
template <> struct Vec8<bf16> {
  static __device__ __forceinline__ f32x8 load(const bf16* p) {
    const bf16x8 t = *(const bf16x8*)p;
    return f32x8{f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]}, f32x4{(float)t[4], (float)t[5], (float)t[6], (float)t[7]}};
  }
  static __device__ __forceinline__ void store(bf16* p, f32x8 v) {
    const bf16x8 t = {(bf16)v.lo[0], (bf16)v.lo[1], (bf16)v.lo[2], (bf16)v.lo[3],
                      (bf16)v.hi[0], (bf16)v.hi[1], (bf16)v.hi[2], (bf16)v.hi[3]};
    *(bf16x8*)p = t;
  }
};
__device__ __forceinline__ f32x8 ld8(const float* p) { return f32x8{*(const f32x4*)p, *(const f32x4*)(p + 4)}; }
__device__ __forceinline__ f32x8 operator+(f32x8 a, f32x8 b) { return f32x8{a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ f32x8 operator-(f32x8 a, f32x8 b) { return f32x8{a.lo - b.lo, a.hi - b.hi}; }
__device__ __forceinline__ f32x8 operator*(f32x8 a, f32x8 b) { return f32x8{a.lo * b.lo, a.hi * b.hi}; }
__device__ __forceinline__ f32x8 operator*(f32x8 a, float b) { return f32x8{a.lo * b, a.hi * b}; }
__device__ __forceinline__ f32x8 zero8() { return f32x8{f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}; }

struct BnMap {
  int cthreads, rlanes, cgroups;  // column threads per block (8 channels each), row lanes per block, grid.y
};
static BnMap bn_map(int C) {
  const int c8 = C / 8;
  int ct = 1;
  while (ct < c8 && ct < 256) ct <<= 1;
  BnMap m;
  m.cthreads = ct;
  m.rlanes = 256 / ct;
  m.cgroups = cdiv(c8, ct);
  return m;
}

// part[chunk][2][C]: sum x, sum x^2 over the chunk's rows
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const T* __restrict__ x, float* __restrict__ part, int M,
                                                               int C, int cthreads, int rows_per_chunk) {
  __shared__ f32x4 red[4][256];
  const int ct = threadIdx.x % cthreads, rl = threadIdx.x / cthreads, rlanes = 256 / cthreads;
  const int col = (blockIdx.y * cthreads + ct) * 8;
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  f32x8 s = zero8(), q = zero8();
  if (col < C) {
    int r = r0 + rl;
    for (; r + 3 * rlanes < r1; r += 4 * rlanes) {  // four independent 16-byte loads in flight per lane
      const f32x8 v0 = Vec8<T>::load(x + (long)r * C + col), v1 = Vec8<T>::load(x + (long)(r + rlanes) * C + col);
      const f32x8 v2 = Vec8<T>::load(x + (long)(r + 2 * rlanes) * C + col), v3 = Vec8<T>::load(x + (long)(r + 3 * rlanes) * C + col);
      s = s + v0; q = q + v0 * v0;
      s = s + v1; q = q + v1 * v1;
      s = s + v2; q = q + v2 * v2;
      s = s + v3; q = q + v3 * v3;
    }
    for (; r < r1; r += rlanes) {
      const f32x8 v = Vec8<T>::load(x + (long)r * C + col);
      s = s + v;
      q = q + v * v;
    }
  }
  red[0][threadIdx.x] = s.lo; red[1][threadIdx.x] = s.hi;
  red[2][threadIdx.x] = q.lo; red[3][threadIdx.x] = q.hi;
  __syncthreads();
  if (rl == 0 && col < C) {
    for (int j = 1; j < rlanes; ++j) {
      s.lo += red[0][j * cthreads + ct]; s.hi += red[1][j * cthreads + ct];
      q.lo += red[2][j * cthreads + ct]; q.hi += red[3][j * cthreads + ct];
    }
    float* ps = part + ((long)blockIdx.x * 2 + 0) * C + col;
    float* pq = part + ((long)blockIdx.x * 2 + 1) * C + col;
    *(f32x4*)ps = s.lo; *(f32x4*)(ps + 4) = s.hi;
    *(f32x4*)pq = q.lo; *(f32x4*)(pq + 4) = q.hi;
  }
}

// finalize kernels: 256 threads = 4 columns x 64 partial-lanes (thread = lane * 4 + column). The 64 lanes of a column
// are summed by a fixed tree: xor-shuffles over the 16 lanes a wave holds, then the four waves through LDS (a single
// thread walking 64 LDS values in double took ~3 us of these 6 us launches; 114 of them per step).
#define BN_FIN_COLS 4
#define BN_FIN_LANES 64
template <typename V, int COLS = BN_FIN_COLS>
__device__ __forceinline__ V bn_fin_reduce(V v, V* red /* [4][COLS] */) {
#pragma unroll
  for (int o = COLS; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);  // lanes with the same column inside the wave
  const int cc = threadIdx.x & (COLS - 1), wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < COLS) red[wave * COLS + cc] = v;
  __syncthreads();
  return red[cc] + red[COLS + cc] + red[2 * COLS + cc] + red[3 * COLS + cc];
}
// COLS columns per workgroup, 256 / COLS partial-lanes per column. COLS = 1 is for the per-slice partials of the convolution
// epilogues (GemmParams::colstat: up to 12544 rows of only 64-256 columns): with 4 columns per workgroup a C = 64 layer ran on 16
// workgroups whose threads each walked 49 rows (10 us against 5.9 for the 1024-chunk partials of the streamed statistics pass)
template <int COLS>
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ part, int chunks, int M, int C,
                                                                float eps, float momentum, float* __restrict__ mean,
                                                                float* __restrict__ invstd, float* __restrict__ running_mean,
                                                                float* __restrict__ running_var) {
  __shared__ double rs[4 * COLS], rq[4 * COLS];
  constexpr int LANES = 256 / COLS;
  const int cc = threadIdx.x & (COLS - 1), r = threadIdx.x / COLS;
  const int c = blockIdx.x * COLS + cc;
  double s = 0, q = 0;
  if (c < C) {
    int b = r;
    for (; b + LANES < chunks; b += 2 * LANES) {  // two rows (four loads) in flight per thread
      const float s0 = part[((long)b * 2 + 0) * C + c], q0 = part[((long)b * 2 + 1) * C + c];
      const float s1 = part[((long)(b + LANES) * 2 + 0) * C + c], q1 = part[((long)(b + LANES) * 2 + 1) * C + c];
      s += s0; q += q0;
      s += s1; q += q1;
    }
    for (; b < chunks; b += LANES) {
      s += part[((long)b * 2 + 0) * C + c];
      q += part[((long)b * 2 + 1) * C + c];
    }
  }
  s = bn_fin_reduce<double, COLS>(s, rs);
  q = bn_fin_reduce<double, COLS>(q, rq);
  if (r != 0 || c >= C) return;
  const double mu = s / M;
  double var = q / M - mu * mu;
  if (var < 0) var = 0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unb = M > 1 ? var * ((double)M / (M - 1)) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
  }
}
static void launch_stats_finalize(const float* part, int chunks, int M, int C, float eps, float momentum, float* mean,
                                  float* invstd, float* running_mean, float* running_var, hipStream_t st) {
  if ((long)chunks * BN_FIN_COLS >= 2048 && C <= 512)
    hipLaunchKernelGGL(bn_stats_finalize_kernel<1>, dim3(C), dim3(256), 0, st, part, chunks, M, C, eps, momentum, mean, invstd,
                       running_mean, running_var);
  else
    hipLaunchKernelGGL(bn_stats_finalize_kernel<BN_FIN_COLS>, dim3(cdiv(C, BN_FIN_COLS)), dim3(256), 0, st, part, chunks, M, C,
                       eps, momentum, mean, invstd, running_mean, running_var);
}

// eval mode: statistics come from the running buffers
__global__ void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                     float eps, int C, float* __restrict__ mean, float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = running_mean[c];
  invstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// inference-mode fold: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale (the scale goes into the
// convolution's weights, the shift into its bias: no BatchNorm kernel at all in a forward-only pass)
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ running_mean, const float* __restrict__ running_var, float eps, int C,
                               float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] * (1.0f / sqrtf(running_var[c] + eps));
  scale[c] = sc;
  shift[c] = beta[c] - running_mean[c] * sc;
}
// w'[o][k] = w[o][k] * scale[o] (fp32 master weights in, storage type out): fold-then-convolve, the classic inference form
template <typename T>
__global__ __launch_bounds__(256) void bn_fold_weights_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                              T* __restrict__ out, int rows, int K, int ld_out) {
  const int o = blockIdx.x;
  const float sc = scale[o];
  for (int k = threadIdx.x; k < K; k += 256) out[(long)o * ld_out + k] = from_f32<T>(w[(long)o * K + k] * sc);
}
// scale / shift of an inference-mode BatchNorm and the convolution weights with the scale folded in: conv(x, w) * scale + shift =
// conv(x, w') + shift. `w` [rows][K] fp32, `wout` [rows][ld_out] in the storage type (ld_out >= K: the stem's zero-padded K).
int bn_fold(int dtype, const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
            int C, float* scale, float* shift, const float* w, void* wout, int K, int ld_out, hipStream_t st) {
  if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || !w || !wout || C <= 0 || K <= 0) return MMSA_ERR_ARG;
  hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, gamma, beta, running_mean, running_var, eps, C, scale, shift);
  if (dtype == MMSA_BF16)
    hipLaunchKernelGGL(bn_fold_weights_kernel<bf16>, dim3(C), dim3(256), 0, st, w, (const float*)scale, (bf16*)wout, C, K, ld_out);
  else
    hipLaunchKernelGGL(bn_fold_weights_kernel<float>, dim3(C), dim3(256), 0, st, w, (const float*)scale, (float*)wout, C, K, ld_out);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// y = act( (x - mean) * invstd * gamma + beta (+ res) )
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const T* __restrict__ res,
                                                       T* __restrict__ y, int M, int C, int cthreads, int act,
                                                       unsigned char* __restrict__ mask) {
  const int ct = threadIdx.x % cthreads, rl = threadIdx.x / cthreads, rlanes = 256 / cthreads;
  const int col = (blockIdx.y * cthreads + ct) * 8;
  if (col >= C) return;
  const f32x8 mu = ld8(mean + col), is = ld8(invstd + col);
  const f32x8 sc = is * ld8(gamma + col);
  const f32x8 sh = ld8(beta + col) - mu * sc;
  const long stride = (long)gridDim.x * rlanes;
  auto one = [&](long r, f32x8 xv, f32x8 rv) {
    f32x8 v = xv * sc + sh;
    if (res) v = v + rv;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v.lo[e] = apply_act(v.lo[e], act); v.hi[e] = apply_act(v.hi[e], act); }
    Vec8<T>::store(y + r * C + col, v);
    if (mask) {
      unsigned bits = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) bits |= (v.lo[e] > 0.f ? 1u << e : 0u) | (v.hi[e] > 0.f ? 16u << e : 0u);
      mask[(r * C + col) >> 3] = (unsigned char)bits;
    }
  };
  for (long r = (long)blockIdx.x * rlanes + rl; r < M; r += stride) {
    const f32x8 x0 = Vec8<T>::load(x + r * C + col);
    f32x8 q0 = zero8();
    if (res) q0 = Vec8<T>::load(res + r * C + col);
    one(r, x0, q0);
  }
}

// backward pass 1: dz = dy * act'(z); partial sums of dz and dz * xhat.  ReLU mask comes from the saved output y
// (y > 0 <=> z > 0, also with a residual) or, when y == null (no residual), is recomputed from x with the forward's own
// expression x*sc + sh, which saves one of the three streamed reads; GELU (no residual in the reference) recomputes z.
__device__ __forceinline__ f32x4 bn_dz4(f32x4 dy, f32x4 xh, f32x4 g, f32x4 b, f32x4 yv, int act) {
  f32x4 dz = dy;
  if (act == MMSA_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) dz[e] = yv[e] > 0.f ? dy[e] : 0.f;
  } else if (act == MMSA_ACT_GELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) dz[e] = dy[e] * gelu_erf_grad(xh[e] * g[e] + b[e]);
  }
  return dz;
}
__device__ __forceinline__ f32x8 bn_dz(f32x8 dy, f32x8 xh, f32x8 g, f32x8 b, f32x8 yv, int act) {
  return f32x8{bn_dz4(dy.lo, xh.lo, g.lo, b.lo, yv.lo, act), bn_dz4(dy.hi, xh.hi, g.hi, b.hi, yv.hi, act)};
}

__device__ __forceinline__ f32x8 mask8(unsigned bits) {  // the saved ReLU mask as a stand-in for the output's sign
  f32x8 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) { v.lo[e] = (bits >> e) & 1u ? 1.f : 0.f; v.hi[e] = (bits >> (4 + e)) & 1u ? 1.f : 0.f; }
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             const T* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ part, int M, int C, int cthreads,
                                                             int rows_per_chunk, int act,
                                                             const unsigned char* __restrict__ mask) {
  __shared__ f32x4 red[4][256];
  const int ct = threadIdx.x % cthreads, rl = threadIdx.x / cthreads, rlanes = 256 / cthreads;
  const int col = (blockIdx.y * cthreads + ct) * 8;
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  f32x8 s = zero8(), q = zero8();
  if (col < C) {
    const f32x8 mu = ld8(mean + col), is = ld8(invstd + col);
    const f32x8 g = ld8(gamma + col), b = ld8(beta + col);
    const f32x8 sc = is * g, sh = b - mu * sc;  // as bn_apply_kernel: the ReLU mask without a saved output (y == null)
    const bool rd_y = act == MMSA_ACT_RELU && y && !mask;
    const bool rd_m = act == MMSA_ACT_RELU && mask;
    auto one = [&](f32x8 xr, f32x8 dv, f32x8 yv) {
      const f32x8 xh = (xr - mu) * is;
      if (act == MMSA_ACT_RELU && !y && !mask) yv = xr * sc + sh;
      const f32x8 dz = bn_dz(dv, xh, g, b, yv, act);
      s = s + dz;
      q = q + dz * xh;
    };
    int r = r0 + rl;
    for (; r + rlanes < r1; r += 2 * rlanes) {  // two rows (4-6 independent 16-byte loads) in flight per lane
      const long o0 = (long)r * C + col, o1 = (long)(r + rlanes) * C + col;
      const f32x8 x0 = Vec8<T>::load(x + o0), x1 = Vec8<T>::load(x + o1);
      const f32x8 d0 = Vec8<T>::load(dy + o0), d1 = Vec8<T>::load(dy + o1);
      f32x8 y0 = zero8(), y1 = zero8();
      if (rd_y) { y0 = Vec8<T>::load(y + o0); y1 = Vec8<T>::load(y + o1); }
      if (rd_m) { y0 = mask8(mask[o0 >> 3]); y1 = mask8(mask[o1 >> 3]); }
      one(x0, d0, y0);
      one(x1, d1, y1);
    }
    for (; r < r1; r += rlanes) {
      const long o0 = (long)r * C + col;
      f32x8 y0 = zero8();
      if (rd_y) y0 = Vec8<T>::load(y + o0);
      if (rd_m) y0 = mask8(mask[o0 >> 3]);
      one(Vec8<T>::load(x + o0), Vec8<T>::load(dy + o0), y0);
    }
  }
  red[0][threadIdx.x] = s.lo; red[1][threadIdx.x] = s.hi;
  red[2][threadIdx.x] = q.lo; red[3][threadIdx.x] = q.hi;
  __syncthreads();
  if (rl == 0 && col < C) {
    for (int j = 1; j < rlanes; ++j) {
      s.lo += red[0][j * cthreads + ct]; s.hi += red[1][j * cthreads + ct];
      q.lo += red[2][j * cthreads + ct]; q.hi += red[3][j * cthreads + ct];
    }
    float* ps = part + ((long)blockIdx.x * 2 + 0) * C + col;
    float* pq = part + ((long)blockIdx.x * 2 + 1) * C + col;
    *(f32x4*)ps = s.lo; *(f32x4*)(ps + 4) = s.hi;
    *(f32x4*)pq = q.lo; *(f32x4*)(pq + 4) = q.hi;
  }
}

// sums[0][C] = dbeta, sums[1][C] = dgamma (kept for pass 2) and (+)= into the parameter gradients
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int chunks, int C,
                                                              float* __restrict__ sums, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate) {
  __shared__ float rs[4 * BN_FIN_COLS], rq[4 * BN_FIN_COLS];
  const int cc = threadIdx.x & (BN_FIN_COLS - 1), r = threadIdx.x / BN_FIN_COLS;
  const int c = blockIdx.x * BN_FIN_COLS + cc;
  float s = 0, q = 0;
  if (c < C)
    for (int b = r; b < chunks; b += BN_FIN_LANES) {
      s += part[((long)b * 2 + 0) * C + c];
      q += part[((long)b * 2 + 1) * C + c];
    }
  s = bn_fin_reduce(s, rs);
  q = bn_fin_reduce(q, rq);
  if (r != 0 || c >= C) return;
  sums[c] = s;
  sums[C + c] = q;
  if (dgamma) {
    dgamma[c] = accumulate ? dgamma[c] + q : q;
    dbeta[c] = accumulate ? dbeta[c] + s : s;
  }
}

// backward pass 2: dx = gamma*invstd*(dz - dbeta/M - xhat*dgamma/M) (training) or gamma*invstd*dz (eval);
// optionally also writes dz (gradient of the residual branch)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const T* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ sums, T* __restrict__ dx,
                                                           T* __restrict__ dres, int M, int C, int cthreads, int act,
                                                           int training, const unsigned char* __restrict__ mask) {
  const int ct = threadIdx.x % cthreads, rl = threadIdx.x / cthreads, rlanes = 256 / cthreads;
  const int col = (blockIdx.y * cthreads + ct) * 8;
  if (col >= C) return;
  const f32x8 mu = ld8(mean + col), is = ld8(invstd + col);
  const f32x8 g = ld8(gamma + col), b = ld8(beta + col);
  f32x8 mb = zero8(), mg = zero8();
  if (training) {
    const float invM = 1.0f / (float)M;
    mb = ld8(sums + col) * invM;
    mg = ld8(sums + C + col) * invM;
  }
  const f32x8 gi = g * is;
  const f32x8 sc = is * g, sh = b - mu * sc;
  const bool rd_y = act == MMSA_ACT_RELU && y && !mask;
  const bool rd_m = act == MMSA_ACT_RELU && mask;
  const long stride = (long)gridDim.x * rlanes;
  auto one = [&](long r, f32x8 xr, f32x8 dv, f32x8 yv) {
    const f32x8 xh = (xr - mu) * is;
    if (act == MMSA_ACT_RELU && !y && !mask) yv = xr * sc + sh;
    const f32x8 dz = bn_dz(dv, xh, g, b, yv, act);
    if (dres) Vec8<T>::store(dres + r * C + col, dz);
    Vec8<T>::store(dx + r * C + col, gi * (dz - mb - xh * mg));
  };
  for (long r = (long)blockIdx.x * rlanes + rl; r < M; r += stride) {
    const long o0 = r * C + col;
    f32x8 y0 = zero8();
    if (rd_y) y0 = Vec8<T>::load(y + o0);
    if (rd_m) y0 = mask8(mask[o0 >> 3]);
    one(r, Vec8<T>::load(x + o0), Vec8<T>::load(dy + o0), y0);
  }
}

// ---- small batches (the fusion head: 64 rows x 128-256 features, fp32) -------------------------------------------------
// One launch instead of three per direction: a workgroup owns 8 features and all rows (<= BN_SMALL_ROWS), keeps its
// elements in registers between the statistics and the apply phase.  Same expressions as the streamed kernels (fp32
// per-lane partial sums, fp64 combine for the forward statistics, fp32 for the backward sums).
#define BN_SMALL_ROWS 256
#define BN_SMALL_RPT (BN_SMALL_ROWS / 32)
__device__ __forceinline__ float bn_dz1(float dy, float xh, float g, float b, float z, int act) {
  if (act == MMSA_ACT_RELU) return z > 0.f ? dy : 0.f;
  if (act == MMSA_ACT_GELU) return dy * gelu_erf_grad(xh * g + b);
  return dy;
}
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ mean,
                                                           float* __restrict__ invstd, float* __restrict__ y, int M, int C,
                                                           float eps, float momentum, int act) {
  __shared__ double rs[32][8], rq[32][8];
  const int cc = threadIdx.x & 7, rl = threadIdx.x >> 3, c = blockIdx.x * 8 + cc;
  float xs[BN_SMALL_RPT];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < BN_SMALL_RPT; ++i) {
    const int r = rl + 32 * i;
    xs[i] = r < M ? x[(long)r * C + c] : 0.f;
    s += xs[i];
    q += xs[i] * xs[i];
  }
  rs[rl][cc] = s;
  rq[rl][cc] = q;
  __syncthreads();
  double S = 0, Q = 0;
#pragma unroll 8
  for (int j = 0; j < 32; ++j) { S += rs[j][cc]; Q += rq[j][cc]; }
  const double mu = S / M;
  double var = Q / M - mu * mu;
  if (var < 0) var = 0;
  const float muf = (float)mu, isf = (float)(1.0 / sqrt(var + (double)eps));
  if (rl == 0) {
    mean[c] = muf;
    invstd[c] = isf;
    if (running_mean) {
      const double unb = M > 1 ? var * ((double)M / (M - 1)) : var;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
  }
  const float sc = isf * gamma[c], sh = beta[c] - muf * sc;
#pragma unroll
  for (int i = 0; i < BN_SMALL_RPT; ++i) {
    const int r = rl + 32 * i;
    if (r < M) y[(long)r * C + c] = apply_act(xs[i] * sc + sh, act);
  }
}
// drop_mask (or null): dy is the gradient behind a Dropout whose keep bytes these are — dropout_bwd_kernel's expression applied on
// load. relu_in: x is relu(pre-activation) and dx is wanted at the pre-activation (Linear -> ReLU -> BatchNorm units) —
// ew2d_kernel's EW_RELU_BWD applied on store. One launch instead of three for the fusion head's units.
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int accumulate, int M, int C, int act,
                                                           const unsigned char* __restrict__ drop_mask, float drop_p,
                                                           int relu_in) {
  __shared__ float rs[32][8], rq[32][8];
  const int cc = threadIdx.x & 7, rl = threadIdx.x >> 3, c = blockIdx.x * 8 + cc;
  const float mu = mean[c], is = invstd[c], g = gamma[c], b = beta[c];
  const float sc = is * g, sh = b - mu * sc;
  float dz[BN_SMALL_RPT], xh[BN_SMALL_RPT];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < BN_SMALL_RPT; ++i) {
    const int r = rl + 32 * i;
    dz[i] = 0.f;
    xh[i] = 0.f;
    if (r < M) {
      const float xr = x[(long)r * C + c];
      xh[i] = (xr - mu) * is;
      float d = dy[(long)r * C + c];
      if (drop_mask) d = drop_mask[(long)r * C + c] ? d / (1.f - drop_p) : 0.f;
      dz[i] = bn_dz1(d, xh[i], g, b, xr * sc + sh, act);
    }
    s += dz[i];
    q += dz[i] * xh[i];
  }
  rs[rl][cc] = s;
  rq[rl][cc] = q;
  __syncthreads();
  float S = 0.f, Q = 0.f;
#pragma unroll 8
  for (int j = 0; j < 32; ++j) { S += rs[j][cc]; Q += rq[j][cc]; }
  if (rl == 0 && dgamma) {
    dgamma[c] = accumulate ? dgamma[c] + Q : Q;
    dbeta[c] = accumulate ? dbeta[c] + S : S;
  }
  const float invM = 1.0f / (float)M, mb = S * invM, mg = Q * invM, gi = g * is;
#pragma unroll
  for (int i = 0; i < BN_SMALL_RPT; ++i) {
    const int r = rl + 32 * i;
    if (r < M) {
      const float v = gi * (dz[i] - mb - xh[i] * mg);
      // (x = mu + xh / is exactly as loaded: the sign test reads the stored relu output again, cheap and exact)
      dx[(long)r * C + c] = relu_in ? (x[(long)r * C + c] > 0.f ? v : 0.f) : v;
    }
  }
}
static bool bn_small_ok(int dtype, int M, int training, const void* res, const void* mask) {
  static const bool off = mmsa_disabled("bn_small");
  return !off && dtype == MMSA_F32 && training && M <= BN_SMALL_ROWS && !res && !mask;
}

// Minimum rows a lane walks in the partial-sum kernels. 16 leaves the mid-size layers with 49-392 workgroups for 256 CUs
// (C = 64 at 200704 rows: 392; C = 256 at 12544 rows: 98), yet 4 (4x the chunks) measured SLOWER on the whole step (19.66 vs
// 19.46 ms): every extra chunk is another partial row for the finalize kernels, 114 latency-bound launches per step.
// MMSA_BN_RPL overrides (A/B hook).
static int bn_rows_per_lane() {
  static const int v = [] { const char* e = MMSA_EXP_ENV("MMSA_BN_RPL"); const int x = e ? atoi(e) : 0; return x > 0 ? x : 16; }();
  return v;
}

size_t bn_ws_bytes(int C) { return ((size_t)BN_CHUNKS * 2 * C + 2 * C) * sizeof(float); }

template <typename T>
static int bn_forward_t(const T* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                        float* mean, float* invstd, const T* res, T* y, float* ws, int M, int C, float eps, float momentum,
                        int act, int training, hipStream_t st, unsigned char* mask, const float* pre_part, int pre_rows) {
  const BnMap m = bn_map(C);
  if (training && pre_part && pre_rows > 0) {  // the producer of x already summed its columns slice by slice
    launch_stats_finalize(pre_part, pre_rows, M, C, eps, momentum, mean, invstd, running_mean, running_var, st);
  } else if (training) {
    const int rpc = max(cdiv(M, BN_CHUNKS), m.rlanes * bn_rows_per_lane());
    const int chunks = cdiv(M, rpc);
    hipLaunchKernelGGL(bn_stats_partial_kernel<T>, dim3(chunks, m.cgroups), dim3(256), 0, st, x, ws, M, C, m.cthreads, rpc);
    launch_stats_finalize((const float*)ws, chunks, M, C, eps, momentum, mean, invstd, running_mean, running_var, st);
  } else {
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, (const float*)running_mean,
                       (const float*)running_var, eps, C, mean, invstd);
  }
  const int gx = (int)min((long)cdiv(M, m.rlanes), 4096L);
  hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(gx, m.cgroups), dim3(256), 0, st, x, (const float*)mean,
                     (const float*)invstd, gamma, beta, res, y, M, C, m.cthreads, act, mask);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

int bn_forward(int dtype, const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
               float* mean, float* invstd, const void* res, void* y, float* ws, int M, int C, float eps, float momentum,
               int act, int training, hipStream_t st, unsigned char* relu_mask, const float* pre_part, int pre_rows) {
  if (C % 8 || M <= 0) return MMSA_ERR_ARG;
  if (!training && (!running_mean || !running_var)) return MMSA_ERR_ARG;
  if (relu_mask && act != MMSA_ACT_RELU) return MMSA_ERR_ARG;
  if (bn_small_ok(dtype, M, training, res, relu_mask)) {
    hipLaunchKernelGGL(bn_small_fwd_kernel, dim3(C / 8), dim3(256), 0, st, (const float*)x, gamma, beta, running_mean,
                       running_var, mean, invstd, (float*)y, M, C, eps, momentum, act);
    MMSA_CHECK_LAUNCH();
    return MMSA_OK;
  }
  if (dtype == MMSA_BF16)
    return bn_forward_t<bf16>((const bf16*)x, gamma, beta, running_mean, running_var, mean, invstd, (const bf16*)res,
                              (bf16*)y, ws, M, C, eps, momentum, act, training, st, relu_mask, pre_part, pre_rows);
  return bn_forward_t<float>((const float*)x, gamma, beta, running_mean, running_var, mean, invstd, (const float*)res,
                             (float*)y, ws, M, C, eps, momentum, act, training, st, relu_mask, pre_part, pre_rows);
}

template <typename T>
static int bn_backward_t(const T* dy, const T* x, const T* y, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, T* dx, T* dres, float* dgamma, float* dbeta, int accumulate, float* ws, int M,
                         int C, int act, int training, hipStream_t st, const unsigned char* mask) {
  const BnMap m = bn_map(C);
  const int rpc = max(cdiv(M, BN_CHUNKS), m.rlanes * bn_rows_per_lane());
  const int chunks = cdiv(M, rpc);
  float* sums = ws + (size_t)BN_CHUNKS * 2 * C;
  hipLaunchKernelGGL(bn_bwd_partial_kernel<T>, dim3(chunks, m.cgroups), dim3(256), 0, st, dy, x, y, mean, invstd, gamma,
                     beta, ws, M, C, m.cthreads, rpc, act, mask);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, BN_FIN_COLS)), dim3(256), 0, st, (const float*)ws, chunks, C, sums,
                     dgamma, dbeta, accumulate);
  const int gx = (int)min((long)cdiv(M, m.rlanes), 4096L);
  hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(gx, m.cgroups), dim3(256), 0, st, dy, x, y, mean, invstd, gamma, beta,
                     (const float*)sums, dx, dres, M, C, m.cthreads, act, training, mask);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

int bn_backward(int dtype, const void* dy, const void* x, const void* y, const float* mean, const float* invstd,
                const float* gamma, const float* beta, void* dx, void* dres, float* dgamma, float* dbeta, int accumulate,
                float* ws, int M, int C, int act, int training, hipStream_t st, const unsigned char* relu_mask) {
  if (C % 8 || M <= 0) return MMSA_ERR_ARG;
  if (relu_mask && act != MMSA_ACT_RELU) return MMSA_ERR_ARG;
  // (a saved output with ReLU means the forward may have added a residual: the streamed kernels read the sign from it)
  if (bn_small_ok(dtype, M, training, dres, relu_mask) && !(act == MMSA_ACT_RELU && y)) {
    hipLaunchKernelGGL(bn_small_bwd_kernel, dim3(C / 8), dim3(256), 0, st, (const float*)dy, (const float*)x, mean, invstd,
                       gamma, beta, (float*)dx, dgamma, dbeta, accumulate, M, C, act, (const unsigned char*)nullptr, 0.f, 0);
    MMSA_CHECK_LAUNCH();
    return MMSA_OK;
  }
  if (dtype == MMSA_BF16)
    return bn_backward_t<bf16>((const bf16*)dy, (const bf16*)x, (const bf16*)y, mean, invstd, gamma, beta, (bf16*)dx,
                               (bf16*)dres, dgamma, dbeta, accumulate, ws, M, C, act, training, st, relu_mask);
  return bn_backward_t<float>((const float*)dy, (const float*)x, (const float*)y, mean, invstd, gamma, beta, (float*)dx,
                              (float*)dres, dgamma, dbeta, accumulate, ws, M, C, act, training, st, relu_mask);
}

// The fusion head's unit backward up to its Linear: [Dropout backward ->] BatchNorm backward [-> ReLU backward] in one launch
// (fp32, <= BN_SMALL_ROWS rows, training statistics); MMSA_ERR_UNSUPPORTED otherwise (the caller launches the separate kernels).
int bn_small_backward_unit(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, float* dx, float* dgamma, float* dbeta, int accumulate, int M, int C, int act,
                           int training, const unsigned char* drop_mask, float drop_p, int relu_in, hipStream_t st) {
  if (C % 8 || M <= 0) return MMSA_ERR_ARG;
  if (!bn_small_ok(MMSA_F32, M, training, nullptr, nullptr) || act == MMSA_ACT_RELU) return MMSA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(bn_small_bwd_kernel, dim3(C / 8), dim3(256), 0, st, dy, x, mean, invstd, gamma, beta, dx, dgamma, dbeta,
                     accumulate, M, C, act, drop_mask, drop_p, relu_in);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
