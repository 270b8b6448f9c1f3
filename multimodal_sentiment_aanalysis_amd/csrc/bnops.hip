// BatchNorm over the rows of an [M][C] matrix (NHWC activations: M = B*H*W; fusion head: M = B), training
// statistics, fused apply (+ residual, + ReLU/GELU) and backward. HBM-bound: every kernel streams rows with
// 4-element vector accesses; a block's 256 threads are laid out as (C/4 column threads) x (row lanes) so that
// even C = 64 keeps all lanes busy on fully contiguous rows. Reductions are two-stage and order-fixed
// (per-chunk partials -> finalize in double), hence bitwise reproducible.
//
// Semantics follow nn.BatchNorm1d/2d as the reference uses them (MultimodalModel.py:181,186,194,380,419,423;
// eps 1e-5, momentum 0.1): normalise with the biased batch variance, update running_var with the unbiased one.
#include "common.h"
#include "gemm_epilogue.h"
#include "ops.h"

#define BN_CHUNKS 1024

struct BnMap {
  int cthreads, rlanes, cgroups;  // column threads per block, row lanes per block, grid.y
};
static BnMap bn_map(int C) {
  const int c4 = C / 4;
  int ct = 1;
  while (ct < c4 && ct < 256) ct <<= 1;
  BnMap m;
  m.cthreads = ct;
  m.rlanes = 256 / ct;
  m.cgroups = cdiv(c4, ct);
  return m;
}

// part[chunk][2][C]: sum x, sum x^2 over the chunk's rows
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const T* __restrict__ x, float* __restrict__ part, int M,
                                                               int C, int cthreads, int rows_per_chunk) {
  __shared__ f32x4 red[2][256];
  const int ct = threadIdx.x % cthreads, rl = threadIdx.x / cthreads, rlanes = 256 / cthreads;
  const int col = (blockIdx.y * cthreads + ct) * 4;
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  if (col < C)
    for (int r = r0 + rl; r < r1; r += rlanes) {
      const f32x4 v = Vec4<T>::load(x + (long)r * C + col);
      s += v;
      q += v * v;
    }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  if (rl == 0 && col < C) {
    for (int j = 1; j < rlanes; ++j) {
      s += red[0][j * cthreads + ct];
      q += red[1][j * cthreads + ct];
    }
    *(f32x4*)(part + ((long)blockIdx.x * 2 + 0) * C + col) = s;
    *(f32x4*)(part + ((long)blockIdx.x * 2 + 1) * C + col) = q;
  }
}

// finalize kernels: 256 threads = 16 columns x 16 partial-lanes (fixed summation tree, see partial_finalize_kernel)
#define BN_FIN_COLS 4
#define BN_FIN_LANES 64
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ part, int chunks, int M, int C,
                                                                float eps, float momentum, float* __restrict__ mean,
                                                                float* __restrict__ invstd, float* __restrict__ running_mean,
                                                                float* __restrict__ running_var) {
  __shared__ double rs[BN_FIN_LANES][BN_FIN_COLS + 1], rq[BN_FIN_LANES][BN_FIN_COLS + 1];
  const int cc = threadIdx.x & (BN_FIN_COLS - 1), r = threadIdx.x / BN_FIN_COLS;
  const int c = blockIdx.x * BN_FIN_COLS + cc;
  double s = 0, q = 0;
  if (c < C)
    for (int b = r; b < chunks; b += BN_FIN_LANES) {
      s += part[((long)b * 2 + 0) * C + c];
      q += part[((long)b * 2 + 1) * C + c];
    }
  rs[r][cc] = s;
  rq[r][cc] = q;
  __syncthreads();
  if (r != 0 || c >= C) return;
  s = 0; q = 0;
#pragma unroll
  for (int k = 0; k < BN_FIN_LANES; ++k) { s += rs[k][cc]; q += rq[k][cc]; }
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

// eval mode: statistics come from the running buffers
__global__ void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                     float eps, int C, float* __restrict__ mean, float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = running_mean[c];
  invstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// y = act( (x - mean) * invstd * gamma + beta (+ res) )
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const T* __restrict__ res,
                                                       T* __restrict__ y, int M, int C, int cthreads, int act) {
  const int ct = threadIdx.x % cthreads, rl = threadIdx.x / cthreads, rlanes = 256 / cthreads;
  const int col = (blockIdx.y * cthreads + ct) * 4;
  if (col >= C) return;
  const f32x4 mu = *(const f32x4*)(mean + col), is = *(const f32x4*)(invstd + col);
  const f32x4 sc = is * *(const f32x4*)(gamma + col);
  const f32x4 sh = *(const f32x4*)(beta + col) - mu * sc;
  for (long r = (long)blockIdx.x * rlanes + rl; r < M; r += (long)gridDim.x * rlanes) {
    f32x4 v = Vec4<T>::load(x + r * C + col) * sc + sh;
    if (res) v += Vec4<T>::load(res + r * C + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
    Vec4<T>::store(y + r * C + col, v);
  }
}

// backward pass 1: dz = dy * act'(z); partial sums of dz and dz * xhat.  ReLU mask comes from the saved output y
// (y > 0 <=> z > 0, also with a residual) or, when y == null (no residual), is recomputed from x with the forward's own
// expression x*sc + sh, which saves one of the three streamed reads; GELU (no residual in the reference) recomputes z.
template <typename T>
__device__ __forceinline__ f32x4 bn_dz(f32x4 dy, f32x4 xh, f32x4 g, f32x4 b, f32x4 yv, int act) {
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

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             const T* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ part, int M, int C, int cthreads,
                                                             int rows_per_chunk, int act) {
  __shared__ f32x4 red[2][256];
  const int ct = threadIdx.x % cthreads, rl = threadIdx.x / cthreads, rlanes = 256 / cthreads;
  const int col = (blockIdx.y * cthreads + ct) * 4;
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  if (col < C) {
    const f32x4 mu = *(const f32x4*)(mean + col), is = *(const f32x4*)(invstd + col);
    const f32x4 g = *(const f32x4*)(gamma + col), b = *(const f32x4*)(beta + col);
    const f32x4 sc = is * g, sh = b - mu * sc;  // as bn_apply_kernel: the ReLU mask without a saved output (y == null)
    for (int r = r0 + rl; r < r1; r += rlanes) {
      const f32x4 xr = Vec4<T>::load(x + (long)r * C + col);
      const f32x4 xh = (xr - mu) * is;
      f32x4 yv = {0, 0, 0, 0};
      if (act == MMSA_ACT_RELU) yv = y ? Vec4<T>::load(y + (long)r * C + col) : xr * sc + sh;
      const f32x4 dz = bn_dz<T>(Vec4<T>::load(dy + (long)r * C + col), xh, g, b, yv, act);
      s += dz;
      q += dz * xh;
    }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  if (rl == 0 && col < C) {
    for (int j = 1; j < rlanes; ++j) {
      s += red[0][j * cthreads + ct];
      q += red[1][j * cthreads + ct];
    }
    *(f32x4*)(part + ((long)blockIdx.x * 2 + 0) * C + col) = s;
    *(f32x4*)(part + ((long)blockIdx.x * 2 + 1) * C + col) = q;
  }
}

// sums[0][C] = dbeta, sums[1][C] = dgamma (kept for pass 2) and (+)= into the parameter gradients
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int chunks, int C,
                                                              float* __restrict__ sums, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate) {
  __shared__ float rs[BN_FIN_LANES][BN_FIN_COLS + 1], rq[BN_FIN_LANES][BN_FIN_COLS + 1];
  const int cc = threadIdx.x & (BN_FIN_COLS - 1), r = threadIdx.x / BN_FIN_COLS;
  const int c = blockIdx.x * BN_FIN_COLS + cc;
  float s = 0, q = 0;
  if (c < C)
    for (int b = r; b < chunks; b += BN_FIN_LANES) {
      s += part[((long)b * 2 + 0) * C + c];
      q += part[((long)b * 2 + 1) * C + c];
    }
  rs[r][cc] = s;
  rq[r][cc] = q;
  __syncthreads();
  if (r != 0 || c >= C) return;
  s = 0; q = 0;
#pragma unroll
  for (int k = 0; k < BN_FIN_LANES; ++k) { s += rs[k][cc]; q += rq[k][cc]; }
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
                                                           int training) {
  const int ct = threadIdx.x % cthreads, rl = threadIdx.x / cthreads, rlanes = 256 / cthreads;
  const int col = (blockIdx.y * cthreads + ct) * 4;
  if (col >= C) return;
  const f32x4 mu = *(const f32x4*)(mean + col), is = *(const f32x4*)(invstd + col);
  const f32x4 g = *(const f32x4*)(gamma + col), b = *(const f32x4*)(beta + col);
  f32x4 mb = {0, 0, 0, 0}, mg = {0, 0, 0, 0};
  if (training) {
    const float invM = 1.0f / (float)M;
    mb = *(const f32x4*)(sums + col) * invM;
    mg = *(const f32x4*)(sums + C + col) * invM;
  }
  const f32x4 gi = g * is;
  const f32x4 sc = is * g, sh = b - mu * sc;
  for (long r = (long)blockIdx.x * rlanes + rl; r < M; r += (long)gridDim.x * rlanes) {
    const f32x4 xr = Vec4<T>::load(x + r * C + col);
    const f32x4 xh = (xr - mu) * is;
    f32x4 yv = {0, 0, 0, 0};
    if (act == MMSA_ACT_RELU) yv = y ? Vec4<T>::load(y + r * C + col) : xr * sc + sh;
    const f32x4 dz = bn_dz<T>(Vec4<T>::load(dy + r * C + col), xh, g, b, yv, act);
    if (dres) Vec4<T>::store(dres + r * C + col, dz);
    Vec4<T>::store(dx + r * C + col, gi * (dz - mb - xh * mg));
  }
}

size_t bn_ws_bytes(int C) { return ((size_t)BN_CHUNKS * 2 * C + 2 * C) * sizeof(float); }

template <typename T>
static int bn_forward_t(const T* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                        float* mean, float* invstd, const T* res, T* y, float* ws, int M, int C, float eps, float momentum,
                        int act, int training, hipStream_t st) {
  const BnMap m = bn_map(C);
  if (training) {
    const int rpc = max(cdiv(M, BN_CHUNKS), m.rlanes * 16);
    const int chunks = cdiv(M, rpc);
    hipLaunchKernelGGL(bn_stats_partial_kernel<T>, dim3(chunks, m.cgroups), dim3(256), 0, st, x, ws, M, C, m.cthreads, rpc);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(cdiv(C, BN_FIN_COLS)), dim3(256), 0, st, (const float*)ws, chunks, M, C, eps,
                       momentum, mean, invstd, running_mean, running_var);
  } else {
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, (const float*)running_mean,
                       (const float*)running_var, eps, C, mean, invstd);
  }
  const int gx = (int)min((long)cdiv(M, m.rlanes), 2048L);
  hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(gx, m.cgroups), dim3(256), 0, st, x, (const float*)mean,
                     (const float*)invstd, gamma, beta, res, y, M, C, m.cthreads, act);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

int bn_forward(int dtype, const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
               float* mean, float* invstd, const void* res, void* y, float* ws, int M, int C, float eps, float momentum,
               int act, int training, hipStream_t st) {
  if (C % 4 || M <= 0) return MMSA_ERR_ARG;
  if (!training && (!running_mean || !running_var)) return MMSA_ERR_ARG;
  if (dtype == MMSA_BF16)
    return bn_forward_t<bf16>((const bf16*)x, gamma, beta, running_mean, running_var, mean, invstd, (const bf16*)res,
                              (bf16*)y, ws, M, C, eps, momentum, act, training, st);
  return bn_forward_t<float>((const float*)x, gamma, beta, running_mean, running_var, mean, invstd, (const float*)res,
                             (float*)y, ws, M, C, eps, momentum, act, training, st);
}

template <typename T>
static int bn_backward_t(const T* dy, const T* x, const T* y, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, T* dx, T* dres, float* dgamma, float* dbeta, int accumulate, float* ws, int M,
                         int C, int act, int training, hipStream_t st) {
  const BnMap m = bn_map(C);
  const int rpc = max(cdiv(M, BN_CHUNKS), m.rlanes * 16);
  const int chunks = cdiv(M, rpc);
  float* sums = ws + (size_t)BN_CHUNKS * 2 * C;
  hipLaunchKernelGGL(bn_bwd_partial_kernel<T>, dim3(chunks, m.cgroups), dim3(256), 0, st, dy, x, y, mean, invstd, gamma,
                     beta, ws, M, C, m.cthreads, rpc, act);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, BN_FIN_COLS)), dim3(256), 0, st, (const float*)ws, chunks, C, sums,
                     dgamma, dbeta, accumulate);
  const int gx = (int)min((long)cdiv(M, m.rlanes), 2048L);
  hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(gx, m.cgroups), dim3(256), 0, st, dy, x, y, mean, invstd, gamma, beta,
                     (const float*)sums, dx, dres, M, C, m.cthreads, act, training);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

int bn_backward(int dtype, const void* dy, const void* x, const void* y, const float* mean, const float* invstd,
                const float* gamma, const float* beta, void* dx, void* dres, float* dgamma, float* dbeta, int accumulate,
                float* ws, int M, int C, int act, int training, hipStream_t st) {
  if (C % 4 || M <= 0) return MMSA_ERR_ARG;
  if (dtype == MMSA_BF16)
    return bn_backward_t<bf16>((const bf16*)dy, (const bf16*)x, (const bf16*)y, mean, invstd, gamma, beta, (bf16*)dx,
                               (bf16*)dres, dgamma, dbeta, accumulate, ws, M, C, act, training, st);
  return bn_backward_t<float>((const float*)dy, (const float*)x, (const float*)y, mean, invstd, gamma, beta, (float*)dx,
                              (float*)dres, dgamma, dbeta, accumulate, ws, M, C, act, training, st);
}
