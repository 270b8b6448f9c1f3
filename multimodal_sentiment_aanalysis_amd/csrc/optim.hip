// Step tail of the reference trainer as two kernels over the flat parameter / gradient buffers:
//   torch.nn.utils.clip_grad_norm_(params, 1.0)  (Trainer.py:80)  -> grad_sumsq (two-stage, order-fixed)
//   optim.AdamW(lr 1e-4, weight_decay 0.01).step() (Trainer.py:19-21,81) -> adamw_step (also refreshes the
//   bf16 working copy of the weights in the same pass, so no separate cast is needed).
// HBM-bound: 16 B/param read (w, g, m, v) + 12-14 B/param written. The clip coefficient is computed on the
// device from the norm (no host sync); `grad_scale` folds the 1/world_size of the data-parallel average in.
#include "common.h"
#include "ops.h"

#define SUMSQ_BLOCKS 1024

__global__ __launch_bounds__(256) void grad_sumsq_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ part) {
  __shared__ double red[4];
  double acc = 0.0;
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = *(const f32x4*)(g + i * 4);
    acc += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; acc += (double)v * v; }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// norm_out[0] = grad_scale * sqrt(sum); norm_out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)), or -1 = "skip
// this step" when the norm (or the optional device scalar *loss) is not finite: Trainer.py:74-76 leaves the weights alone
// on a NaN loss. *step_count (optional) counts the steps that are applied: the AdamW bias correction must not advance on
// a skipped batch, and the host never learns which ones were skipped (no sync).
__global__ __launch_bounds__(256) void grad_norm_finalize_kernel(const double* __restrict__ part, int nblk, float grad_scale,
                                                                 float max_norm, const float* __restrict__ loss,
                                                                 int* __restrict__ step_count, float* __restrict__ norm_out,
                                                                 float b1, float b2) {
  // one block, fixed summation tree (a single thread walking the 1024 partials took 57 us)
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const float norm = (float)(sqrt(red[0]) * (double)grad_scale);
  norm_out[0] = norm;
  const bool ok = isfinite(norm) && (!loss || isfinite(*loss));
  norm_out[1] = !ok ? -1.f : (max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f);
  if (step_count) {
    // the AdamW bias corrections of the step this norm belongs to, computed ONCE here (norm_out[2], [3]): as two powf per
    // thread of the 1 M-thread AdamW kernel they cost 0.2 ms per step (adamw_kernel 630 -> 846 us)
    const int t = *step_count + (ok ? 1 : 0);
    if (ok) *step_count = t;
    norm_out[2] = 1.f - powf(b1, (float)max(t, 1));
    norm_out[3] = sqrtf(1.f - powf(b2, (float)max(t, 1)));
  }
}

size_t grad_norm_ws_bytes() { return 2 * SUMSQ_BLOCKS * sizeof(double); }

// the same norm over several disjoint ranges of one gradient buffer (the trainable sub-ranges of a curriculum phase,
// dataLoader/MultiTaskTrainer.py:50-177): each range gets a share of the partial blocks, one finalize over all of them
int grad_norm_ranges(const float* g, const long* offs, const long* lens, int nr, float grad_scale, float max_norm,
                     float* norm_out, void* ws, hipStream_t st, const float* loss, int* step_count, float b1, float b2) {
  if (nr < 1 || nr > 256) return MMSA_ERR_ARG;
  long total = 0;
  for (int r = 0; r < nr; ++r) {
    if (lens[r] <= 0 || offs[r] < 0) return MMSA_ERR_ARG;
    total += lens[r];
  }
  int used = 0;
  for (int r = 0; r < nr; ++r) {
    long want = (long)((double)SUMSQ_BLOCKS * (double)lens[r] / (double)total);
    const long cap = (lens[r] / 4 + 255) / 256 + 1;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    if (used + want > 2 * SUMSQ_BLOCKS) return MMSA_ERR_ARG;
    hipLaunchKernelGGL(grad_sumsq_partial_kernel, dim3((int)want), dim3(256), 0, st, g + offs[r], lens[r], (double*)ws + used);
    used += (int)want;
  }
  hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, used, grad_scale, max_norm,
                     loss, step_count, norm_out, b1, b2);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// The two halves of the norm as separate calls (round 4). (a) sum of squares over ranges, left as ONE double on the device:
// what a rank of the reduce-scatter step contributes for the gradient shards it owns (the doubles of all ranks are then summed by a
// one-element all-reduce), or what a gradient range contributes the moment it is announced (its partial sums then run under
// the rest of the backward instead of behind it). (b) the finalize over n such doubles.
__global__ __launch_bounds__(256) void grad_sumsq_reduce_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}
int grad_sumsq_ranges(const float* g, const long* offs, const long* lens, int nr, double* sumsq, void* ws, hipStream_t st) {
  if (nr < 0 || nr > 256) return MMSA_ERR_ARG;
  if (nr == 0) {  // a rank that owns nothing contributes zero
    if (hipMemsetAsync(sumsq, 0, sizeof(double), st) != hipSuccess) return MMSA_ERR_LAUNCH;
    return MMSA_OK;
  }
  long total = 0;
  for (int r = 0; r < nr; ++r) {
    if (lens[r] <= 0 || offs[r] < 0) return MMSA_ERR_ARG;
    total += lens[r];
  }
  int used = 0;
  for (int r = 0; r < nr; ++r) {
    long want = (long)((double)SUMSQ_BLOCKS * (double)lens[r] / (double)total);
    const long cap = (lens[r] / 4 + 255) / 256 + 1;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    if (used + want > 2 * SUMSQ_BLOCKS) return MMSA_ERR_ARG;
    hipLaunchKernelGGL(grad_sumsq_partial_kernel, dim3((int)want), dim3(256), 0, st, g + offs[r], lens[r], (double*)ws + used);
    used += (int)want;
  }
  hipLaunchKernelGGL(grad_sumsq_reduce_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, used, sumsq);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
int grad_norm_from_sumsq(const double* sumsq, int n, float grad_scale, float max_norm, float* norm_out, hipStream_t st,
                         const float* loss, int* step_count, float b1, float b2) {
  if (n < 1 || n > 2 * SUMSQ_BLOCKS) return MMSA_ERR_ARG;
  hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(256), 0, st, sumsq, n, grad_scale, max_norm, loss, step_count,
                     norm_out, b1, b2);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

int grad_norm(const float* g, long n, float grad_scale, float max_norm, float* norm_out, void* ws, hipStream_t st,
              const float* loss, int* step_count, float b1, float b2) {
  const int blocks = (int)min((n / 4 + 255) / 256 + 1, (long)SUMSQ_BLOCKS);
  hipLaunchKernelGGL(grad_sumsq_partial_kernel, dim3(blocks), dim3(256), 0, st, g, n, (double*)ws);
  hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, blocks, grad_scale, max_norm,
                     loss, step_count, norm_out, b1, b2);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// g *= clip coefficient (norm_clip[1]; nothing when the step is skipped): torch's clip_grad_norm_ scales EVERY gradient it was
// given in place, also those no optimizer owns — the reference's phase 3 (MultiTaskTrainer.py:147-177) clips four modules'
// gradients but steps (and zeroes) only the valence head's, so the others keep accumulating the scaled values.
__global__ __launch_bounds__(256) void grad_scale_clip_kernel(float* __restrict__ g, long n, const float* __restrict__ norm_clip) {
  const float c = norm_clip[1];
  if (c < 0.f || c == 1.f) return;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) g[i] *= c;
}
int grad_scale_clip(float* g, long n, const float* norm_clip, hipStream_t st) {
  if (n <= 0) return MMSA_ERR_ARG;
  const int blocks = (int)min((n + 255) / 256, 2048L);
  hipLaunchKernelGGL(grad_scale_clip_kernel, dim3(blocks), dim3(256), 0, st, g, n, norm_clip);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// torch.optim.AdamW (decoupled weight decay on every parameter, bias-corrected):
//   w *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; w -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16* __restrict__ w16, long n, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    const float* __restrict__ norm_clip, float grad_scale,
                                                    const int* __restrict__ step_count) {
  const float clip = norm_clip ? norm_clip[1] : 1.f;
  if (clip < 0.f) return;  // non-finite gradient norm / loss: the step is skipped, w, m, v and the bf16 copy stay as they are
  if (step_count) {        // bias corrections of the device-side count of applied steps: written by the norm's finalize kernel
    bc1 = norm_clip[2];
    bc2_sqrt = norm_clip[3];
  }
  const float coef = clip * grad_scale;
  // loop invariants, explicitly: with the corrections coming from memory the compiler no longer treats lr / bc1 as a uniform
  // scalar and redid both divisions per element (770 us instead of 630 for 135.6 M parameters)
  const float step_size = lr / bc1, inv_bc2 = 1.f / bc2_sqrt, decay = 1.f - lr * wd, omb1 = 1.f - b1, omb2 = 1.f - b2;
  const long n4 = n / 4;
  // every stream is touched once per step (540 MB each: nothing survives in L2 / MALL until its next use), so all
  // fp32 accesses are non-temporal; two independent float4 groups per thread keep 8 loads in flight
  auto upd = [&](f32x4& wv, f32x4& mv, f32x4& vv, f32x4 gv) __attribute__((always_inline)) {
    gv = gv * coef;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      wv[e] *= decay;
      mv[e] = b1 * mv[e] + omb1 * gv[e];
      vv[e] = b2 * vv[e] + omb2 * gv[e] * gv[e];
      wv[e] -= step_size * mv[e] / (sqrtf(vv[e]) * inv_bc2 + eps);
    }
  };
  auto ld = [&](const float* q, long i) __attribute__((always_inline)) -> f32x4 { return __builtin_nontemporal_load((const f32x4*)(q + i * 4)); };
  auto st = [&](float* q, long i, f32x4 x) __attribute__((always_inline)) { __builtin_nontemporal_store(x, (f32x4*)(q + i * 4)); };
  auto st16 = [&](long i, f32x4 wv) __attribute__((always_inline)) {
    if (w16) {
      bf16x4 t = {(bf16)wv[0], (bf16)wv[1], (bf16)wv[2], (bf16)wv[3]};
      *(bf16x4*)(w16 + i * 4) = t;  // read again by the next forward: regular store
    }
  };
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const long j = i + stride;
    f32x4 w0 = ld(w, i), m0 = ld(m, i), v0 = ld(v, i), g0 = ld(g, i);
    f32x4 w1 = ld(w, j), m1 = ld(m, j), v1 = ld(v, j), g1 = ld(g, j);
    upd(w0, m0, v0, g0);
    upd(w1, m1, v1, g1);
    st(w, i, w0); st(m, i, m0); st(v, i, v0); st16(i, w0);
    st(w, j, w1); st(m, j, m1); st(v, j, v1); st16(j, w1);
  }
  for (; i < n4; i += stride) {
    f32x4 w0 = ld(w, i), m0 = ld(m, i), v0 = ld(v, i), g0 = ld(g, i);
    upd(w0, m0, v0, g0);
    st(w, i, w0); st(m, i, m0); st(v, i, v0); st16(i, w0);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = n4 * 4 + threadIdx.x;
    float wv = w[i] * decay;
    const float gv = g[i] * coef;
    const float mv = b1 * m[i] + omb1 * gv, vv = b2 * v[i] + omb2 * gv * gv;
    wv -= step_size * mv / (sqrtf(vv) * inv_bc2 + eps);
    w[i] = wv; m[i] = mv; v[i] = vv;
    if (w16) w16[i] = (bf16)wv;
  }
}

int adamw_step(float* w, const float* g, float* m, float* v, void* w16, long n, float lr, float b1, float b2, float eps,
               float wd, int step, const float* norm_clip, float grad_scale, hipStream_t st, const int* step_count) {
  if (n <= 0 || (step < 1 && !step_count)) return MMSA_ERR_ARG;
  const float bc1 = 1.f - powf(b1, (float)(step < 1 ? 1 : step));
  const float bc2_sqrt = sqrtf(1.f - powf(b2, (float)(step < 1 ? 1 : step)));
  const int blocks = (int)min((n / 4 + 255) / 256 + 1, 4096L);
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, st, w, g, m, v, (bf16*)w16, n, lr, b1, b2, eps, wd, bc1,
                     bc2_sqrt, norm_clip, grad_scale, step_count);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
