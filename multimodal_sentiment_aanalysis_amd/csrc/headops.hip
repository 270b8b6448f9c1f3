// Small fused kernels of the fusion head (fp32): the modality-sequence attention core, L2 normalisation,
// modality pooling, gate/mix, dynamic-weight softmax + scaled concat, fused cross-entropy forward+backward,
// dropout and a few strided element-wise helpers. Shapes are tiny (B x 256, sequences of 1-4 modality tokens),
// so each kernel keeps a whole row / (batch, head) in registers and reduces with wavefront shuffles.
//
// Reference arithmetic: MML_ZYC/MultimodalModel.py:108-149 (CrossModalTransformer), :374-404 (ME-MHACL fusion),
// :171-176,:298-306 (dynamic weights), Trainer.py:17,68 (CrossEntropyLoss).
#include "common.h"
#include "gemm_epilogue.h"
#include "ops.h"

#define MAXL 4

// ------------------------------------------------------------------------------------------------ L2 normalise
// y = x / max(||x||, eps)   (F.normalize, MultimodalModel.py:388-390); one wave per row
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ nrm, int M, int E, float eps, long ldy) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float s = 0.f;
  for (int e = lane; e < E; e += 64) { const float v = x[(long)row * E + e]; s += v * v; }
  const float n = fmaxf(sqrtf(wave_sum(s)), eps);
  if (lane == 0) nrm[row] = n;
  for (int e = lane; e < E; e += 64) y[(long)row * ldy + e] = x[(long)row * E + e] / n;
}
// dx = (dy - y (y . dy)) / n   (for ||x|| > eps; below eps the map is linear: dx = dy / eps)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ nrm, float* __restrict__ dx, int M, int E,
                                                         float eps, int accumulate, long ldin) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float s = 0.f;
  for (int e = lane; e < E; e += 64) s += y[(long)row * ldin + e] * dy[(long)row * ldin + e];
  s = wave_sum(s);
  const float n = nrm[row];
  if (n <= eps) s = 0.f;
  for (int e = lane; e < E; e += 64) {
    const float g = (dy[(long)row * ldin + e] - y[(long)row * ldin + e] * s) / n;
    float* d = dx + (long)row * E + e;
    *d = accumulate ? *d + g : g;
  }
}
// ldy / ldin: row stride of the normalised rows (and of their gradient) - the A2 head keeps them interleaved as tokens of
// one sequence [B][modalities][E]; 0 means dense (E)
int l2norm_fwd_ld(const float* x, float* y, long ldy, float* nrm, int M, int E, float eps, hipStream_t st) {
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, x, y, nrm, M, E, eps, ldy ? ldy : (long)E);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
int l2norm_bwd_ld(const float* dy, const float* y, long ldin, const float* nrm, float* dx, int M, int E, float eps,
                  int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, dy, y, nrm, dx, M, E, eps, accumulate,
                     ldin ? ldin : (long)E);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
int l2norm_fwd(const float* x, float* y, float* nrm, int M, int E, float eps, hipStream_t st) {
  return l2norm_fwd_ld(x, y, 0, nrm, M, E, eps, st);
}
int l2norm_bwd(const float* dy, const float* y, const float* nrm, float* dx, int M, int E, float eps, int accumulate,
               hipStream_t st) {
  return l2norm_bwd_ld(dy, y, 0, nrm, dx, M, E, eps, accumulate, st);
}

// ------------------------------------------------------------------------------------------------ MHA core
// One workgroup per batch element, one thread per embedding dim e (head = e / hd). The QK^T dot products are
// reduced across the hd lanes of a head with __shfl_xor (hd in {16,32,64} divides the 64-lane wavefront), the
// Lq x Lk scores, the softmax and its backward live in registers. q/k/v are row-strided views: row (b*L + l).
__device__ __forceinline__ float head_sum(float v, int hd) {
  for (int o = hd >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ void mha_core_fwd_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ k, long ldk,
                                    const float* __restrict__ v, long ldv, float* __restrict__ ctx, long ldc,
                                    float* __restrict__ probs, int Lq, int Lk, int E, int heads) {
  const int b = blockIdx.x, e = threadIdx.x, hd = E / heads, h = e / hd;
  const float scale = rsqrtf((float)hd);
  float qv[MAXL], kv[MAXL], vv[MAXL];
  for (int l = 0; l < Lq; ++l) qv[l] = q[((long)b * Lq + l) * ldq + e] * scale;  // torch scales q before QK^T
  for (int l = 0; l < Lk; ++l) { kv[l] = k[((long)b * Lk + l) * ldk + e]; vv[l] = v[((long)b * Lk + l) * ldv + e]; }
  for (int i = 0; i < Lq; ++i) {
    float s[MAXL], mx = -INFINITY;
    for (int j = 0; j < Lk; ++j) { s[j] = head_sum(qv[i] * kv[j], hd); mx = fmaxf(mx, s[j]); }
    float sum = 0.f;
    for (int j = 0; j < Lk; ++j) { s[j] = __expf(s[j] - mx); sum += s[j]; }
    float o = 0.f;
    for (int j = 0; j < Lk; ++j) {
      const float p = s[j] / sum;
      o = fmaf(p, vv[j], o);
      if ((e % hd) == 0) probs[(((long)b * heads + h) * Lq + i) * Lk + j] = p;
    }
    ctx[((long)b * Lq + i) * ldc + e] = o;
  }
}

__global__ void mha_core_bwd_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ k, long ldk,
                                    const float* __restrict__ v, long ldv, const float* __restrict__ probs,
                                    const float* __restrict__ dctx, long ldc, float* __restrict__ dq, long lddq,
                                    float* __restrict__ dk, long lddk, float* __restrict__ dv, long lddv, int Lq, int Lk,
                                    int E, int heads) {
  const int b = blockIdx.x, e = threadIdx.x, hd = E / heads, h = e / hd;
  const float scale = rsqrtf((float)hd);
  float qv[MAXL], kv[MAXL], vv[MAXL], dkv[MAXL], dvv[MAXL];
  for (int l = 0; l < Lq; ++l) qv[l] = q[((long)b * Lq + l) * ldq + e];
  for (int l = 0; l < Lk; ++l) {
    kv[l] = k[((long)b * Lk + l) * ldk + e];
    vv[l] = v[((long)b * Lk + l) * ldv + e];
    dkv[l] = 0.f; dvv[l] = 0.f;
  }
  for (int i = 0; i < Lq; ++i) {
    const float dc = dctx[((long)b * Lq + i) * ldc + e];
    float p[MAXL], dp[MAXL], delta = 0.f;
    for (int j = 0; j < Lk; ++j) {
      p[j] = probs[(((long)b * heads + h) * Lq + i) * Lk + j];
      dp[j] = head_sum(dc * vv[j], hd);
      delta += p[j] * dp[j];
    }
    float dqi = 0.f;
    for (int j = 0; j < Lk; ++j) {
      const float ds = p[j] * (dp[j] - delta) * scale;
      dqi = fmaf(ds, kv[j], dqi);
      dkv[j] = fmaf(ds, qv[i], dkv[j]);
      dvv[j] = fmaf(p[j], dc, dvv[j]);
    }
    dq[((long)b * Lq + i) * lddq + e] = dqi;
  }
  for (int j = 0; j < Lk; ++j) {
    dk[((long)b * Lk + j) * lddk + e] = dkv[j];
    dv[((long)b * Lk + j) * lddv + e] = dvv[j];
  }
}

static bool mha_shape_ok(int Lq, int Lk, int E, int heads) {
  if (Lq < 1 || Lk < 1 || Lq > MAXL || Lk > MAXL || heads < 1 || E % heads || E > 1024 || E % 64) return false;
  const int hd = E / heads;
  return hd == 16 || hd == 32 || hd == 64;
}
int mha_core_fwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* ctx, long ldc,
                 float* probs, int B, int Lq, int Lk, int E, int heads, hipStream_t st) {
  if (!mha_shape_ok(Lq, Lk, E, heads)) return MMSA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(mha_core_fwd_kernel, dim3(B), dim3(E), 0, st, q, ldq, k, ldk, v, ldv, ctx, ldc, probs, Lq, Lk, E, heads);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
int mha_core_bwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, const float* probs,
                 const float* dctx, long ldc, float* dq, long lddq, float* dk, long lddk, float* dv, long lddv, int B,
                 int Lq, int Lk, int E, int heads, hipStream_t st) {
  if (!mha_shape_ok(Lq, Lk, E, heads)) return MMSA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(mha_core_bwd_kernel, dim3(B), dim3(E), 0, st, q, ldq, k, ldk, v, ldv, probs, dctx, ldc, dq, lddq, dk,
                     lddk, dv, lddv, Lq, Lk, E, heads);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// ------------------------------------------------------------------------------------------------ modality pooling
// x [B][L][E] -> y [B][E]; mode 0 = max with argmax (MultimodalModel.py:401), 1 = mean (ME-MHACL/model.py:73)
__global__ void seq_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx,
                                    int B, int L, int E, int mode) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * E) return;
  const int b = (int)(i / E), e = (int)(i % E);
  if (mode == 0) {
    float best = x[((long)b * L) * E + e];
    int bi = 0;
    for (int l = 1; l < L; ++l) {
      const float v = x[((long)b * L + l) * E + e];
      if (v > best) { best = v; bi = l; }
    }
    y[i] = best;
    idx[i] = (unsigned char)bi;
  } else {
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += x[((long)b * L + l) * E + e];
    y[i] = s / (float)L;
  }
}
__global__ void seq_pool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx,
                                    float* __restrict__ dx, int B, int L, int E, int mode) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * E) return;
  const int b = (int)(i / E), e = (int)(i % E);
  for (int l = 0; l < L; ++l)
    dx[((long)b * L + l) * E + e] = mode == 0 ? (idx[i] == l ? dy[i] : 0.f) : dy[i] / (float)L;
}
int seq_pool_fwd(const float* x, float* y, unsigned char* idx, int B, int L, int E, int mode, hipStream_t st) {
  hipLaunchKernelGGL(seq_pool_fwd_kernel, dim3(cdiv((long)B * E, 256)), dim3(256), 0, st, x, y, idx, B, L, E, mode);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
int seq_pool_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int L, int E, int mode, hipStream_t st) {
  hipLaunchKernelGGL(seq_pool_bwd_kernel, dim3(cdiv((long)B * E, 256)), dim3(256), 0, st, dy, idx, dx, B, L, E, mode);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// ------------------------------------------------------------------------------------------------ gate mix
// mix = g*q + (1-g)*a   (MultimodalModel.py:148; g is the sigmoid output of the gate Linear, LayerNorm follows)
__global__ void gate_mix_fwd_kernel(const float* __restrict__ g, const float* __restrict__ q, long ldq,
                                    const float* __restrict__ a, long lda, float* __restrict__ mix, int B, int E) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * E) return;
  const int b = (int)(i / E), e = (int)(i % E);
  const float gv = g[i], qv = q[(long)b * ldq + e], av = a[(long)b * lda + e];
  mix[i] = gv * qv + (1.f - gv) * av;
}
// dq = dmix*g ; da = dmix*(1-g) ; dgpre = dmix*(q-a)*g*(1-g)   (sigmoid folded in)
__global__ void gate_mix_bwd_kernel(const float* __restrict__ dmix, const float* __restrict__ g, const float* __restrict__ q,
                                    long ldq, const float* __restrict__ a, long lda, float* __restrict__ dq,
                                    float* __restrict__ da, float* __restrict__ dgpre, int B, int E) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * E) return;
  const int b = (int)(i / E), e = (int)(i % E);
  const float gv = g[i], d = dmix[i];
  dq[i] = d * gv;
  da[i] = d * (1.f - gv);
  dgpre[i] = d * (q[(long)b * ldq + e] - a[(long)b * lda + e]) * gv * (1.f - gv);
}
int gate_mix_fwd(const float* g, const float* q, long ldq, const float* a, long lda, float* mix, int B, int E, hipStream_t st) {
  hipLaunchKernelGGL(gate_mix_fwd_kernel, dim3(cdiv((long)B * E, 256)), dim3(256), 0, st, g, q, ldq, a, lda, mix, B, E);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
int gate_mix_bwd(const float* dmix, const float* g, const float* q, long ldq, const float* a, long lda, float* dq, float* da,
                 float* dgpre, int B, int E, hipStream_t st) {
  hipLaunchKernelGGL(gate_mix_bwd_kernel, dim3(cdiv((long)B * E, 256)), dim3(256), 0, st, dmix, g, q, ldq, a, lda, dq, da,
                     dgpre, B, E);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// ------------------------------------------------------------------------------------------------ dynamic weights
// w = softmax(wl[b][0:3]); out[b] = [f1*w0 | f2*w1 | f3*w2]   (MultimodalModel.py:175, 302-306); one wave per row
__global__ __launch_bounds__(256) void weighted_concat_fwd_kernel(const float* __restrict__ wl, const float* __restrict__ f1,
                                                                  const float* __restrict__ f2, const float* __restrict__ f3,
                                                                  float* __restrict__ w, float* __restrict__ out, int B,
                                                                  int E) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float a0 = wl[row * 3 + 0], a1 = wl[row * 3 + 1], a2 = wl[row * 3 + 2];
  const float mx = fmaxf(a0, fmaxf(a1, a2));
  const float e0 = __expf(a0 - mx), e1 = __expf(a1 - mx), e2 = __expf(a2 - mx);
  const float inv = 1.f / (e0 + e1 + e2);
  const float w0 = e0 * inv, w1 = e1 * inv, w2 = e2 * inv;
  if (lane == 0) { w[row * 3 + 0] = w0; w[row * 3 + 1] = w1; w[row * 3 + 2] = w2; }
  float* o = out + (long)row * 3 * E;
  for (int e = lane; e < E; e += 64) {
    o[e] = f1[(long)row * E + e] * w0;
    o[E + e] = f2[(long)row * E + e] * w1;
    o[2 * E + e] = f3[(long)row * E + e] * w2;
  }
}
// df_i = dout_i * w_i ; dwl = w * (dw - sum_j w_j dw_j), dw_i = <dout_i, f_i>
__global__ __launch_bounds__(256) void weighted_concat_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                                  const float* __restrict__ f1, const float* __restrict__ f2,
                                                                  const float* __restrict__ f3, float* __restrict__ df1,
                                                                  float* __restrict__ df2, float* __restrict__ df3,
                                                                  float* __restrict__ dwl, int B, int E) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float w0 = w[row * 3 + 0], w1 = w[row * 3 + 1], w2 = w[row * 3 + 2];
  const float* d = dout + (long)row * 3 * E;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int e = lane; e < E; e += 64) {
    const float d0 = d[e], d1 = d[E + e], d2 = d[2 * E + e];
    s0 += d0 * f1[(long)row * E + e];
    s1 += d1 * f2[(long)row * E + e];
    s2 += d2 * f3[(long)row * E + e];
    df1[(long)row * E + e] = d0 * w0;
    df2[(long)row * E + e] = d1 * w1;
    df3[(long)row * E + e] = d2 * w2;
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) {
    const float dot = w0 * s0 + w1 * s1 + w2 * s2;
    dwl[row * 3 + 0] = w0 * (s0 - dot);
    dwl[row * 3 + 1] = w1 * (s1 - dot);
    dwl[row * 3 + 2] = w2 * (s2 - dot);
  }
}
int weighted_concat_fwd(const float* wl, const float* f1, const float* f2, const float* f3, float* w, float* out, int B, int E,
                        hipStream_t st) {
  hipLaunchKernelGGL(weighted_concat_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, wl, f1, f2, f3, w, out, B, E);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
int weighted_concat_bwd(const float* dout, const float* w, const float* f1, const float* f2, const float* f3, float* df1,
                        float* df2, float* df3, float* dwl, int B, int E, hipStream_t st) {
  hipLaunchKernelGGL(weighted_concat_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, dout, w, f1, f2, f3, df1, df2, df3, dwl,
                     B, E);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// ------------------------------------------------------------------------------------------------ cross entropy
// loss = mean_b(-log softmax(logits)[b, y_b]); dlogits = grad_scale * (softmax - onehot) / B. Single workgroup,
// fixed reduction order. nn.CrossEntropyLoss(): Trainer.py:17,68; Tester.py:20,57.
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                         float* __restrict__ loss, float* __restrict__ dlogits,
                                                         float* __restrict__ probs, int B, int C, float grad_scale) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float* z = logits + (long)b * C;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[c]);
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += __expf(z[c] - mx);
    const float lse = mx + __logf(sum);
    long long y = labels[b];
    if (y < 0) y = 0;
    if (y >= C) y = C - 1;
    acc += lse - z[y];
    for (int c = 0; c < C; ++c) {
      const float p = __expf(z[c] - lse);
      if (probs) probs[(long)b * C + c] = p;
      if (dlogits) dlogits[(long)b * C + c] = grad_scale * (p - (c == y ? 1.f : 0.f)) / (float)B;
    }
  }
  const float total = block_sum(acc, red);
  if (threadIdx.x == 0) *loss = total / (float)B;
}
int ce_fwd_bwd(const float* logits, const long long* labels, float* loss, float* dlogits, float* probs, int B, int C,
               float grad_scale, hipStream_t st) {
  hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(1), dim3(256), 0, st, logits, labels, loss, dlogits, probs, B, C, grad_scale);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// ------------------------------------------------------------------------------------------------ dropout
// y = x * keep / (1-p); mask saved as bytes. Counter-based (seed, element index): reproducible per step.
__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ mask,
                                   long n, float p, unsigned long long seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned char keep = hash_uniform(seed, (unsigned long long)i) >= p;
    mask[i] = keep;
    y[i] = keep ? x[i] / (1.f - p) : 0.f;
  }
}
__global__ void dropout_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ mask, float* __restrict__ dx,
                                   long n, float p) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = mask[i] ? dy[i] / (1.f - p) : 0.f;
}
int dropout_fwd(const float* x, float* y, unsigned char* mask, long n, float p, unsigned long long seed, hipStream_t st) {
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3((int)min((n + 255) / 256, 1024L)), dim3(256), 0, st, x, y, mask, n, p, seed);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
int dropout_bwd(const float* dy, const unsigned char* mask, float* dx, long n, float p, hipStream_t st) {
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3((int)min((n + 255) / 256, 1024L)), dim3(256), 0, st, dy, mask, dx, n, p);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// ------------------------------------------------------------------------------------------------ element-wise 2-D
// out[r][c] = op(a[r][c], b[r][c]) on row-strided fp32 views
#define EW_COPY 0
#define EW_ADD 1
#define EW_RELU_BWD 2  // a * (b > 0)
#define EW_GELU_BWD 3  // a * gelu'(b)
__global__ void ew2d_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                            float* __restrict__ out, long ldo, int rows, int cols, int op) {
  const long total = (long)rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    const float av = a[(long)r * lda + c];
    float v;
    switch (op) {
      case EW_ADD: v = av + b[(long)r * ldb + c]; break;
      case EW_RELU_BWD: v = b[(long)r * ldb + c] > 0.f ? av : 0.f; break;
      case EW_GELU_BWD: v = av * gelu_erf_grad(b[(long)r * ldb + c]); break;
      default: v = av;
    }
    out[(long)r * ldo + c] = v;
  }
}
// out[r] = [a[r] | b[r] | c[r]] and its backward da = dcat[:, :E] + add0, db = dcat[:, E:2E], dc = dcat[:, 2E:]
// (the attention-weights MLP input of the weighted head, MultimodalModel.py:264)
__global__ void cat3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                            float* __restrict__ out, int rows, int E) {
  const long total = (long)rows * 3 * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (3 * E);
    const int j = (int)(i - r * 3 * E), part = j / E, e = j - part * E;
    const float* src = part == 0 ? a : (part == 1 ? b : c);
    out[i] = src[r * E + e];
  }
}
__global__ void cat3_bwd_kernel(const float* __restrict__ dcat, const float* __restrict__ add0, float* __restrict__ da,
                                float* __restrict__ db, float* __restrict__ dc, int rows, int E) {
  const long total = (long)rows * 3 * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (3 * E);
    const int j = (int)(i - r * 3 * E), part = j / E, e = j - part * E;
    const float v = dcat[i];
    if (part == 0) da[r * E + e] = v + add0[r * E + e];
    else if (part == 1) db[r * E + e] = v;
    else dc[r * E + e] = v;
  }
}
int cat3_fwd(const float* a, const float* b, const float* c, float* out, int rows, int E, hipStream_t st) {
  const long total = (long)rows * 3 * E;
  hipLaunchKernelGGL(cat3_kernel, dim3((int)min((total + 255) / 256, 2048L)), dim3(256), 0, st, a, b, c, out, rows, E);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
int cat3_bwd(const float* dcat, const float* add0, float* da, float* db, float* dc, int rows, int E, hipStream_t st) {
  const long total = (long)rows * 3 * E;
  hipLaunchKernelGGL(cat3_bwd_kernel, dim3((int)min((total + 255) / 256, 2048L)), dim3(256), 0, st, dcat, add0, da, db, dc,
                     rows, E);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

int ew2d(int op, const float* a, long lda, const float* b, long ldb, float* out, long ldo, int rows, int cols, hipStream_t st) {
  const long total = (long)rows * cols;
  hipLaunchKernelGGL(ew2d_kernel, dim3((int)min((total + 255) / 256, 2048L)), dim3(256), 0, st, a, lda, b, ldb, out, ldo,
                     rows, cols, op);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
