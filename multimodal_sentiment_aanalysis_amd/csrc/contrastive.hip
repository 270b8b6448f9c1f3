// N1: the reference's two contrastive losses, each as ONE fused forward + backward launch.
//   mode 0  supervised InfoNCE with a (learnable) temperature — MultimodalTransformerModel.compute_contrastive_loss,
//           MultimodalModel.py:232-260: f = normalize(feat); S = f1 f2^T / T; pos_mask = same label, diagonal excluded;
//           S -= rowmax(S); e = exp(S); loss_i = -log((sum_j e pos + 1e-12) / (sum_j e + 1e-12)); mean over rows.
//           The row maximum is part of the autograd graph in the reference (torch.max), so its gradient path (to the
//           first arg-max of the row) is reproduced; it only matters through the two 1e-12 terms.
//   mode 1  two-view supervised contrastive loss — train.contrastive_loss, train.py:16-40: z = [normalize(z1);
//           normalize(z2)]; S = z z^T / T (T = 0.1, no shift); the diagonal is removed from the soft-max denominator;
//           loss_i = -sum_j mask_ij (S_ij - log(sum_{k != i} e^{S_ik} + 1e-8)) / (sum_j mask_ij + 1e-8); mean over the 2B rows.
// Shapes are B x 256 / 2B x 128 (B = 64): one workgroup walks the phases (normalise -> S -> row statistics and dS ->
// feature gradients -> normalisation backward) through a small global workspace; every reduction has a fixed order.
#include "common.h"
#include "ops.h"

#define CL_THREADS 1024

struct ClArgs {
  const float* x1;      // [B][D]
  const float* x2;      // [B][D]
  const long long* labels;
  const float* temp_ptr;  // mode 0: device scalar
  float temp;             // mode 1
  float* loss;
  float* dx1;
  float* dx2;
  float* dtemp;  // mode 0, may be null
  float* ws;
  int B, D, mode;
  float grad_scale;
};

size_t contrastive_ws_bytes(int B, int D) {
  const size_t R = 2 * (size_t)B, N = R;
  return (2 * R * D + R + 2 * N * N + 2 * N) * sizeof(float);
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(CL_THREADS) void contrastive_kernel(ClArgs a) {
  const int B = a.B, D = a.D, R = 2 * B;
  const int N = a.mode == 0 ? B : R;  // rows / columns of S
  float* fn = a.ws;                   // [R][D] normalised rows: x1 rows then x2 rows
  float* dfn = fn + (size_t)R * D;    // [R][D]
  float* inv = dfn + (size_t)R * D;   // [R]  1 / max(||x||, 1e-12)
  float* S = inv + R;                 // [N][N]
  float* dS = S + (size_t)N * N;      // [N][N]
  float* rowloss = dS + (size_t)N * N;  // [N]
  float* rowdt = rowloss + N;           // [N]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = CL_THREADS / 64;
  const float T = a.mode == 0 ? *a.temp_ptr : a.temp;
  const float invT = 1.0f / T;

  // 1. F.normalize(x, dim=1): x / max(||x||_2, 1e-12)
  for (int r = wave; r < R; r += nw) {
    const float* x = (r < B ? a.x1 + (size_t)r * D : a.x2 + (size_t)(r - B) * D);
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += x[d] * x[d];
    s = wave_sum_f(s);
    const float iv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    for (int d = lane; d < D; d += 64) fn[(size_t)r * D + d] = x[d] * iv;
    if (lane == 0) inv[r] = iv;
  }
  __syncthreads();
  // rows of S: mode 0: i -> fn[i] (x1), j -> fn[B + j] (x2); mode 1: both index the 2B stacked rows
  const float* FA = fn;
  const float* FB = a.mode == 0 ? fn + (size_t)B * D : fn;
  // 2. S = FA FB^T / T
  for (int idx = tid; idx < N * N; idx += CL_THREADS) {
    const int i = idx / N, j = idx - i * N;
    const float* p = FA + (size_t)i * D;
    const float* q = FB + (size_t)j * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(p[d], q[d], s);
    S[idx] = s * invT;
  }
  __syncthreads();
  // 3. row statistics, row loss, dS (already scaled by grad_scale / N)
  const float gs = a.grad_scale / (float)N;
  for (int i = wave; i < N; i += nw) {
    const long long li = a.labels[a.mode == 0 ? i : (i < B ? i : i - B)];
    const float* Si = S + (size_t)i * N;
    float* dSi = dS + (size_t)i * N;
    if (a.mode == 0) {
      float m = -INFINITY;
      int am = 0x7fffffff;
      for (int j = lane; j < N; j += 64) {
        const float v = Si[j];
        if (v > m) { m = v; am = j; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {  // max value, smallest index among equal maxima (torch.max on CPU)
        const float om = __shfl_xor(m, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (om > m || (om == m && oa < am)) { m = om; am = oa; }
      }
      float pos = 0.f, all = 0.f;
      for (int j = lane; j < N; j += 64) {
        const float e = __expf(Si[j] - m);
        const bool pm = j != i && a.labels[j] == li;
        all += e;
        if (pm) pos += e;
      }
      pos = wave_sum_f(pos);
      all = wave_sum_f(all);
      const float ip = 1.0f / (pos + 1e-12f), ia = 1.0f / (all + 1e-12f);
      const float gmax = pos * ip - all * ia;  // through -max(S_i): d(-log(pos+eps) + log(all+eps)) / dm
      float dt = 0.f;
      for (int j = lane; j < N; j += 64) {
        const float e = __expf(Si[j] - m);
        const bool pm = j != i && a.labels[j] == li;
        float g = e * ia - (pm ? e * ip : 0.f);
        if (j == am) g += gmax;
        dSi[j] = g * gs;
        dt += g * Si[j];
      }
      dt = wave_sum_f(dt);
      if (lane == 0) {
        rowloss[i] = -logf((pos + 1e-12f) * ia);
        rowdt[i] = -dt * invT;  // dS/dT = -S / T
      }
    } else {
      float sum = 0.f, cnt = 0.f, msum = 0.f;
      for (int j = lane; j < N; j += 64) {
        if (j == i) continue;
        const float s = Si[j];
        sum += __expf(s);
        const long long lj = a.labels[j < B ? j : j - B];
        if (lj == li) { cnt += 1.f; msum += s; }
      }
      sum = wave_sum_f(sum);
      cnt = wave_sum_f(cnt);
      msum = wave_sum_f(msum);
      const float lse = logf(sum + 1e-8f), ic = 1.0f / (cnt + 1e-8f), isum = 1.0f / (sum + 1e-8f);
      for (int j = lane; j < N; j += 64) {
        float g = 0.f;
        if (j != i) {
          const long long lj = a.labels[j < B ? j : j - B];
          g = cnt * ic * __expf(Si[j]) * isum - (lj == li ? ic : 0.f);
        }
        dSi[j] = g * gs;
      }
      if (lane == 0) {
        rowloss[i] = -(msum - cnt * lse) * ic;
        rowdt[i] = 0.f;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    float l = 0.f, dt = 0.f;
    for (int i = 0; i < N; ++i) { l += rowloss[i]; dt += rowdt[i]; }
    *a.loss = l / (float)N;
    if (a.dtemp && a.mode == 0) *a.dtemp = dt * gs;
  }
  // 4. gradients of the normalised rows
  for (int idx = tid; idx < R * D; idx += CL_THREADS) {
    const int r = idx / D, d = idx - r * D;
    float g = 0.f;
    if (a.mode == 0) {
      if (r < B) {
        for (int j = 0; j < N; ++j) g = fmaf(dS[(size_t)r * N + j], FB[(size_t)j * D + d], g);
      } else {
        const int j = r - B;
        for (int i = 0; i < N; ++i) g = fmaf(dS[(size_t)i * N + j], FA[(size_t)i * D + d], g);
      }
    } else {
      for (int j = 0; j < N; ++j) g = fmaf(dS[(size_t)r * N + j] + dS[(size_t)j * N + r], fn[(size_t)j * D + d], g);
    }
    dfn[idx] = g * invT;
  }
  __syncthreads();
  // 5. through the normalisation: dx = (dfn - fn (fn . dfn)) / max(||x||, eps)   (no projection term below eps)
  for (int r = wave; r < R; r += nw) {
    const float* f = fn + (size_t)r * D;
    const float* g = dfn + (size_t)r * D;
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot += f[d] * g[d];
    dot = wave_sum_f(dot);
    const float iv = inv[r];
    if (iv >= 1e12f) dot = 0.f;
    float* out = r < B ? a.dx1 + (size_t)r * D : a.dx2 + (size_t)(r - B) * D;
    for (int d = lane; d < D; d += 64) out[d] = (g[d] - f[d] * dot) * iv;
  }
}

static int contrastive_launch(ClArgs a, hipStream_t st) {
  if (a.B <= 0 || a.D <= 0 || a.B > 1024 || a.D > 4096) return MMSA_ERR_ARG;
  hipLaunchKernelGGL(contrastive_kernel, dim3(1), dim3(CL_THREADS), 0, st, a);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

int infonce_fwd_bwd(const float* feat1, const float* feat2, const long long* labels, const float* temperature, float* loss,
                    float* dfeat1, float* dfeat2, float* dtemp, int B, int D, float grad_scale, float* ws, hipStream_t st) {
  ClArgs a{feat1, feat2, labels, temperature, 0.f, loss, dfeat1, dfeat2, dtemp, ws, B, D, 0, grad_scale};
  return contrastive_launch(a, st);
}
int supcon_fwd_bwd(const float* z1, const float* z2, const long long* labels, float temperature, float* loss, float* dz1,
                   float* dz2, int B, int D, float grad_scale, float* ws, hipStream_t st) {
  if (!(temperature > 0.f)) return MMSA_ERR_ARG;
  ClArgs a{z1, z2, labels, nullptr, temperature, loss, dz1, dz2, nullptr, ws, B, D, 1, grad_scale};
  return contrastive_launch(a, st);
}
