// The fusion head's Linear -> BatchNorm1d -> activation -> Dropout units as ONE launch each (the head sits in the step's serial
// section: both encoders wait on these B x 128..768 problems; profiles/r04: ~100 launches of 2-8 us, each behind ~4 us of dispatch
// latency). A/B on one box (MMSA_DISABLE=head_units), ms per step: 16.49-16.50 against 16.54-16.60.
//
// What did NOT pay and was removed again: the whole one-key CrossModalTransformer (in-projection -> out-projection -> gate -> mix
// -> LayerNorm, four dependent row-local products) as one launch with 16 batch rows per workgroup, and its backward as a row-local
// launch + one 8-problem batch-row launch: -38 launches per step, but a 16-row chain is serial work for ONE CU — 33 MFLOP of exact
// fp32 products per module at 256 FLOP/clk/CU = 13.6 us on four CUs before any latency, where the separate launches spread each
// product over 64 workgroups — and the step did not move (16.51-16.53 against 16.49-16.50 with only the units fused).
#include "common.h"
#include "ops.h"

bool head_units_on() { return !mmsa_disabled("head_units"); }

// ---- the unit (MultimodalModel.py:179-225, 377-381, 416-424) ---------------------------------------------------------------------
// (were 3: batch-row GEMM, bn_small_fwd_kernel, dropout_fwd_kernel.) BatchNorm1d sums over the batch rows, so a workgroup takes 16
// output features and ALL rows (<= 256): its 4 waves compute the 16-row tiles of the Linear (gemm_tiny_kernel's summation order:
// chunk c of 64 k-values into partial c % KS), leave them in LDS, and the statistics / apply / Dropout phases are
// bn_small_fwd_kernel's and dropout_fwd_kernel's expressions on that tile (same (row lane, column) partition of the sums).
#define HU_ROWS 256
template <int K>
__global__ __launch_bounds__(256) void unit_fwd_kernel(UnitFwd p) {
  constexpr int NCH = K / 64, KS = K >= 256 ? 4 : (K >= 128 ? 2 : 1);
  __shared__ float zt[HU_ROWS][17];
  __shared__ double rs[32][16], rq[32][16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const float* wrow = p.W + (long)(n0 + r) * K + g * 16;
  const float bias = p.bias[n0 + r];
  for (int tm = w; tm * 16 < p.M; tm += 4) {
    const int m = tm * 16 + r;
    const float* xrow = p.x + (long)min(m, p.M - 1) * p.ldx + g * 16;
    const bool live = m < p.M;  // (rows past the batch: zeros, like the batch-row kernel; they reach no stored row and no statistic)
    f32x4 part[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) part[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int CB = NCH < 4 ? NCH : 4;  // chunks whose 2 x 4 operand vectors per lane are in flight together
#pragma unroll
    for (int cb = 0; cb < NCH; cb += CB) {
      f32x4 a[CB][4], b[CB][4];
#pragma unroll
      for (int u = 0; u < CB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a[u][q] = *(const f32x4*)(xrow + (cb + u) * 64 + 4 * q);
          b[u][q] = *(const f32x4*)(wrow + (cb + u) * 64 + 4 * q);
        }
      __builtin_amdgcn_sched_barrier(0);  // (all of the block's loads before its first MFMA: one memory round trip per block)
#pragma unroll
      for (int u = 0; u < CB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 av = live ? a[u][q] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            part[(cb + u) % KS] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b[u][q][e], part[(cb + u) % KS], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 v = part[0];
#pragma unroll
    for (int u = 1; u < KS; ++u) v += part[u];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = tm * 16 + 4 * g + j;
      const float val = apply_act(v[j] + bias, p.lin_act);
      zt[row][r] = val;
      if (row < p.M) p.z[(long)row * p.N + n0 + r] = val;
    }
  }
  __syncthreads();
  const int rl = tid >> 3;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int cc = (tid & 7) + 8 * hh;
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < HU_ROWS / 32; ++i) {
      const int rr = rl + 32 * i;
      const float xv = rr < p.M ? zt[rr][cc] : 0.f;
      s += xv;
      q += xv * xv;
    }
    rs[rl][cc] = s;
    rq[rl][cc] = q;
  }
  __syncthreads();
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int cc = (tid & 7) + 8 * hh, c = n0 + cc;
    double S = 0, Q = 0;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) { S += rs[j][cc]; Q += rq[j][cc]; }
    const double mu = S / p.M;
    double var = Q / p.M - mu * mu;
    if (var < 0) var = 0;
    const float muf = (float)mu, isf = (float)(1.0 / sqrt(var + (double)p.eps));
    if (rl == 0) {
      p.mean[c] = muf;
      p.invstd[c] = isf;
      if (p.rmean) {
        const double unb = p.M > 1 ? var * ((double)p.M / (p.M - 1)) : var;
        p.rmean[c] = (float)((1.0 - p.momentum) * p.rmean[c] + p.momentum * mu);
        p.rvar[c] = (float)((1.0 - p.momentum) * p.rvar[c] + p.momentum * unb);
      }
    }
    const float sc = isf * p.gamma[c], sh = p.beta[c] - muf * sc;
#pragma unroll
    for (int i = 0; i < HU_ROWS / 32; ++i) {
      const int rr = rl + 32 * i;
      if (rr < p.M) {
        const long idx = (long)rr * p.N + c;
        float y = apply_act(zt[rr][cc] * sc + sh, p.bn_act);
        if (p.y) p.y[idx] = y;
        if (p.mask) {
          const unsigned char keep = hash_uniform(p.seed, (unsigned long long)idx) >= p.drop_p;
          p.mask[idx] = keep;
          y = keep ? y / (1.f - p.drop_p) : 0.f;
          p.yd[idx] = y;
        }
        if (p.out2) p.out2[idx] = y;
      }
    }
  }
}

int unit_fwd_fused(const UnitFwd& p, int K, hipStream_t st) {
  if (p.M <= 0 || p.M > HU_ROWS || (p.N % 16) || (p.ldx % 4) || ((uintptr_t)p.x & 15) || ((uintptr_t)p.W & 15)) return MMSA_ERR_UNSUPPORTED;
  const dim3 grid(p.N / 16);
  switch (K) {
    case 128: hipLaunchKernelGGL(unit_fwd_kernel<128>, grid, dim3(256), 0, st, p); break;
    case 256: hipLaunchKernelGGL(unit_fwd_kernel<256>, grid, dim3(256), 0, st, p); break;
    case 768: hipLaunchKernelGGL(unit_fwd_kernel<768>, grid, dim3(256), 0, st, p); break;
    default: return MMSA_ERR_UNSUPPORTED;
  }
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
