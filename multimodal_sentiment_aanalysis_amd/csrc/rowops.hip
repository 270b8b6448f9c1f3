// HBM-bound row/column kernels of the BERT side: LayerNorm fwd/bwd (with the residual add done by the GEMM
// epilogue upstream), embedding gather + LayerNorm, embedding scatter, column sums (bias gradients) and the
// deterministic two-stage column-partial finalizer. All activations are [rows][H] with H % 4 == 0; one wave
// handles one row with 4-element vector accesses, statistics in fp32.
#include "common.h"
#include "gemm_epilogue.h"  // Vec4
#include "ops.h"

// NCH = per-lane 4-element chunks (template): 1 -> H <= 256, 3 -> H <= 768 (BERT-base: no dead fourth chunk), 4 -> H <= 1024,
// 8 -> H <= 2048

// ------------------------------------------------------------------------------------------------ LayerNorm
template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int M, int H, float eps, unsigned char* __restrict__ q_out,
                                                            float* __restrict__ q_scales) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nch = H >> 2;
  for (int row = blockIdx.x * 4 + w; row < M; row += gridDim.x * 4) {
    const T* xr = x + (long)row * H;
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {  // loads are unconditional (clamped chunk, result masked): a branch around each
      const int ch = lane + 64 * c;   // load makes hipcc wait for every chunk separately — serial memory round trips
      const f32x4 t = Vec4<T>::load(xr + min(ch, nch - 1) * 4);
      v[c] = ch < nch ? t : f32x4{0.f, 0.f, 0.f, 0.f};
      s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[c][e] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
    T* yr = y + (long)row * H;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      const int chc = min(ch, nch - 1);
      const f32x4 g = *(const f32x4*)(gamma + chc * 4), b = *(const f32x4*)(beta + chc * 4);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[c][e] - mean) * rstd * g[e] + b[e];
      if (ch < nch) Vec4<T>::store(yr + ch * 4, o);
      v[c] = o;
    }
    // fp8 mode (gemm_fp8.hip): the row is in registers — its e4m3 image with the row's own scale (max |stored y| / 448) comes
    // out of this pass instead of a quantizer pass over y (fp8_quant_rows_kernel's arithmetic on the bf16-ROUNDED values)
    if (q_out) {
      float am = 0.f;
      bool bad = false;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (lane + 64 * c < nch) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[c][e] = (float)(T)v[c][e];
            const float a = fabsf(v[c][e]);
            bad |= !(a <= 3.0e38f);
            am = fmaxf(am, a);
          }
        }
      }
      if (bad) am = __builtin_inff();
      am = fmaxf(wave_max(am), 1e-20f);
      const float inv = 448.0f / am;
      if (lane == 0) q_scales[row] = am * (1.0f / 448.0f);
      unsigned char* qr = q_out + (long)row * H;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
          float f[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) f[e] = fminf(fmaxf(v[c][e] * inv, -448.0f), 448.0f);
          int pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
          pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], pk, true);
          *(int*)(qr + ch * 4) = pk;
        }
      }
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma ; per-block partial dgamma/dbeta -> part[blk][NQ][H]
// NQ = 3 adds the column sums of the stored dx: in BERT every LayerNorm input gradient is also the output gradient of
// the Linear in front of it, so its bias gradient comes out of this pass instead of a separate column-sum launch pair.
template <typename T, int NCH>
__global__ __launch_bounds__(512) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, T* __restrict__ dx,
                                                            float* __restrict__ part, int M, int H, int NQ) {
  extern __shared__ float lds[];  // [NW waves][H], reused per quantity
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, NW = blockDim.x >> 6;
  const int nch = H >> 2;
  f32x4 ag[NCH], ab[NCH], ax[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) { ag[c] = f32x4{0, 0, 0, 0}; ab[c] = f32x4{0, 0, 0, 0}; ax[c] = f32x4{0, 0, 0, 0}; }
  // two rows per iteration: both rows' loads are issued before either row's reductions (two memory round trips in
  // flight per wave instead of a serial load -> reduce -> store chain); written out twice, not as lambdas over the
  // accumulator arrays, which hipcc moved to scratch
  const int rstride = gridDim.x * NW;
  for (int rowb = blockIdx.x * NW + w; rowb < M; rowb += 2 * rstride) {
    const int rowA = rowb;
    const bool okA = rowA < M;
    const T* xrA = x + (long)(okA ? rowA : 0) * H;
    const T* drA = dy + (long)(okA ? rowA : 0) * H;
    const float muA = mean[okA ? rowA : 0], rsA = rstd[okA ? rowA : 0];
    f32x4 xvA[NCH], dvA[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      const int chc = min(ch, nch - 1);  // unconditional loads (clamped), masked below: no branch per chunk
      xvA[c] = Vec4<T>::load(xrA + chc * 4); dvA[c] = Vec4<T>::load(drA + chc * 4);
    }
    const int rowB = rowb + rstride;
    const bool okB = rowB < M;
    const T* xrB = x + (long)(okB ? rowB : 0) * H;
    const T* drB = dy + (long)(okB ? rowB : 0) * H;
    const float muB = mean[okB ? rowB : 0], rsB = rstd[okB ? rowB : 0];
    f32x4 xvB[NCH], dvB[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      const int chc = min(ch, nch - 1);  // unconditional loads (clamped), masked below: no branch per chunk
      xvB[c] = Vec4<T>::load(xrB + chc * 4); dvB[c] = Vec4<T>::load(drB + chc * 4);
    }
    f32x4 xhA[NCH], ggA[NCH];
    float s1A = 0.f, s2A = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      {
        const f32x4 g = *(const f32x4*)(gamma + min(ch, nch - 1) * 4);
        const float live = (ch < nch && okA) ? 1.f : 0.f;  // dead chunk / row past the end: contributes zeros
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dvl = dvA[c][e] * live;
          xhA[c][e] = (xvA[c][e] - muA) * rsA;
          ggA[c][e] = dvl * g[e];
          s1A += ggA[c][e];
          s2A += ggA[c][e] * xhA[c][e];
          ag[c][e] += dvl * xhA[c][e];
          ab[c][e] += dvl;
        }
      }
    }
    f32x4 xhB[NCH], ggB[NCH];
    float s1B = 0.f, s2B = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      {
        const f32x4 g = *(const f32x4*)(gamma + min(ch, nch - 1) * 4);
        const float live = (ch < nch && okB) ? 1.f : 0.f;  // dead chunk / row past the end: contributes zeros
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dvl = dvB[c][e] * live;
          xhB[c][e] = (xvB[c][e] - muB) * rsB;
          ggB[c][e] = dvl * g[e];
          s1B += ggB[c][e];
          s2B += ggB[c][e] * xhB[c][e];
          ag[c][e] += dvl * xhB[c][e];
          ab[c][e] += dvl;
        }
      }
    }
    s1A = wave_sum(s1A) / (float)H;
    s2A = wave_sum(s2A) / (float)H;
    if (okA) {
      T* oxr = dx + (long)rowA * H;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = rsA * (ggA[c][e] - s1A - xhA[c][e] * s2A);
            ax[c][e] += q_f32<T>(o[e]);  // the value as stored (what a column sum over dx would read)
          }
          Vec4<T>::store(oxr + ch * 4, o);
        }
      }
    }
    s1B = wave_sum(s1B) / (float)H;
    s2B = wave_sum(s2B) / (float)H;
    if (okB) {
      T* oxr = dx + (long)rowB * H;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = rsB * (ggB[c][e] - s1B - xhB[c][e] * s2B);
            ax[c][e] += q_f32<T>(o[e]);  // the value as stored (what a column sum over dx would read)
          }
          Vec4<T>::store(oxr + ch * 4, o);
        }
      }
    }
  }
  // combine the waves of the block, one quantity at a time through [NW][H] floats of LDS, then write the block partial
  for (int qn = 0; qn < NQ; ++qn) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) *(f32x4*)(lds + w * H + ch * 4) = qn == 0 ? ag[c] : (qn == 1 ? ab[c] : ax[c]);
    }
    __syncthreads();
    for (int col = threadIdx.x; col < H; col += blockDim.x) {
      float t = 0.f;
      for (int ww = 0; ww < NW; ++ww) t += lds[ww * H + col];
      part[((long)blockIdx.x * NQ + qn) * H + col] = t;
    }
    __syncthreads();
  }
}

// ---- bf16 rows of 768 / 1024 elements: half a wave per row, 16-byte accesses (round 4) -----------------------------------------
// A 768-element bf16 row is 1536 bytes = 32 lanes x 3 chunks of 16 bytes: each half of a wave takes one row (a wave-instruction
// then moves 1 KiB — two 512-byte row segments — instead of 512 bytes with the 8-byte accesses of the general kernels above, which
// streamed at 2.9-3.2 TB/s at 8192 x 768), row statistics are 32-lane reductions (5 xor steps), and the loads of the wave's NEXT
// row pair are in flight while the current one is reduced. Same expressions as the general kernels, another summation order.
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ void unpack8(bf16x8 t, float (&f)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)t[e];
}

template <int NC>  // H = 256 NC
__global__ __launch_bounds__(256) void layernorm_fwd_hw_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, bf16* __restrict__ y,
                                                               float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                               int M, float eps, unsigned char* __restrict__ q_out,
                                                               float* __restrict__ q_scales) {
  constexpr int H = 256 * NC;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane & 31, half = lane >> 5;
  const int npairs = (M + 1) >> 1, stride = gridDim.x * 4;
  int pr = blockIdx.x * 4 + w;
  if (pr >= npairs) return;
  float g[NC][8], b[NC][8];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = (h + 32 * c) * 8;
    const f32x4 g0 = *(const f32x4*)(gamma + col), g1 = *(const f32x4*)(gamma + col + 4);
    const f32x4 b0 = *(const f32x4*)(beta + col), b1 = *(const f32x4*)(beta + col + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { g[c][e] = g0[e]; g[c][4 + e] = g1[e]; b[c][e] = b0[e]; b[c][4 + e] = b1[e]; }
  }
  bf16x8 cur[NC], nxt[NC];
  auto load = [&](bf16x8 (&t)[NC], int pair) __attribute__((always_inline)) {
    const int row = min(2 * pair + half, M - 1);  // (the odd last row: the idle half re-reads it, nothing is stored)
    const bf16* xr = x + (long)row * H + h * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) t[c] = *(const bf16x8*)(xr + 256 * c);
  };
  load(cur, pr);
  for (; pr < npairs; pr += stride) {
    const bool more = pr + stride < npairs;
    if (more) load(nxt, pr + stride);
    const int row = 2 * pr + half;
    const bool live = row < M;
    float v[NC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      unpack8(cur[c], v[c]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[c][e];
    }
    const float mean = half_sum(s) * (1.0f / (float)H);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; q += d * d; }
    const float rstd = rsqrtf(half_sum(q) * (1.0f / (float)H) + eps);
    if (h == 0 && live) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
    bf16* yr = y + (long)row * H + h * 8;
    float am = 0.f;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = (bf16)((v[c][e] - mean) * rstd * g[c][e] + b[c][e]);
        v[c][e] = (float)o[e];  // the value as stored (the fp8 image below quantizes what the bf16 consumers see)
        const float a = fabsf(v[c][e]);
        bad |= !(a <= 3.0e38f);
        am = fmaxf(am, a);
      }
      if (live) *(bf16x8*)(yr + 256 * c) = o;
    }
    if (q_out) {  // fp8 mode: the row's e4m3 image with its own scale (layernorm_fwd_kernel's arithmetic)
      if (bad) am = __builtin_inff();
      am = fmaxf(half_max(am), 1e-20f);
      const float inv = 448.0f / am;
      if (h == 0 && live) q_scales[row] = am * (1.0f / 448.0f);
      unsigned char* qr = q_out + (long)row * H + h * 8;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(v[c][e] * inv, -448.0f), 448.0f);
        int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
        p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], p0, true);
        int p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
        p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], p1, true);
        if (live) *(int2*)(qr + 256 * c) = make_int2(p0, p1);
      }
    }
    if (more) {
#pragma unroll
      for (int c = 0; c < NC; ++c) cur[c] = nxt[c];
    }
  }
}

// backward of the same rows: dx and the per-workgroup partials of dgamma / dbeta (/ column sums of the stored dx) in the layout of
// layernorm_bwd_kernel ([blk][NQ][H]; the finalize launch is shared)
template <int NC>
__global__ __launch_bounds__(512) void layernorm_bwd_hw_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, bf16* __restrict__ dx,
                                                               float* __restrict__ part, int M, int NQ) {
  constexpr int H = 256 * NC;
  extern __shared__ float lds[];  // [NW waves][H], reused per quantity
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, NW = blockDim.x >> 6, h = lane & 31, half = lane >> 5;
  const int npairs = (M + 1) >> 1, stride = gridDim.x * NW;
  float g[NC][8];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = (h + 32 * c) * 8;
    const f32x4 g0 = *(const f32x4*)(gamma + col), g1 = *(const f32x4*)(gamma + col + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { g[c][e] = g0[e]; g[c][4 + e] = g1[e]; }
  }
  float ag[NC][8], ab[NC][8], ax[NC][8];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[c][e] = 0.f; ab[c][e] = 0.f; ax[c][e] = 0.f; }
  bf16x8 cx[NC], cd[NC], nx[NC], nd[NC];
  auto load = [&](bf16x8 (&tx)[NC], bf16x8 (&td)[NC], int pair) __attribute__((always_inline)) {
    const int row = min(2 * pair + half, M - 1);
    const bf16* xr = x + (long)row * H + h * 8;
    const bf16* dr = dy + (long)row * H + h * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) { tx[c] = *(const bf16x8*)(xr + 256 * c); td[c] = *(const bf16x8*)(dr + 256 * c); }
  };
  int pr = blockIdx.x * NW + w;
  if (pr < npairs) load(cx, cd, pr);
  for (; pr < npairs; pr += stride) {
    const bool more = pr + stride < npairs;
    if (more) load(nx, nd, pr + stride);
    const int row = 2 * pr + half;
    const bool live = row < M;
    const int rc = min(row, M - 1);
    const float mu = mean[rc], rs = rstd[rc];
    const float lv = live ? 1.f : 0.f;  // the idle half of an odd last pair contributes zeros
    float xh[NC][8], gg[NC][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float xv[8], dv[8];
      unpack8(cx[c], xv);
      unpack8(cd[c], dv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dvl = dv[e] * lv;
        xh[c][e] = (xv[e] - mu) * rs;
        gg[c][e] = dvl * g[c][e];
        s1 += gg[c][e];
        s2 += gg[c][e] * xh[c][e];
        ag[c][e] += dvl * xh[c][e];
        ab[c][e] += dvl;
      }
    }
    s1 = half_sum(s1) * (1.0f / (float)H);
    s2 = half_sum(s2) * (1.0f / (float)H);
    bf16* oxr = dx + (long)row * H + h * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = (bf16)(rs * (gg[c][e] - s1 - xh[c][e] * s2));
        ax[c][e] += (float)o[e] * lv;  // the value as stored (what a column sum over dx would read)
      }
      if (live) *(bf16x8*)(oxr + 256 * c) = o;
    }
    if (more) {
#pragma unroll
      for (int c = 0; c < NC; ++c) { cx[c] = nx[c]; cd[c] = nd[c]; }
    }
  }
  // the two halves of a wave hold the same columns (other rows): add them, then combine the waves through LDS as the general kernel
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ag[c][e] += __shfl_xor(ag[c][e], 32, 64);
      ab[c][e] += __shfl_xor(ab[c][e], 32, 64);
      ax[c][e] += __shfl_xor(ax[c][e], 32, 64);
    }
  for (int qn = 0; qn < NQ; ++qn) {
    if (half == 0) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float* dst = lds + w * H + (h + 32 * c) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = qn == 0 ? ag[c][e] : (qn == 1 ? ab[c][e] : ax[c][e]);
      }
    }
    __syncthreads();
    for (int col = threadIdx.x; col < H; col += blockDim.x) {
      float t = 0.f;
      for (int ww = 0; ww < NW; ++ww) t += lds[ww * H + col];
      part[((long)blockIdx.x * NQ + qn) * H + col] = t;
    }
    __syncthreads();
  }
}

// out_q[j] (+)= sum_b part[(b * nq + q) * n + j] for q < nq (up to 3 outputs in one launch; fixed order as below)
__global__ __launch_bounds__(256) void partial_finalize_multi_kernel(const float* __restrict__ part, int nblk, int nq, int n,
                                                                     float* __restrict__ out0, float* __restrict__ out1,
                                                                     float* __restrict__ out2, int accumulate) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int jj = blockIdx.x * 16 + c;  // column of the [nq * n] wide partial rows
  float t = 0.f;
  if (jj < nq * n)
    for (int b = r; b < nblk; b += 16) t += part[(long)b * nq * n + jj];
  red[r][c] = t;
  __syncthreads();
  if (r != 0 || jj >= nq * n) return;
  t = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) t += red[k][c];
  const int q = jj / n, j = jj - q * n;
  float* out = q == 0 ? out0 : (q == 1 ? out1 : out2);
  if (!out) return;  // a frozen parameter: its gradient is not produced (engine backwards, N2)
  out[j] = accumulate ? out[j] + t : t;
}

// out[j] (+)= scale * sum_b part[b * stride + j]  for j in [0, n)   (fixed order: bitwise reproducible)
// 256 threads = 16 columns x 16 partial-lanes: lane r sums partials r, r+16, ... (in order), then the 16 lane sums are
// added in lane order — a fixed summation tree, so results are reproducible; 16x shorter serial chains than one
// thread per column, and a wave reads 16 consecutive columns of 4 different partial rows (64-byte segments).
__global__ __launch_bounds__(256) void partial_finalize_kernel(const float* __restrict__ part, int nblk, long stride, int n,
                                                               float* __restrict__ out, int accumulate, float scale) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + c;
  float t = 0.f;
  if (j < n)
    for (int b = r; b < nblk; b += 16) t += part[(long)b * stride + j];
  red[r][c] = t;
  __syncthreads();
  if (r == 0 && j < n) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][c];
    s *= scale;
    out[j] = accumulate ? out[j] + s : s;
  }
}

int partial_finalize(const float* part, int nblk, long stride, int n, float* out, int accumulate, float scale,
                     hipStream_t st) {
  hipLaunchKernelGGL(partial_finalize_kernel, dim3(cdiv(n, 16)), dim3(256), 0, st, part, nblk, stride, n, out,
                     accumulate, scale);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

#define LN_DISPATCH(KERNEL, T, H, ...)                                                           \
  do {                                                                                           \
    if ((H) <= 256) hipLaunchKernelGGL((KERNEL<T, 1>), __VA_ARGS__);                              \
    else if ((H) <= 768) hipLaunchKernelGGL((KERNEL<T, 3>), __VA_ARGS__);                         \
    else if ((H) <= 1024) hipLaunchKernelGGL((KERNEL<T, 4>), __VA_ARGS__);                        \
    else hipLaunchKernelGGL((KERNEL<T, 8>), __VA_ARGS__);                                         \
  } while (0)

// q_out / q_scales (optional, bf16 only): also the per-token e4m3 image of y ([M][H] bytes) and its scales [M]
int layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                  int M, int H, float eps, hipStream_t st, void* q_out, float* q_scales) {
  if (H % 4 || H > 2048) return MMSA_ERR_ARG;
  if (q_out && (dtype != MMSA_BF16 || !q_scales)) return MMSA_ERR_ARG;
  // bf16 rows of 768 / 1024 elements: the half-wave kernel (16-byte accesses, next row pair prefetched); MMSA_DISABLE=ln_halfwave
  if (dtype == MMSA_BF16 && (H == 768 || H == 1024) && M >= 64 && !mmsa_disabled("ln_halfwave")) {
    const int gridh = min(cdiv(cdiv(M, 2), 4), 1024);
    if (H == 768)
      hipLaunchKernelGGL(layernorm_fwd_hw_kernel<3>, dim3(gridh), dim3(256), 0, st, (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, M,
                         eps, (unsigned char*)q_out, q_scales);
    else
      hipLaunchKernelGGL(layernorm_fwd_hw_kernel<4>, dim3(gridh), dim3(256), 0, st, (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, M,
                         eps, (unsigned char*)q_out, q_scales);
    MMSA_CHECK_LAUNCH();
    return MMSA_OK;
  }
  const int grid = min(cdiv(M, 4), 4096);
  if (dtype == MMSA_BF16)
    LN_DISPATCH(layernorm_fwd_kernel, bf16, H, dim3(grid), dim3(256), 0, st, (const bf16*)x, gamma, beta, (bf16*)y, mean,
                rstd, M, H, eps, (unsigned char*)q_out, q_scales);
  else
    LN_DISPATCH(layernorm_fwd_kernel, float, H, dim3(grid), dim3(256), 0, st, (const float*)x, gamma, beta, (float*)y,
                mean, rstd, M, H, eps, (unsigned char*)nullptr, (float*)nullptr);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// one wave per row, 8 waves per workgroup, up to 256 workgroups (one per CU, 8 waves; fewer partial rows for the finalize: with 256 x 4 waves the
// kernel ran one wave per SIMD and streamed at 1.5 TB/s)
#define LN_BWD_BLOCKS 256
#define LN_BWD_WAVES 8
size_t layernorm_bwd_ws_bytes(int H) { return (size_t)LN_BWD_BLOCKS * 3 * H * sizeof(float); }

// dxsum (optional): column sums of the stored dx (+)= -> the bias gradient of the Linear that produced this LN's input
// Batched finalize of deferred LayerNorm backwards (round 4): the parameter gradients of a LayerNorm (and the bias gradient that
// rides along) are needed only by the optimizer, but their finalize was a 6 us launch between two GEMMs of the text encoder's serial
// backward chain, 25 times a step. With `defer` set, layernorm_bwd launches the row kernel only — into a partial buffer of the
// CALLER'S that stays untouched until the flush — and records the finalize as a job; layernorm_bwd_finalize_batch runs up to
// LN_FIN_MAX jobs as ONE launch (blockIdx.y = job). Same kernel body, same summation order: bit-identical gradients.
__global__ __launch_bounds__(256) void partial_finalize_batch_kernel(LnFinBatch b) {
  __shared__ float red[16][17];
  const LnFinJob j = b.job[blockIdx.y];
  const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int jj = blockIdx.x * 16 + c;
  const int n = b.H, nq = j.nq;
  float t = 0.f;
  if (jj < nq * n)
    for (int k = r; k < j.nblk; k += 16) t += j.part[(long)k * nq * n + jj];
  red[r][c] = t;
  __syncthreads();
  if (r != 0 || jj >= nq * n) return;
  t = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) t += red[k][c];
  const int q = jj / n, col = jj - q * n;
  float* out = q == 0 ? j.out0 : (q == 1 ? j.out1 : j.out2);
  if (!out) return;
  out[col] = j.accumulate ? out[col] + t : t;
}
int layernorm_bwd_finalize_batch(const LnFinJob* jobs, int n, int H, hipStream_t st) {
  for (int i = 0; i < n; i += LN_FIN_MAX) {
    LnFinBatch b;
    b.H = H;
    const int m = n - i < LN_FIN_MAX ? n - i : LN_FIN_MAX;
    for (int k = 0; k < m; ++k) b.job[k] = jobs[i + k];
    hipLaunchKernelGGL(partial_finalize_batch_kernel, dim3(cdiv(3 * H, 16), m), dim3(256), 0, st, b);
    MMSA_CHECK_LAUNCH();
  }
  return MMSA_OK;
}

int layernorm_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                  void* dx, float* dgamma, float* dbeta, int accumulate, float* ws, int M, int H, hipStream_t st,
                  float* dxsum, LnFinJob* defer) {
  if (H % 4 || H > 2048) return MMSA_ERR_UNSUPPORTED;
  if (defer) defer->part = nullptr;
  const int nw = H > 1024 ? 4 : LN_BWD_WAVES;
  const int grid = min(cdiv(M, nw), LN_BWD_BLOCKS);
  const size_t lds = (size_t)nw * H * sizeof(float);
  const int nq = dxsum ? 3 : 2;
  // (H = 768 only: at 1024 the three accumulator sets + two row pairs of raw data exceed 256 registers)
  if (dtype == MMSA_BF16 && H == 768 && M >= 64 && !mmsa_disabled("ln_halfwave")) {
    const int gridh = min(cdiv(cdiv(M, 2), nw), LN_BWD_BLOCKS);
    hipLaunchKernelGGL(layernorm_bwd_hw_kernel<3>, dim3(gridh), dim3(64 * nw), lds, st, (const bf16*)dy, (const bf16*)x, mean, rstd,
                       gamma, (bf16*)dx, ws, M, nq);
    MMSA_CHECK_LAUNCH();
    if (!dgamma && !dbeta && !dxsum) return MMSA_OK;
    if (defer) { *defer = LnFinJob{ws, gridh, nq, accumulate, dgamma, dbeta, dxsum}; return MMSA_OK; }
    hipLaunchKernelGGL(partial_finalize_multi_kernel, dim3(cdiv(nq * H, 16)), dim3(256), 0, st, (const float*)ws, gridh, nq, H,
                       dgamma, dbeta, dxsum, accumulate);
    MMSA_CHECK_LAUNCH();
    return MMSA_OK;
  }
  if (dtype == MMSA_BF16)
    LN_DISPATCH(layernorm_bwd_kernel, bf16, H, dim3(grid), dim3(64 * nw), lds, st, (const bf16*)dy, (const bf16*)x, mean, rstd,
                gamma, (bf16*)dx, ws, M, H, nq);
  else
    LN_DISPATCH(layernorm_bwd_kernel, float, H, dim3(grid), dim3(64 * nw), lds, st, (const float*)dy, (const float*)x, mean,
                rstd, gamma, (float*)dx, ws, M, H, nq);
  MMSA_CHECK_LAUNCH();
  if (!dgamma && !dbeta && !dxsum) return MMSA_OK;  // wholly frozen layer: only dx was wanted
  if (defer) { *defer = LnFinJob{ws, grid, nq, accumulate, dgamma, dbeta, dxsum}; return MMSA_OK; }
  // partial layout is [blk][nq][H]: one launch finalizes every output (null = that parameter is frozen)
  hipLaunchKernelGGL(partial_finalize_multi_kernel, dim3(cdiv(nq * H, 16)), dim3(256), 0, st, (const float*)ws, grid, nq, H,
                     dgamma, dbeta, dxsum, accumulate);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// ------------------------------------------------------------------------------------------------ column sums
// part[chunk][N] = sum over the chunk's rows of x[row][:]; finalize adds the chunks in order (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, long ldx, float* __restrict__ part,
                                                             int M, int N, int rows_per_chunk) {
  // 256 threads = 16 column threads (4 columns each: a 64-column group per block) x 16 row lanes, so a narrow matrix
  // (N = 768: 12 column groups) still launches chunks x 12 workgroups instead of `chunks`.
  __shared__ f32x4 red[256];
  const int ct = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int col = (blockIdx.y * 16 + ct) * 4;
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  f32x4 a = {0, 0, 0, 0};
  if (col < N)
    for (int r = r0 + rl; r < r1; r += 16) a += Vec4<T>::load(x + (long)r * ldx + col);
  red[threadIdx.x] = a;
  __syncthreads();
  if (rl == 0 && col < N) {
#pragma unroll
    for (int j = 1; j < 16; ++j) a += red[j * 16 + ct];
    *(f32x4*)(part + (long)blockIdx.x * N + col) = a;
  }
}

#define COLSUM_CHUNKS 128
size_t colsum_ws_bytes(int N) { return (size_t)COLSUM_CHUNKS * N * sizeof(float); }

// small / unaligned widths (the 3-class logits) and short matrices: one thread per column, fixed order
template <typename T>
__global__ void colsum_small_kernel(const T* __restrict__ x, long ldx, float* __restrict__ out, int accumulate, int M, int N) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  float a = 0.f;
  int r = 0;
  for (; r + 8 <= M; r += 8) {  // eight independent loads in flight, summed in row order
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = to_f32<T>(x[(long)(r + e) * ldx + c]);
#pragma unroll
    for (int e = 0; e < 8; ++e) a += v[e];
  }
  for (; r < M; ++r) a += to_f32<T>(x[(long)r * ldx + c]);
  out[c] = accumulate ? out[c] + a : a;
}

int colsum(int dtype, const void* x, long ldx, float* out, int accumulate, float* ws, int M, int N, hipStream_t st) {
  // few rows (the fusion head: M = batch): one pass, one launch (the two-stage form costs a second launch for nothing)
  if (N % 4 || ldx % 4 || M <= 128) {
    if (dtype == MMSA_BF16)
      hipLaunchKernelGGL(colsum_small_kernel<bf16>, dim3(cdiv(N, 64)), dim3(64), 0, st, (const bf16*)x, ldx, out, accumulate, M, N);
    else
      hipLaunchKernelGGL(colsum_small_kernel<float>, dim3(cdiv(N, 64)), dim3(64), 0, st, (const float*)x, ldx, out, accumulate, M, N);
    MMSA_CHECK_LAUNCH();
    return MMSA_OK;
  }
  const int rpc = cdiv(M, COLSUM_CHUNKS);
  const int chunks = cdiv(M, rpc);
  dim3 grid(chunks, cdiv(N, 64));
  if (dtype == MMSA_BF16)
    hipLaunchKernelGGL(colsum_partial_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, ldx, ws, M, N, rpc);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, ws, M, N, rpc);
  MMSA_CHECK_LAUNCH();
  return partial_finalize(ws, chunks, N, N, out, accumulate, 1.f, st);
}

// ------------------------------------------------------------------------------------------------ embeddings
// e[m][:] = word[ids[m]] + pos[m % S] + type[0]   (HF BertEmbeddings with default position / token-type ids)
template <typename T>
__global__ __launch_bounds__(256) void embed_gather_kernel(const long long* __restrict__ ids, const T* __restrict__ word,
                                                           const T* __restrict__ pos, const T* __restrict__ type,
                                                           T* __restrict__ e, int M, int S, int H, int vocab) {
  const int nch = H >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)M * nch; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / nch), ch = (int)(i - (long)m * nch);
    long long id = ids[m];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    const f32x4 w = Vec4<T>::load(word + id * H + ch * 4);
    const f32x4 p = Vec4<T>::load(pos + (long)(m % S) * H + ch * 4);
    const f32x4 t = Vec4<T>::load(type + ch * 4);
    Vec4<T>::store(e + (long)m * H + ch * 4, w + p + t);
  }
}

int embed_gather(int dtype, const long long* ids, const void* word, const void* pos, const void* type, void* e, int M,
                 int S, int H, int vocab, hipStream_t st) {
  if (H % 4) return MMSA_ERR_ARG;
  const long total = (long)M * (H / 4);
  const int grid = (int)min((total + 255) / 256, (long)4096);
  if (dtype == MMSA_BF16)
    hipLaunchKernelGGL(embed_gather_kernel<bf16>, dim3(grid), dim3(256), 0, st, ids, (const bf16*)word, (const bf16*)pos,
                       (const bf16*)type, (bf16*)e, M, S, H, vocab);
  else
    hipLaunchKernelGGL(embed_gather_kernel<float>, dim3(grid), dim3(256), 0, st, ids, (const float*)word,
                       (const float*)pos, (const float*)type, (float*)e, M, S, H, vocab);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// dword[ids[m]] += de[m] (fp32 atomics: rows collide only for repeated tokens); dpos[s] (+)= sum_b de[b*S+s]
template <typename T>
__global__ __launch_bounds__(256) void embed_scatter_word_kernel(const long long* __restrict__ ids,
                                                                 const T* __restrict__ de, float* __restrict__ dword,
                                                                 int M, int H, int vocab) {
  const int nch = H >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)M * nch; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / nch), ch = (int)(i - (long)m * nch);
    long long id = ids[m];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    const f32x4 g = Vec4<T>::load(de + (long)m * H + ch * 4);
    float* dst = dword + id * H + ch * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(dst + e, g[e]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_pos_grad_kernel(const T* __restrict__ de, float* __restrict__ dpos, int B,
                                                             int S, int H, int accumulate) {
  const int nch = H >> 2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)S * nch) return;
  const int s = (int)(i / nch), ch = (int)(i - (long)s * nch);
  f32x4 a = {0, 0, 0, 0};
  for (int b = 0; b < B; ++b) a += Vec4<T>::load(de + ((long)b * S + s) * H + ch * 4);
  float* dst = dpos + (long)s * H + ch * 4;
  if (accumulate) a += *(const f32x4*)dst;
  *(f32x4*)dst = a;
}

int embed_backward(int dtype, const long long* ids, const void* de, float* dword, float* dpos, float* dtype0,
                   int accumulate, float* ws, int B, int S, int H, int vocab, int maxpos, hipStream_t st) {
  if (H % 4) return MMSA_ERR_ARG;
  const int M = B * S;
  if (!accumulate) {
    if (hipMemsetAsync(dword, 0, (size_t)vocab * H * sizeof(float), st) != hipSuccess) return MMSA_ERR_LAUNCH;
    if (maxpos > S &&
        hipMemsetAsync(dpos + (long)S * H, 0, (size_t)(maxpos - S) * H * sizeof(float), st) != hipSuccess)
      return MMSA_ERR_LAUNCH;
  }
  const long total = (long)M * (H / 4);
  const int grid = (int)min((total + 255) / 256, (long)4096);
  const int pgrid = cdiv((long)S * (H / 4), 256);
  if (dtype == MMSA_BF16) {
    hipLaunchKernelGGL(embed_scatter_word_kernel<bf16>, dim3(grid), dim3(256), 0, st, ids, (const bf16*)de, dword, M, H,
                       vocab);
    hipLaunchKernelGGL(embed_pos_grad_kernel<bf16>, dim3(pgrid), dim3(256), 0, st, (const bf16*)de, dpos, B, S, H,
                       accumulate);
  } else {
    hipLaunchKernelGGL(embed_scatter_word_kernel<float>, dim3(grid), dim3(256), 0, st, ids, (const float*)de, dword, M, H,
                       vocab);
    hipLaunchKernelGGL(embed_pos_grad_kernel<float>, dim3(pgrid), dim3(256), 0, st, (const float*)de, dpos, B, S, H,
                       accumulate);
  }
  MMSA_CHECK_LAUNCH();
  // token-type row 0 receives every token's gradient (default token_type_ids = 0); row 1 gets none
  return colsum(dtype, de, H, dtype0, accumulate, ws, M, H, st);
}

// ------------------------------------------------------------------------------------------------ tanh backward
// dx = dy * (1 - y^2)  (BERT pooler)
template <typename T>
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                       long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 d = Vec4<T>::load(dy + i * 4), v = Vec4<T>::load(y + i * 4);
    Vec4<T>::store(dx + i * 4, d * (1.0f - v * v));
  }
}
int tanh_bwd(int dtype, const void* dy, const void* y, void* dx, long n, hipStream_t st) {
  if (n % 4) return MMSA_ERR_ARG;
  const int grid = (int)min((n / 4 + 255) / 256, 2048L);
  if (dtype == MMSA_BF16)
    hipLaunchKernelGGL(tanh_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)dy, (const bf16*)y, (bf16*)dx, n / 4);
  else
    hipLaunchKernelGGL(tanh_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dy, (const float*)y, (float*)dx,
                       n / 4);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// ------------------------------------------------------------------------------------------------ grouped bias gradients
__global__ __launch_bounds__(256) void bias_pick_kernel(const float* __restrict__ src, float* __restrict__ dst, int n,
                                                        const float* __restrict__ src2, float* __restrict__ dst2, int n2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[(long)i * 8];
  else if (i - n < n2) dst2[i - n] = src2[(long)(i - n) * 8];
}
int bias_pick(const float* src, float* dst, int n, const float* src2, float* dst2, int n2, hipStream_t st) {
  if (n <= 0 || n2 < 0) return MMSA_ERR_ARG;
  hipLaunchKernelGGL(bias_pick_kernel, dim3(cdiv(n + n2, 256)), dim3(256), 0, st, src, dst, n, src2, dst2, n2);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
// the same for up to BIAS_PICK_MAX deferred (src, dst, n) triples in one launch (a BERT backward's 24 grouped bias gradients)
__global__ __launch_bounds__(256) void bias_pick_batch_kernel(BiasPickBatch b) {
  const BiasPickJob j = b.job[blockIdx.y];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < j.n; i += gridDim.x * 256) j.dst[i] = j.src[(long)i * 8];
}
int bias_pick_batch(const BiasPickJob* jobs, int n, hipStream_t st) {
  for (int i = 0; i < n; i += BIAS_PICK_MAX) {
    BiasPickBatch b;
    const int m = n - i < BIAS_PICK_MAX ? n - i : BIAS_PICK_MAX;
    int nmax = 1;
    for (int k = 0; k < m; ++k) { b.job[k] = jobs[i + k]; nmax = jobs[i + k].n > nmax ? jobs[i + k].n : nmax; }
    hipLaunchKernelGGL(bias_pick_batch_kernel, dim3(cdiv(nmax, 256), m), dim3(256), 0, st, b);
    MMSA_CHECK_LAUNCH();
  }
  return MMSA_OK;
}
int fill_ones_bf16(void* dst, long n, hipStream_t st) {
  if (hipMemsetD16Async((hipDeviceptr_t)dst, 0x3F80, (size_t)n, st) != hipSuccess) return MMSA_ERR_LAUNCH;  // bf16 1.0
  return MMSA_OK;
}
