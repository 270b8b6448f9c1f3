// BERT self-attention (head_dim 64) for gfx950: softmax(Q K^T / 8 + mask) V, forward and backward.
//
// Input is the fused QKV projection output qkv[B*S][3*Hd] (Q | K | V column blocks, head h at columns h*64);
// output ctx[B*S][Hd]; backward consumes d_ctx and writes d_qkv in the same packed layout.
//
// MFMA kernels (bf16): one workgroup per (batch, head); a wave owns 16 query rows. Products are issued with
// swapped operands (D = X·Y with the "row" operand in the A slot) so that a lane always holds ONE query (or
// key) column and 4 consecutive rows of the other index — which makes (a) the softmax a per-lane loop plus two
// __shfl_xor steps across the 4 lanes that share a query, and (b) the probability tile directly usable as the
// next MFMA's B-slot operand with a permuted key order that the V/K operand matches through its
// ds_read_b64_tr_b16 row addresses (no LDS round trip for P in the forward, none for dS in dQ).
// K/V/Q/dO live in LDS as [S][64] images with a 16-byte XOR swizzle that serves both ds_read_b128 row reads and
// transposed reads; the backward keeps P and dS as [S][S] bf16 images for the key-owned dK / dV products.
//
// SIMT kernels (fp32 or bf16 storage): 4 lanes per query/key, each owning 16 of the 64 head dims. They are the
// exact-fp32 execution mode and the on-device cross-check of the MFMA kernels.
#include <stdlib.h>
#include "common.h"
#include "ops.h"

#define HD 64
#define MASK_NEG -3.0e38f

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// [rows][64] bf16 image, 128-byte rows, 16-byte chunk index XORed with (row & 7)
__device__ __forceinline__ int img_off(int row, int col) { return row * 128 + ((((col >> 3) ^ (row & 7)) << 4) | ((col & 7) << 1)); }
// [rows][S] bf16 image (P, dS): 32-byte chunk index XORed with f(row) masked to the row's chunk count
__device__ __forceinline__ int sq_off(int row, int col, int S, int swz_mask) {
  const int f = ((row & 3) | (((row >> 3) & 1) << 2)) & swz_mask;
  return row * (S * 2) + ((((col >> 4) ^ f) << 5) | ((col & 15) << 1));
}

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* base, int off0, int off1) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off1));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ bf16x8 pack_bf16x8(f32x4 a, f32x4 b) {
  bf16x8 r = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
  return r;
}

// cooperative copy of a [S][64] bf16 sub-matrix (row stride ld elements) into a swizzled LDS image
__device__ __forceinline__ void stage_rows(unsigned char* img, const bf16* src, long ld, int S, int tid, int nthreads) {
  for (int c = tid; c < S * 8; c += nthreads) {
    const int row = c >> 3, ch = c & 7;
    *(bf16x8*)(img + img_off(row, ch * 8)) = *(const bf16x8*)(src + (long)row * ld + ch * 8);
  }
}

#define MAX_TILES 16  // S <= 256 for the forward (scores of one query block live in registers)

// ------------------------------------------------------------------------------------------------ MFMA forward
// NW waves per workgroup: 4 (64 queries) in general; 8 when S is a multiple of 128 — at S = 128 that is ONE workgroup per
// (batch, head): K and V are staged once instead of twice, and with 60 registers per lane four 33 KiB workgroups fit a CU, so
// B * heads = 768 workgroups are one round on the chip where 1536 four-wave workgroups took a round and a half. Same per-wave
// code, bit-identical results; 21.3 -> 14.1 us at B = 64, S = 128 (47.2 -> 29.4 at B = 32, S = 256, 16 heads), tools/microbench/
// bench_attn.py. (The same idea for the backward — K, V and Q, dO sharing two LDS images so that three workgroups fit a CU — was
// measured and removed: 33.4 -> 49.4 us; the register-lean form it needs spills and the second staging is exposed.)
template <int NT, int NW = 4>  // NT = S / 16
__global__ __launch_bounds__(NW * 64, (NW == 8 && NT == 8) ? 6 : 1) void attn_fwd_mfma_kernel(const bf16* __restrict__ qkv, const float* __restrict__ mask,
                                                                bf16* __restrict__ ctx, int S, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Hd = heads * HD, ld = 3 * Hd;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  unsigned char* Kt = smem;
  unsigned char* Vt = smem + S * 128;
  float* mb = (float*)(smem + 2 * S * 128);
  const bf16* base = qkv + (long)b * S * ld + h * HD;
  stage_rows(Kt, base + Hd, ld, S, tid, NW * 64);
  stage_rows(Vt, base + 2 * Hd, ld, S, tid, NW * 64);
  for (int i = tid; i < S; i += NW * 64) mb[i] = (mask == nullptr || mask[(long)b * S + i] != 0.f) ? 0.f : 1.f;
  __syncthreads();
  const int q0 = (blockIdx.x * NW + wave) * 16;
  if (q0 >= S) return;  // no barriers below

  bf16x8 qf[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) qf[kk] = *(const bf16x8*)(base + (long)(q0 + r16) * ld + kk * 32 + 8 * g);

  f32x4 s[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    s[t] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 kf = *(const bf16x8*)(Kt + img_off(16 * t + r16, kk * 32 + 8 * g));
      s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[t], 0, 0, 0);
    }
  }
  // lane holds query q0+r16, keys 16t + 4g + r
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = mb[16 * t + 4 * g + r] != 0.f ? MASK_NEG : s[t][r] * scale;
      s[t][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __expf(s[t][r] - mx);
      s[t][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;

  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0, 0, 0, 0};
  const int qq = r16 >> 2, pp = r16 & 3;
#pragma unroll
  for (int ks = 0; ks < NT / 2; ++ks) {
    const bf16x8 pf = pack_bf16x8(s[2 * ks] * inv, s[2 * ks + 1] * inv);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int col = 16 * dt + 4 * pp;
      const bf16x8 vf = tr_pair(Vt, img_off(32 * ks + 4 * g + qq, col), img_off(32 * ks + 16 + 4 * g + qq, col));
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
    }
  }
  if constexpr (NT & 1) {  // odd tile count: last 16 keys with a zero upper half
    const int t = NT - 1;
    const bf16x8 pf = pack_bf16x8(s[t] * inv, f32x4{0, 0, 0, 0});
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int col = 16 * dt + 4 * pp;
      const int row = 16 * t + 4 * g + qq;
      const bf16x8 vf = tr_pair(Vt, img_off(row, col), img_off(row, col));
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
    }
  }
  bf16* orow = ctx + ((long)b * S + q0 + r16) * Hd + h * HD;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    bf16x4 v = {(bf16)o[dt][0], (bf16)o[dt][1], (bf16)o[dt][2], (bf16)o[dt][3]};
    *(bf16x4*)(orow + 16 * dt + 4 * g) = v;
  }
}

// ------------------------------------------------------------------------------------------------ MFMA backward
// 512 threads = 8 waves; wave w owns queries [16w, 16w+16) in phase 1 and keys [16w, 16w+16) in phase 2 (S <= 128;
// longer sequences: attn_bwd_mfma_rc_kernel below).
template <int NT>
__global__ __launch_bounds__(512) void attn_bwd_mfma_kernel(const bf16* __restrict__ qkv, const float* __restrict__ mask,
                                                            const bf16* __restrict__ dctx, bf16* __restrict__ dqkv, int heads,
                                                            float scale) {
  constexpr int S = NT * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Hd = heads * HD, ld = 3 * Hd;
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4, qq = r16 >> 2, pp = r16 & 3;
  unsigned char* Kt = smem;
  unsigned char* Vt = Kt + S * 128;
  unsigned char* Qt = Vt + S * 128;
  unsigned char* Dt = Qt + S * 128;
  unsigned char* Pt = Dt + S * 128;
  unsigned char* St = Pt + S * S * 2;
  float* mb = (float*)(St + S * S * 2);
  constexpr int swz_mask = (NT >= 8 ? 8 : NT) - 1;
  const bf16* base = qkv + (long)b * S * ld + h * HD;
  const bf16* dbase = dctx + (long)b * S * Hd + h * HD;
  stage_rows(Qt, base, ld, S, tid, 512);
  stage_rows(Kt, base + Hd, ld, S, tid, 512);
  stage_rows(Vt, base + 2 * Hd, ld, S, tid, 512);
  stage_rows(Dt, dbase, Hd, S, tid, 512);
  for (int i = tid; i < S; i += 512) mb[i] = (mask == nullptr || mask[(long)b * S + i] != 0.f) ? 0.f : 1.f;
  __syncthreads();

  const int q0 = wave * 16;
  if (q0 < S) {
    bf16x8 qf[2], df[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qf[kk] = *(const bf16x8*)(Qt + img_off(q0 + r16, kk * 32 + 8 * g));
      df[kk] = *(const bf16x8*)(Dt + img_off(q0 + r16, kk * 32 + 8 * g));
    }
    f32x4 s[NT], dp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s[t] = f32x4{0, 0, 0, 0};
      dp[t] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 kf = *(const bf16x8*)(Kt + img_off(16 * t + r16, kk * 32 + 8 * g));
        const bf16x8 vf = *(const bf16x8*)(Vt + img_off(16 * t + r16, kk * 32 + 8 * g));
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[t], 0, 0, 0);
        dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, df[kk], dp[t], 0, 0, 0);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = mb[16 * t + 4 * g + r] != 0.f ? MASK_NEG : s[t][r] * scale;
        s[t][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[t][r] - mx);
        s[t][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    float delta = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // the forward multiplies V by the bf16-rounded probability: use the same value here
        s[t][r] = (float)(bf16)(s[t][r] * inv);
        delta += s[t][r] * dp[t][r];
      }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[r] = s[t][r] * (dp[t][r] - delta) * scale;
      const bf16x4 pb = {(bf16)s[t][0], (bf16)s[t][1], (bf16)s[t][2], (bf16)s[t][3]};
      const bf16x4 db = {(bf16)ds[0], (bf16)ds[1], (bf16)ds[2], (bf16)ds[3]};
      *(bf16x4*)(Pt + sq_off(q0 + r16, 16 * t + 4 * g, S, swz_mask)) = pb;
      *(bf16x4*)(St + sq_off(q0 + r16, 16 * t + 4 * g, S, swz_mask)) = db;
      dp[t] = ds;  // keep dS (fp32) for dQ
    }
    // dQ[q][d] = sum_key dS[q][key] K[key][d]
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < NT / 2; ++ks) {
      const bf16x8 pf = pack_bf16x8(dp[2 * ks], dp[2 * ks + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int col = 16 * dt + 4 * pp;
        const bf16x8 kf = tr_pair(Kt, img_off(32 * ks + 4 * g + qq, col), img_off(32 * ks + 16 + 4 * g + qq, col));
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, pf, o[dt], 0, 0, 0);
      }
    }
    if constexpr (NT & 1) {
      const int t = NT - 1;
      const bf16x8 pf = pack_bf16x8(dp[t], f32x4{0, 0, 0, 0});
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int col = 16 * dt + 4 * pp;
        const int row = 16 * t + 4 * g + qq;
        const bf16x8 kf = tr_pair(Kt, img_off(row, col), img_off(row, col));
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, pf, o[dt], 0, 0, 0);
      }
    }
    bf16* orow = dqkv + ((long)b * S + q0 + r16) * ld + h * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 v = {(bf16)o[dt][0], (bf16)o[dt][1], (bf16)o[dt][2], (bf16)o[dt][3]};
      *(bf16x4*)(orow + 16 * dt + 4 * g) = v;
    }
  }
  __syncthreads();
  // phase 2: wave owns keys k0..k0+15: dV[key][d] = sum_q P[q][key] dO[q][d]; dK[key][d] = sum_q dS[q][key] Q[q][d]
  const int k0 = wave * 16;
  if (k0 < S) {
    f32x4 dv[4], dk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dv[dt] = f32x4{0, 0, 0, 0}; dk[dt] = f32x4{0, 0, 0, 0}; }
    constexpr int QS = (S + 31) / 32;
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
      // k index = query 32qs + 8g + j ; rows past S (odd tile count) are clamped and their P/dS zeroed
      int row0 = 32 * qs + 8 * g + qq, row1 = row0 + 4;
      const bool valid = (32 * qs + 8 * g) < S;
      if (!valid) { row0 = qq; row1 = qq + 4; }
      const int kcol = k0 + 4 * pp;
      bf16x8 pf = tr_pair(Pt, sq_off(row0, kcol, S, swz_mask), sq_off(row1, kcol, S, swz_mask));
      bf16x8 sf = tr_pair(St, sq_off(row0, kcol, S, swz_mask), sq_off(row1, kcol, S, swz_mask));
      if (!valid) {
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        pf = z; sf = z;
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int col = 16 * dt + 4 * pp;
        const bf16x8 dof = tr_pair(Dt, img_off(row0, col), img_off(row1, col));
        const bf16x8 qf = tr_pair(Qt, img_off(row0, col), img_off(row1, col));
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, pf, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, sf, dk[dt], 0, 0, 0);
      }
    }
    bf16* krow = dqkv + ((long)b * S + k0 + r16) * ld + Hd + h * HD;
    bf16* vrow = krow + Hd;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 kv = {(bf16)dk[dt][0], (bf16)dk[dt][1], (bf16)dk[dt][2], (bf16)dk[dt][3]};
      bf16x4 vv = {(bf16)dv[dt][0], (bf16)dv[dt][1], (bf16)dv[dt][2], (bf16)dv[dt][3]};
      *(bf16x4*)(krow + 16 * dt + 4 * g) = kv;
      *(bf16x4*)(vrow + 16 * dt + 4 * g) = vv;
    }
  }
}

// ------------------------------------------------------------------------------------------------ MFMA backward, S <= 256
// The kernel above keeps P and dS as two [S][S] bf16 LDS images for its key-owned phase: 2 * S * S * 2 bytes = 256 KiB at
// S = 256 (BERT-large, BASELINE.json configs[3]) — more than the CU has. This variant keeps NO S x S image: phase 1 (a wave
// owns 16 queries at a time: full-row scores in registers, softmax statistics, dQ) leaves three floats per query in LDS — row
// maximum, 1 / row sum, delta = sum_key P dP — and phase 2 (a wave owns 16 keys at a time) RECOMPUTES the transposed score and
// dP tiles two query tiles at a time from the Q / dO / K / V images (rows = queries, lane = key column), turns them into P^T
// and dS^T with those statistics, and feeds them straight into the dV / dK products as B-slot operands (the accumulator-as-
// operand order: element j of lane group g is query 32 ks + 4 g + j (j < 4) or 32 ks + 16 + 4 g + j - 4, matched by the
// transposed dO / Q reads, exactly as the forward does for P V). Cost: the two S x S x 64 products are done twice (+50 % of
// this kernel's MFMAs, attention being 2.7 % of the model's); LDS: four [S][64] images = 128 KiB at S = 256.
// 512 threads = 8 waves; NT = S / 16 (even) query / key tiles are dealt round-robin to the waves.
template <int NT>
__global__ __launch_bounds__(512) void attn_bwd_mfma_rc_kernel(const bf16* __restrict__ qkv, const float* __restrict__ mask,
                                                               const bf16* __restrict__ dctx, bf16* __restrict__ dqkv, int heads,
                                                               float scale) {
  constexpr int S = NT * 16;
  static_assert(NT % 2 == 0 && NT <= MAX_TILES, "even tile count, S <= 256");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Hd = heads * HD, ld = 3 * Hd;
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4, qq = r16 >> 2, pp = r16 & 3;
  unsigned char* Kt = smem;
  unsigned char* Vt = Kt + S * 128;
  unsigned char* Qt = Vt + S * 128;
  unsigned char* Dt = Qt + S * 128;
  float* mb = (float*)(Dt + S * 128);
  float* rmx = mb + S;     // row maximum of the scaled, masked scores
  float* rinv = rmx + S;   // 1 / sum exp(score - max)
  float* rdel = rinv + S;  // delta = sum_key P dP
  const bf16* base = qkv + (long)b * S * ld + h * HD;
  const bf16* dbase = dctx + (long)b * S * Hd + h * HD;
  stage_rows(Qt, base, ld, S, tid, 512);
  stage_rows(Kt, base + Hd, ld, S, tid, 512);
  stage_rows(Vt, base + 2 * Hd, ld, S, tid, 512);
  stage_rows(Dt, dbase, Hd, S, tid, 512);
  for (int i = tid; i < S; i += 512) mb[i] = (mask == nullptr || mask[(long)b * S + i] != 0.f) ? 0.f : 1.f;
  __syncthreads();

  // ---- phase 1: query-owned. lane: query q0 + r16, keys 16 t + 4 g + r
  for (int qt = wave; qt < NT; qt += 8) {
    const int q0 = qt * 16;
    bf16x8 qf[2], df[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qf[kk] = *(const bf16x8*)(Qt + img_off(q0 + r16, kk * 32 + 8 * g));
      df[kk] = *(const bf16x8*)(Dt + img_off(q0 + r16, kk * 32 + 8 * g));
    }
    f32x4 s[NT], dp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s[t] = f32x4{0, 0, 0, 0};
      dp[t] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 kf = *(const bf16x8*)(Kt + img_off(16 * t + r16, kk * 32 + 8 * g));
        const bf16x8 vf = *(const bf16x8*)(Vt + img_off(16 * t + r16, kk * 32 + 8 * g));
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[t], 0, 0, 0);
        dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, df[kk], dp[t], 0, 0, 0);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = mb[16 * t + 4 * g + r] != 0.f ? MASK_NEG : s[t][r] * scale;
        s[t][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[t][r] - mx);
        s[t][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    float delta = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[t][r] = (float)(bf16)(s[t][r] * inv);  // the forward multiplies V by the bf16-rounded probability
        delta += s[t][r] * dp[t][r];
      }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    if (g == 0) { rmx[q0 + r16] = mx; rinv[q0 + r16] = inv; rdel[q0 + r16] = delta; }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dp[t][r] = s[t][r] * (dp[t][r] - delta) * scale;  // dS (fp32)
    // dQ[q][d] = sum_key dS[q][key] K[key][d]
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < NT / 2; ++ks) {
      const bf16x8 pf = pack_bf16x8(dp[2 * ks], dp[2 * ks + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int col = 16 * dt + 4 * pp;
        const bf16x8 kf = tr_pair(Kt, img_off(32 * ks + 4 * g + qq, col), img_off(32 * ks + 16 + 4 * g + qq, col));
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, pf, o[dt], 0, 0, 0);
      }
    }
    bf16* orow = dqkv + ((long)b * S + q0 + r16) * ld + h * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 v = {(bf16)o[dt][0], (bf16)o[dt][1], (bf16)o[dt][2], (bf16)o[dt][3]};
      *(bf16x4*)(orow + 16 * dt + 4 * g) = v;
    }
  }
  __syncthreads();

  // ---- phase 2: key-owned. lane: key k0 + r16, queries 16 t + 4 g + r (rows of the recomputed transposed tiles)
  for (int kt = wave; kt < NT; kt += 8) {
    const int k0 = kt * 16;
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      kf[kk] = *(const bf16x8*)(Kt + img_off(k0 + r16, kk * 32 + 8 * g));
      vf[kk] = *(const bf16x8*)(Vt + img_off(k0 + r16, kk * 32 + 8 * g));
    }
    const bool masked = mb[k0 + r16] != 0.f;
    f32x4 dv[4], dk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dv[dt] = f32x4{0, 0, 0, 0}; dk[dt] = f32x4{0, 0, 0, 0}; }
#pragma unroll 2
    for (int ks = 0; ks < NT / 2; ++ks) {
      f32x4 pT[2], sT[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * ks + u;
        f32x4 sc = {0, 0, 0, 0}, dpv = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const bf16x8 qf = *(const bf16x8*)(Qt + img_off(16 * t + r16, kk * 32 + 8 * g));
          const bf16x8 df = *(const bf16x8*)(Dt + img_off(16 * t + r16, kk * 32 + 8 * g));
          sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[kk], sc, 0, 0, 0);    // [query 16t + 4g + r][key k0 + r16]
          dpv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vf[kk], dpv, 0, 0, 0);  // dP = dO V^T, same layout
        }
        const f32x4 mx4 = *(const f32x4*)(rmx + 16 * t + 4 * g);
        const f32x4 in4 = *(const f32x4*)(rinv + 16 * t + 4 * g);
        const f32x4 de4 = *(const f32x4*)(rdel + 16 * t + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = masked ? MASK_NEG : sc[r] * scale;
          const float p = (float)(bf16)(__expf(v - mx4[r]) * in4[r]);
          pT[u][r] = p;
          sT[u][r] = p * (dpv[r] - de4[r]) * scale;
        }
      }
      const bf16x8 pf = pack_bf16x8(pT[0], pT[1]);
      const bf16x8 sf = pack_bf16x8(sT[0], sT[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int col = 16 * dt + 4 * pp;
        const bf16x8 dof = tr_pair(Dt, img_off(32 * ks + 4 * g + qq, col), img_off(32 * ks + 16 + 4 * g + qq, col));
        const bf16x8 qtf = tr_pair(Qt, img_off(32 * ks + 4 * g + qq, col), img_off(32 * ks + 16 + 4 * g + qq, col));
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, pf, dv[dt], 0, 0, 0);  // dV[key][d] += sum_q P[q][key] dO[q][d]
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, sf, dk[dt], 0, 0, 0);  // dK[key][d] += sum_q dS[q][key] Q[q][d]
      }
    }
    bf16* krow = dqkv + ((long)b * S + k0 + r16) * ld + Hd + h * HD;
    bf16* vrow = krow + Hd;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 kv = {(bf16)dk[dt][0], (bf16)dk[dt][1], (bf16)dk[dt][2], (bf16)dk[dt][3]};
      bf16x4 vv = {(bf16)dv[dt][0], (bf16)dv[dt][1], (bf16)dv[dt][2], (bf16)dv[dt][3]};
      *(bf16x4*)(krow + 16 * dt + 4 * g) = kv;
      *(bf16x4*)(vrow + 16 * dt + 4 * g) = vv;
    }
  }
}

// ------------------------------------------------------------------------------------------------ fp32 on the matrix cores
// The exact mode's attention (fp32 storage, S <= 128, S % 16 == 0) on v_mfma_f32_16x16x4_f32 — an exact-fp32 fmaf chain per
// output, like the fp32 GEMM — instead of the VALU kernels below (27 of the exact step's 125 ms at B = 64). Same structure as
// the bf16 kernels: one workgroup per (batch, head), 4 waves; a wave owns 16 queries at a time with the whole score row in
// registers (lane = one query column, 4 consecutive keys per accumulator register group); products are issued with the
// "row" operand in the A slot so the probability tile is directly the next product's B-slot operand: for the k step that
// takes register r, lane group g supplies key 16 t + 4 g + r, and the V / K / dO / Q operand reads exactly that row.
// The backward is the recompute form (attn_bwd_mfma_rc_kernel): per-query max / 1/sum / delta in LDS, key-owned phase 2
// recomputes the transposed tiles. LDS images are [S][66] floats (stride 66: the 16-rows x 4-columns operand reads are
// conflict-free). P is NOT rounded here (fp32 storage: the forward multiplies V by the fp32 probability).
#define FA_LD 66
template <int NT>
__global__ __launch_bounds__(256) void attn_fwd_f32_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ mask,
                                                                float* __restrict__ ctx, int heads, float scale) {
  constexpr int S = NT * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Kt = (float*)smem;
  float* Vt = Kt + S * FA_LD;
  float* mb = Vt + S * FA_LD;
  const int Hd = heads * HD, ld = 3 * Hd;
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const float* base = qkv + (long)b * S * ld + h * HD;
  for (int c = tid; c < S * 16; c += 256) {  // float4 chunks of the K and V rows
    const int row = c >> 4, c4 = (c & 15) * 4;
    const f32x4 kv = *(const f32x4*)(base + (long)row * ld + Hd + c4), vv = *(const f32x4*)(base + (long)row * ld + 2 * Hd + c4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { Kt[row * FA_LD + c4 + e] = kv[e]; Vt[row * FA_LD + c4 + e] = vv[e]; }
  }
  for (int i = tid; i < S; i += 256) mb[i] = (mask == nullptr || mask[(long)b * S + i] != 0.f) ? 0.f : 1.f;
  __syncthreads();
  for (int qt = wave; qt < NT; qt += 4) {
    const int q0 = qt * 16;
    float qf[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) qf[ks] = base[(long)(q0 + r16) * ld + 4 * ks + g];
    f32x4 s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s[t] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Kt[(16 * t + r16) * FA_LD + 4 * ks + g], qf[ks], s[t], 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = mb[16 * t + 4 * g + r] != 0.f ? MASK_NEG : s[t][r] * scale;
        s[t][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[t][r] - mx);
        s[t][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = s[t][r] * inv;
        const float* vrow = Vt + (16 * t + 4 * g + r) * FA_LD + r16;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[16 * dt], pv, o[dt], 0, 0, 0);
      }
    float* orow = ctx + ((long)b * S + q0 + r16) * Hd + h * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) *(f32x4*)(orow + 16 * dt + 4 * g) = o[dt];
  }
}

template <int NT>
__global__ __launch_bounds__(256) void attn_bwd_f32_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ mask,
                                                                const float* __restrict__ dctx, float* __restrict__ dqkv, int heads,
                                                                float scale) {
  constexpr int S = NT * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Kt = (float*)smem;
  float* Vt = Kt + S * FA_LD;
  float* Qt = Vt + S * FA_LD;
  float* Dt = Qt + S * FA_LD;
  float* mb = Dt + S * FA_LD;
  float* rmx = mb + S;
  float* rinv = rmx + S;
  float* rdel = rinv + S;
  const int Hd = heads * HD, ld = 3 * Hd;
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const float* base = qkv + (long)b * S * ld + h * HD;
  const float* dbase = dctx + (long)b * S * Hd + h * HD;
  for (int c = tid; c < S * 16; c += 256) {
    const int row = c >> 4, c4 = (c & 15) * 4;
    const f32x4 qv = *(const f32x4*)(base + (long)row * ld + c4), kv = *(const f32x4*)(base + (long)row * ld + Hd + c4);
    const f32x4 vv = *(const f32x4*)(base + (long)row * ld + 2 * Hd + c4), dv = *(const f32x4*)(dbase + (long)row * Hd + c4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      Qt[row * FA_LD + c4 + e] = qv[e]; Kt[row * FA_LD + c4 + e] = kv[e];
      Vt[row * FA_LD + c4 + e] = vv[e]; Dt[row * FA_LD + c4 + e] = dv[e];
    }
  }
  for (int i = tid; i < S; i += 256) mb[i] = (mask == nullptr || mask[(long)b * S + i] != 0.f) ? 0.f : 1.f;
  __syncthreads();

  // ---- phase 1: query-owned. lane: query q0 + r16, keys 16 t + 4 g + r
  for (int qt = wave; qt < NT; qt += 4) {
    const int q0 = qt * 16;
    f32x4 s[NT], dp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s[t] = f32x4{0, 0, 0, 0};
      dp[t] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int o = 4 * ks + g;
        s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Kt[(16 * t + r16) * FA_LD + o], Qt[(q0 + r16) * FA_LD + o], s[t], 0, 0, 0);
        dp[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vt[(16 * t + r16) * FA_LD + o], Dt[(q0 + r16) * FA_LD + o], dp[t], 0, 0, 0);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = mb[16 * t + 4 * g + r] != 0.f ? MASK_NEG : s[t][r] * scale;
        s[t][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[t][r] - mx);
        s[t][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    float delta = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[t][r] *= inv;
        delta += s[t][r] * dp[t][r];
      }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    if (g == 0) { rmx[q0 + r16] = mx; rinv[q0 + r16] = inv; rdel[q0 + r16] = delta; }
    // dQ[q][d] = sum_key dS[q][key] K[key][d]
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ds = s[t][r] * (dp[t][r] - delta) * scale;
        const float* krow = Kt + (16 * t + 4 * g + r) * FA_LD + r16;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(krow[16 * dt], ds, o[dt], 0, 0, 0);
      }
    float* orow = dqkv + ((long)b * S + q0 + r16) * ld + h * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) *(f32x4*)(orow + 16 * dt + 4 * g) = o[dt];
  }
  __syncthreads();

  // ---- phase 2: key-owned. lane: key k0 + r16, queries 16 t + 4 g + r (recomputed transposed tiles)
  for (int kt = wave; kt < NT; kt += 4) {
    const int k0 = kt * 16;
    float kf[16], vf[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) { kf[ks] = Kt[(k0 + r16) * FA_LD + 4 * ks + g]; vf[ks] = Vt[(k0 + r16) * FA_LD + 4 * ks + g]; }
    const bool masked = mb[k0 + r16] != 0.f;
    f32x4 dv[4], dk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dv[dt] = f32x4{0, 0, 0, 0}; dk[dt] = f32x4{0, 0, 0, 0}; }
#pragma unroll 2
    for (int t = 0; t < NT; ++t) {
      f32x4 sc = {0, 0, 0, 0}, dpv = {0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int o = 4 * ks + g;
        sc = __builtin_amdgcn_mfma_f32_16x16x4f32(Qt[(16 * t + r16) * FA_LD + o], kf[ks], sc, 0, 0, 0);    // [query 16t+4g+r][key k0+r16]
        dpv = __builtin_amdgcn_mfma_f32_16x16x4f32(Dt[(16 * t + r16) * FA_LD + o], vf[ks], dpv, 0, 0, 0);  // dP, same layout
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 16 * t + 4 * g + r;
        const float v = masked ? MASK_NEG : sc[r] * scale;
        const float pr = __expf(v - rmx[q]) * rinv[q];
        const float ds = pr * (dpv[r] - rdel[q]) * scale;
        const float* drow = Dt + q * FA_LD + r16;
        const float* qrow = Qt + q * FA_LD + r16;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(drow[16 * dt], pr, dv[dt], 0, 0, 0);  // dV[key][d] += P[q][key] dO[q][d]
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qrow[16 * dt], ds, dk[dt], 0, 0, 0);  // dK[key][d] += dS[q][key] Q[q][d]
        }
      }
    }
    float* krow = dqkv + ((long)b * S + k0 + r16) * ld + Hd + h * HD;
    float* vrow = krow + Hd;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      *(f32x4*)(krow + 16 * dt + 4 * g) = dk[dt];
      *(f32x4*)(vrow + 16 * dt + 4 * g) = dv[dt];
    }
  }
}

// ------------------------------------------------------------------------------------------------ SIMT kernels
// 4 lanes per row (query or key); lane `sub` owns head dims [16 sub, 16 sub + 16).
template <typename T>
__device__ __forceinline__ void load16(const T* p, float* v) {
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = to_f32<T>(p[i]);
}
__device__ __forceinline__ float dot16_quad(const float* a, const float* b) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s = fmaf(a[i], b[i], s);
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  return s;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_simt_kernel(const T* __restrict__ qkv, const float* __restrict__ mask,
                                                            T* __restrict__ ctx, float* __restrict__ stats, int S, int heads,
                                                            float scale) {
  const int Hd = heads * HD, ld = 3 * Hd;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int q = blockIdx.x * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
  const bool active = q < S;
  const int qc = active ? q : S - 1;
  const T* base = qkv + (long)b * S * ld + h * HD + sub * 16;
  float qv[16], kv[16], o[16];
  load16<T>(base + (long)qc * ld, qv);
  float mx = -INFINITY;
  for (int k = 0; k < S; ++k) {
    load16<T>(base + Hd + (long)k * ld, kv);
    float s = dot16_quad(qv, kv) * scale;
    if (mask && mask[(long)b * S + k] == 0.f) s = MASK_NEG;
    mx = fmaxf(mx, s);
  }
  float sum = 0.f;
  for (int k = 0; k < S; ++k) {
    load16<T>(base + Hd + (long)k * ld, kv);
    float s = dot16_quad(qv, kv) * scale;
    if (mask && mask[(long)b * S + k] == 0.f) s = MASK_NEG;
    sum += __expf(s - mx);
  }
  const float inv = 1.f / sum;
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = 0.f;
  for (int k = 0; k < S; ++k) {
    load16<T>(base + Hd + (long)k * ld, kv);
    float s = dot16_quad(qv, kv) * scale;
    if (mask && mask[(long)b * S + k] == 0.f) s = MASK_NEG;
    const float p = q_f32<T>(__expf(s - mx) * inv);
    load16<T>(base + 2 * Hd + (long)k * ld, kv);
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = fmaf(p, kv[i], o[i]);
  }
  if (active) {
    T* orow = ctx + ((long)b * S + q) * Hd + h * HD + sub * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) orow[i] = from_f32<T>(o[i]);
    if (stats && sub == 0) {
      stats[((long)bh * S + q) * 2 + 0] = mx;
      stats[((long)bh * S + q) * 2 + 1] = inv;
    }
  }
}

// pass A (per query): softmax stats, delta = sum_k p dp, and dQ. stats[bh][q] = {max, 1/sum, delta}
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_q_simt_kernel(const T* __restrict__ qkv, const float* __restrict__ mask,
                                                              const T* __restrict__ dctx, T* __restrict__ dqkv,
                                                              float* __restrict__ stats, int S, int heads, float scale) {
  const int Hd = heads * HD, ld = 3 * Hd;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int q = blockIdx.x * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
  const bool active = q < S;
  const int qc = active ? q : S - 1;
  const T* base = qkv + (long)b * S * ld + h * HD + sub * 16;
  float qv[16], dov[16], kv[16], vv[16], dq[16];
  load16<T>(base + (long)qc * ld, qv);
  load16<T>(dctx + ((long)b * S + qc) * Hd + h * HD + sub * 16, dov);
  float mx = -INFINITY;
  for (int k = 0; k < S; ++k) {
    load16<T>(base + Hd + (long)k * ld, kv);
    float s = dot16_quad(qv, kv) * scale;
    if (mask && mask[(long)b * S + k] == 0.f) s = MASK_NEG;
    mx = fmaxf(mx, s);
  }
  float sum = 0.f;
  for (int k = 0; k < S; ++k) {
    load16<T>(base + Hd + (long)k * ld, kv);
    float s = dot16_quad(qv, kv) * scale;
    if (mask && mask[(long)b * S + k] == 0.f) s = MASK_NEG;
    sum += __expf(s - mx);
  }
  const float inv = 1.f / sum;
  float delta = 0.f;
  for (int k = 0; k < S; ++k) {
    load16<T>(base + Hd + (long)k * ld, kv);
    load16<T>(base + 2 * Hd + (long)k * ld, vv);
    float s = dot16_quad(qv, kv) * scale;
    if (mask && mask[(long)b * S + k] == 0.f) s = MASK_NEG;
    const float p = q_f32<T>(__expf(s - mx) * inv);
    delta += p * dot16_quad(dov, vv);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) dq[i] = 0.f;
  for (int k = 0; k < S; ++k) {
    load16<T>(base + Hd + (long)k * ld, kv);
    load16<T>(base + 2 * Hd + (long)k * ld, vv);
    float s = dot16_quad(qv, kv) * scale;
    if (mask && mask[(long)b * S + k] == 0.f) s = MASK_NEG;
    const float p = q_f32<T>(__expf(s - mx) * inv);
    const float ds = q_f32<T>(p * (dot16_quad(dov, vv) - delta) * scale);
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[i] = fmaf(ds, kv[i], dq[i]);
  }
  if (active) {
    T* orow = dqkv + ((long)b * S + q) * ld + h * HD + sub * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) orow[i] = from_f32<T>(dq[i]);
    if (sub == 0) {
      float* st = stats + ((long)bh * S + q) * 3;
      st[0] = mx; st[1] = inv; st[2] = delta;
    }
  }
}

// pass B (per key): dK, dV
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kv_simt_kernel(const T* __restrict__ qkv, const float* __restrict__ mask,
                                                               const T* __restrict__ dctx, T* __restrict__ dqkv,
                                                               const float* __restrict__ stats, int S, int heads,
                                                               float scale) {
  const int Hd = heads * HD, ld = 3 * Hd;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int k = blockIdx.x * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
  const bool active = k < S;
  const int kc = active ? k : S - 1;
  const T* base = qkv + (long)b * S * ld + h * HD + sub * 16;
  float kv[16], vv[16], qv[16], dov[16], dk[16], dv[16];
  load16<T>(base + Hd + (long)kc * ld, kv);
  load16<T>(base + 2 * Hd + (long)kc * ld, vv);
  const bool masked = mask && mask[(long)b * S + kc] == 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { dk[i] = 0.f; dv[i] = 0.f; }
  for (int q = 0; q < S; ++q) {
    load16<T>(base + (long)q * ld, qv);
    load16<T>(dctx + ((long)b * S + q) * Hd + h * HD + sub * 16, dov);
    const float* st = stats + ((long)bh * S + q) * 3;
    float s = dot16_quad(qv, kv) * scale;
    if (masked) s = MASK_NEG;
    const float p = q_f32<T>(__expf(s - st[0]) * st[1]);
    const float ds = q_f32<T>(p * (dot16_quad(dov, vv) - st[2]) * scale);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      dv[i] = fmaf(p, dov[i], dv[i]);
      dk[i] = fmaf(ds, qv[i], dk[i]);
    }
  }
  if (active) {
    T* krow = dqkv + ((long)b * S + k) * ld + Hd + h * HD + sub * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      krow[i] = from_f32<T>(dk[i]);
      krow[Hd + i] = from_f32<T>(dv[i]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ launchers
size_t attention_bwd_ws_bytes(int B, int S, int heads) { return (size_t)B * heads * S * 3 * sizeof(float); }

template <typename T>
static int attn_fwd_simt(const void* qkv, const float* mask, void* ctx, int B, int S, int heads, hipStream_t st) {
  dim3 grid(cdiv(S, 64), B * heads);
  hipLaunchKernelGGL(attn_fwd_simt_kernel<T>, grid, dim3(256), 0, st, (const T*)qkv, mask, (T*)ctx, (float*)nullptr, S,
                     heads, 0.125f);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
template <typename T>
static int attn_bwd_simt(const void* qkv, const float* mask, const void* dctx, void* dqkv, float* ws, int B, int S,
                         int heads, hipStream_t st) {
  dim3 grid(cdiv(S, 64), B * heads);
  hipLaunchKernelGGL(attn_bwd_q_simt_kernel<T>, grid, dim3(256), 0, st, (const T*)qkv, mask, (const T*)dctx, (T*)dqkv, ws,
                     S, heads, 0.125f);
  hipLaunchKernelGGL(attn_bwd_kv_simt_kernel<T>, grid, dim3(256), 0, st, (const T*)qkv, mask, (const T*)dctx, (T*)dqkv,
                     (const float*)ws, S, heads, 0.125f);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

template <int NT>
static int attn_fwd_mfma_nt(const void* qkv, const float* mask, void* ctx, int B, int S, int heads, hipStream_t st) {
  const size_t lds = (size_t)2 * S * 128 + S * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if constexpr (NT % 8 == 0) {
    if (!mmsa_disabled("attn_fwd8")) {
      static bool attr8 = false;
      if (!attr8) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<NT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr8 = true;
      }
      hipLaunchKernelGGL((attn_fwd_mfma_kernel<NT, 8>), dim3(S / 128, B * heads), dim3(512), lds, st, (const bf16*)qkv, mask,
                         (bf16*)ctx, S, heads, 0.125f);
      MMSA_CHECK_LAUNCH();
      return MMSA_OK;
    }
  }
  dim3 grid(cdiv(S, 64), B * heads);
  hipLaunchKernelGGL(attn_fwd_mfma_kernel<NT>, grid, dim3(256), lds, st, (const bf16*)qkv, mask, (bf16*)ctx, S, heads,
                     0.125f);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
template <int NT>
static int attn_bwd_mfma_nt(const void* qkv, const float* mask, const void* dctx, void* dqkv, int B, int heads,
                            hipStream_t st) {
  constexpr int S = NT * 16;
  const size_t lds = (size_t)4 * S * 128 + (size_t)2 * S * S * 2 + S * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(attn_bwd_mfma_kernel<NT>, dim3(B * heads), dim3(512), lds, st, (const bf16*)qkv, mask,
                     (const bf16*)dctx, (bf16*)dqkv, heads, 0.125f);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

template <int NT>
static int attn_bwd_mfma_rc_nt(const void* qkv, const float* mask, const void* dctx, void* dqkv, int B, int heads,
                               hipStream_t st) {
  constexpr int S = NT * 16;
  const size_t lds = (size_t)4 * S * 128 + (size_t)4 * S * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_mfma_rc_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(attn_bwd_mfma_rc_kernel<NT>, dim3(B * heads), dim3(512), lds, st, (const bf16*)qkv, mask,
                     (const bf16*)dctx, (bf16*)dqkv, heads, 0.125f);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

static bool attn_f32_mfma_on() {
  static const bool off = [] { const char* v = getenv("MMSA_F32_SIMT"); return v && atoi(v) != 0; }();
  return !off;
}
template <int NT>
static int attn_fwd_f32_nt(const void* qkv, const float* mask, void* ctx, int B, int heads, hipStream_t st) {
  constexpr int S = NT * 16;
  const size_t lds = (size_t)2 * S * FA_LD * 4 + S * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_f32_mfma_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(attn_fwd_f32_mfma_kernel<NT>, dim3(B * heads), dim3(256), lds, st, (const float*)qkv, mask, (float*)ctx, heads,
                     0.125f);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
template <int NT>
static int attn_bwd_f32_nt(const void* qkv, const float* mask, const void* dctx, void* dqkv, int B, int heads, hipStream_t st) {
  constexpr int S = NT * 16;
  const size_t lds = (size_t)4 * S * FA_LD * 4 + (size_t)4 * S * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_f32_mfma_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(attn_bwd_f32_mfma_kernel<NT>, dim3(B * heads), dim3(256), lds, st, (const float*)qkv, mask, (const float*)dctx,
                     (float*)dqkv, heads, 0.125f);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// impl: 0 = fp32 storage (SIMT), 1 = bf16 storage MFMA, 2 = bf16 storage SIMT (cross-check)
int attention_fwd(int impl, const void* qkv, const float* mask, void* ctx, int B, int S, int heads, int head_dim,
                  hipStream_t st) {
  if (head_dim != HD || S < 1) return MMSA_ERR_UNSUPPORTED;
  if (impl == 0) {  // fp32 storage: the fp32-MFMA kernel for S = 16 .. 128 in steps of 16, else (or MMSA_F32_SIMT=1) the VALU one
    if (attn_f32_mfma_on() && S % 16 == 0 && S <= 128) switch (S / 16) {
      case 1: return attn_fwd_f32_nt<1>(qkv, mask, ctx, B, heads, st);
      case 2: return attn_fwd_f32_nt<2>(qkv, mask, ctx, B, heads, st);
      case 3: return attn_fwd_f32_nt<3>(qkv, mask, ctx, B, heads, st);
      case 4: return attn_fwd_f32_nt<4>(qkv, mask, ctx, B, heads, st);
      case 6: return attn_fwd_f32_nt<6>(qkv, mask, ctx, B, heads, st);
      case 8: return attn_fwd_f32_nt<8>(qkv, mask, ctx, B, heads, st);
      default: break;
    }
    return attn_fwd_simt<float>(qkv, mask, ctx, B, S, heads, st);
  }
  if (impl == 2) return attn_fwd_simt<bf16>(qkv, mask, ctx, B, S, heads, st);
  if (S % 16 || S > 256) return attn_fwd_simt<bf16>(qkv, mask, ctx, B, S, heads, st);
  switch (S / 16) {
    case 1: return attn_fwd_mfma_nt<1>(qkv, mask, ctx, B, S, heads, st);
    case 2: return attn_fwd_mfma_nt<2>(qkv, mask, ctx, B, S, heads, st);
    case 3: return attn_fwd_mfma_nt<3>(qkv, mask, ctx, B, S, heads, st);
    case 4: return attn_fwd_mfma_nt<4>(qkv, mask, ctx, B, S, heads, st);
    case 6: return attn_fwd_mfma_nt<6>(qkv, mask, ctx, B, S, heads, st);
    case 8: return attn_fwd_mfma_nt<8>(qkv, mask, ctx, B, S, heads, st);
    case 12: return attn_fwd_mfma_nt<12>(qkv, mask, ctx, B, S, heads, st);
    case 16: return attn_fwd_mfma_nt<16>(qkv, mask, ctx, B, S, heads, st);
    default: return attn_fwd_simt<bf16>(qkv, mask, ctx, B, S, heads, st);
  }
}

int attention_bwd(int impl, const void* qkv, const float* mask, const void* dctx, void* dqkv, float* ws, int B, int S,
                  int heads, int head_dim, hipStream_t st) {
  if (head_dim != HD || S < 1) return MMSA_ERR_UNSUPPORTED;
  if (impl == 0) {
    if (attn_f32_mfma_on() && S % 16 == 0 && S <= 128) switch (S / 16) {
      case 1: return attn_bwd_f32_nt<1>(qkv, mask, dctx, dqkv, B, heads, st);
      case 2: return attn_bwd_f32_nt<2>(qkv, mask, dctx, dqkv, B, heads, st);
      case 3: return attn_bwd_f32_nt<3>(qkv, mask, dctx, dqkv, B, heads, st);
      case 4: return attn_bwd_f32_nt<4>(qkv, mask, dctx, dqkv, B, heads, st);
      case 6: return attn_bwd_f32_nt<6>(qkv, mask, dctx, dqkv, B, heads, st);
      case 8: return attn_bwd_f32_nt<8>(qkv, mask, dctx, dqkv, B, heads, st);
      default: break;
    }
    return attn_bwd_simt<float>(qkv, mask, dctx, dqkv, ws, B, S, heads, st);
  }
  if (impl == 2) return attn_bwd_simt<bf16>(qkv, mask, dctx, dqkv, ws, B, S, heads, st);
  switch (S) {
    case 16: return attn_bwd_mfma_nt<1>(qkv, mask, dctx, dqkv, B, heads, st);
    case 32: return attn_bwd_mfma_nt<2>(qkv, mask, dctx, dqkv, B, heads, st);
    case 64: return attn_bwd_mfma_nt<4>(qkv, mask, dctx, dqkv, B, heads, st);
    case 128: {  // the recompute variant at S = 128 too: 19.68 vs 19.74 ms per step (MMSA_ATTN_BWD_RC=0: the image-keeping kernel)
      static const bool rc = !mmsa_disabled("attn_bwd_rc");
      return rc ? attn_bwd_mfma_rc_nt<8>(qkv, mask, dctx, dqkv, B, heads, st)
                : attn_bwd_mfma_nt<8>(qkv, mask, dctx, dqkv, B, heads, st);
    }
    case 192: return attn_bwd_mfma_rc_nt<12>(qkv, mask, dctx, dqkv, B, heads, st);
    case 256: return attn_bwd_mfma_rc_nt<16>(qkv, mask, dctx, dqkv, B, heads, st);  // BERT-large S = 256 (configs[3])
    default: return attn_bwd_simt<bf16>(qkv, mask, dctx, dqkv, ws, B, S, heads, st);
  }
}
