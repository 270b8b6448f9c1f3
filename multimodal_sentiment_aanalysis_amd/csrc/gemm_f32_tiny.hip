// Latency-oriented exact-fp32 GEMM for the fusion head's batch-row problems (M = 64-128 rows, N, K <= 768; A1-A5 of
// MultimodalModel.py:108-149, 171-225, 388-451).
//
// These launches are not throughput problems: 64 x 256 x 768 is 25 MFLOP.  The VALU kernel (gemm_simt.hip) walks K in
// 16-wide LDS-staged steps, ~0.15 us of latency each, so it needs a K split over workgroups plus a reducer launch to get
// under 10 us.  Here one WAVE owns a 16 x 16 output tile on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate),
// reads its operands straight from global memory 64 k-values at a time, up to three such chunks in flight at once, and
// the 4 waves of a workgroup split K between them; their partial tiles are added in wave order through LDS (fixed
// summation order: bitwise reproducible) and the usual fused epilogue is applied.  One launch, no workspace.
//
// Operand fragment of the 16x16x4 MFMA: lane (r = lane % 16, g = lane / 16) supplies A[r][k_g] and B[k_g][r].  A dot
// product does not care which real k sits in which (step, g) slot as long as A and B agree, so lane (r, g) takes the 16
// CONTIGUOUS k-values g*16 .. g*16+15 of its chunk (four 16-byte loads when k is the contiguous dimension) and step s uses
// element s.
#include "gemm_epilogue.h"

#define TY_CHUNK 64

template <bool KM, typename T>
__device__ __forceinline__ void tiny_load(const T* __restrict__ X, long ld, int x, int X_n, int kb, int K, float* v, bool vec) {
  // X_n: extent of the non-contracted dimension; x: this lane's row / column in it; kb: first of the lane's 16 k-values
  // vec (workgroup-uniform): the k-contiguous rows can be read 16 bytes at a time (K, ld multiples of the vector, base aligned)
  if (!KM && !vec) {  // e.g. the data gradient of a 3-class head: K = 3
    const T* row = X + (long)x * ld + kb;
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = (x < X_n && kb + s < K) ? to_f32<T>(row[s]) : 0.f;
    return;
  }
  if constexpr (!KM && sizeof(T) == 4) {
    const float* row = (const float*)X + (long)x * ld + kb;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (x < X_n && kb + 4 * q < K) t = *(const f32x4*)(row + 4 * q);  // K % 4 == 0: a float4 never straddles the end
      v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
    }
  } else if constexpr (!KM) {  // bf16 operands (the encoders' batch-row Linears): widened, same exact-product MFMA
    const bf16* row = (const bf16*)X + (long)x * ld + kb;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      bf16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
      if (x < X_n && kb + 8 * q < K) t = *(const bf16x8*)(row + 8 * q);  // K % 8 == 0
#pragma unroll
      for (int e = 0; e < 8; ++e) v[8 * q + e] = (float)t[e];
    }
  } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = (x < X_n && kb + s < K) ? to_f32<T>(X[(long)(kb + s) * ld + x]) : 0.f;
  }
}

// One workgroup's share of one problem: `blk` = its block index inside the problem's tile range
template <typename T, bool A_KM, bool B_KM>
__device__ __forceinline__ void tiny_body(const GemmParams& p, int KS, int tiles_n, int ntiles, int blk, float (*img)[256]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int tpb = 4 / KS;
  const int tib = wave / KS, ks = wave - tib * KS;
  const int tile = blk * tpb + tib;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (tile < ntiles) {
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m = tm * 16 + r, n = tn * 16 + r;
    const T* A = (const T*)p.A;
    const T* B = (const T*)p.B;
    const int nch = (p.K + TY_CHUNK - 1) / TY_CHUNK;
    constexpr int V = 16 / sizeof(T);
    const bool vec_a = !(p.K % V) && !(p.lda % V) && !((uintptr_t)p.A & 15);
    const bool vec_b = !(p.K % V) && !(p.ldb % V) && !((uintptr_t)p.B & 15);
    // up to three chunks per wave in flight at once (K <= 768 with the 4-way split: ONE memory round trip per launch)
    float a[3][16], b[3][16];
    for (int c = ks; c < nch; c += 3 * KS) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int cu = c + u * KS;
        if (cu < nch) {
          const int kb = cu * TY_CHUNK + g * 16;
          tiny_load<A_KM>(A, p.lda, m, p.M, kb, p.K, a[u], vec_a);
          if (B_KM && p.b_ones) {  // the ones column (bias gradient): nothing to read
#pragma unroll
            for (int s = 0; s < 16; ++s) b[u][s] = (n == 0 && kb + s < p.K) ? 1.f : 0.f;
          } else {
            tiny_load<B_KM>(B, p.ldb, n, p.N, kb, p.K, b[u], vec_b);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (c + u * KS < nch) {
#pragma unroll
          for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s], b[u][s], acc, 0, 0, 0);
        }
      }
    }
  }
  // C fragment: acc[j] = C[4g + j][r]
#pragma unroll
  for (int j = 0; j < 4; ++j) img[wave][(4 * g + j) * 16 + r] = acc[j];
  __syncthreads();
  const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
  for (int t = 0; t < tpb; ++t) {
    const int tl = blk * tpb + t;
    if (tl >= ntiles) break;
    const int tm = tl / tiles_n, tn = tl - tm * tiles_n;
    const int m = tm * 16 + row, n = tn * 16 + col;
    float v = img[t * KS][threadIdx.x];
    for (int k = 1; k < KS; ++k) v += img[t * KS + k][threadIdx.x];
    if (m < p.M && n < p.N) gemm_epilogue1<T>(p, m, n, v);
  }
}

template <typename T, bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void gemm_tiny_kernel(GemmParams p, int KS, int tiles_n, int ntiles) {
  __shared__ float img[4][256];
  tiny_body<T, A_KM, B_KM>(p, KS, tiles_n, ntiles, blockIdx.x, img);
}

// Up to 3 problems of any operand layout in one launch (the backward of one Linear: dW = dy^T x, db = dy^T 1, dx = dy W):
// a workgroup belongs to exactly one problem (block ranges), so the layout dispatch is workgroup-uniform.
#define TINY_MAX_PROBS 3
struct TinyBatch {
  GemmParams p[TINY_MAX_PROBS];
  int blk_begin[TINY_MAX_PROBS + 1];
  int KS[TINY_MAX_PROBS], tiles_n[TINY_MAX_PROBS], ntiles[TINY_MAX_PROBS];
  int n;
};
__global__ __launch_bounds__(256) void gemm_tiny_multi_kernel(TinyBatch tb) {
  __shared__ float img[4][256];
  int q = 0;
#pragma unroll
  for (int i = 1; i < TINY_MAX_PROBS; ++i)
    if (i < tb.n && (int)blockIdx.x >= tb.blk_begin[i]) q = i;
  const GemmParams& p = tb.p[q];
  const int blk = blockIdx.x - tb.blk_begin[q];
  if (p.a_kmajor) {
    if (p.b_kmajor) tiny_body<float, true, true>(p, tb.KS[q], tb.tiles_n[q], tb.ntiles[q], blk, img);
    else tiny_body<float, true, false>(p, tb.KS[q], tb.tiles_n[q], tb.ntiles[q], blk, img);
  } else {
    if (p.b_kmajor) tiny_body<float, false, true>(p, tb.KS[q], tb.tiles_n[q], tb.ntiles[q], blk, img);
    else tiny_body<float, false, false>(p, tb.KS[q], tb.tiles_n[q], tb.ntiles[q], blk, img);
  }
}

template <typename T>
static bool tiny_eligible(const GemmParams& p) {
  static const bool off = mmsa_disabled("f32_tiny");
  if (off || p.gather != 0 || p.c_gw > 0 || p.M <= 0 || p.N <= 0 || p.K <= 0 || p.scale_a) return false;
  if (p.b_ones && (!p.b_kmajor || p.N != 1 || sizeof(T) != 4)) return false;
  // beyond: operand re-reads (no LDS sharing between the tiles of a workgroup) start to cost.  bf16: one notch higher so
  // that the BERT pooler (64 x 768 x 768) is in
  if ((double)p.M * p.N * p.K > (double)(1 << (sizeof(T) == 4 ? 25 : 26))) return false;
  return true;  // any stride / alignment: rows that cannot be read 16 bytes at a time take the scalar loads
}
bool gemm_f32_tiny_eligible(const GemmParams& p) { return tiny_eligible<float>(p); }
bool gemm_bf16_tiny_eligible(const GemmParams& p) {
  static const bool off = mmsa_disabled("bf16_tiny");
  return !off && p.M <= 128 && tiny_eligible<bf16>(p);
}

template <typename T>
static int tiny_launch(const GemmParams& pin, hipStream_t st) {
  GemmParams p = pin;
  p.split_k = 1;  // the waves of a workgroup split K; no slabs, no reducer
  p.ws = nullptr;
  const int tiles_m = cdiv(p.M, 16), tiles_n = cdiv(p.N, 16), ntiles = tiles_m * tiles_n;
  const int KS = p.K >= 256 ? 4 : (p.K >= 128 ? 2 : 1);
  const dim3 grid(cdiv(ntiles, 4 / KS));
#define TINY_CASE(AK, BK) \
  hipLaunchKernelGGL((gemm_tiny_kernel<T, AK, BK>), grid, dim3(256), 0, st, p, KS, tiles_n, ntiles)
  if (p.a_kmajor) { if (p.b_kmajor) TINY_CASE(true, true); else TINY_CASE(true, false); }
  else            { if (p.b_kmajor) TINY_CASE(false, true); else TINY_CASE(false, false); }
#undef TINY_CASE
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
int gemm_f32_tiny_launch(const GemmParams& p, hipStream_t st) { return tiny_launch<float>(p, st); }
int gemm_bf16_tiny_launch(const GemmParams& p, hipStream_t st) { return tiny_launch<bf16>(p, st); }

int gemm_f32_tiny_launch_multi(const GemmParams* ps, int n, hipStream_t st) {
  if (n < 1 || n > TINY_MAX_PROBS) return MMSA_ERR_UNSUPPORTED;
  TinyBatch tb;
  tb.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (!tiny_eligible<float>(ps[i])) return MMSA_ERR_UNSUPPORTED;
    GemmParams& p = tb.p[i];
    p = ps[i];
    p.split_k = 1;
    p.ws = nullptr;
    const int tiles_m = cdiv(p.M, 16);
    tb.tiles_n[i] = cdiv(p.N, 16);
    tb.ntiles[i] = tiles_m * tb.tiles_n[i];
    tb.KS[i] = p.K >= 256 ? 4 : (p.K >= 128 ? 2 : 1);
    tb.blk_begin[i] = blocks;
    blocks += cdiv(tb.ntiles[i], 4 / tb.KS[i]);
  }
  for (int i = n; i <= TINY_MAX_PROBS; ++i) tb.blk_begin[i] = blocks;
  for (int i = n; i < TINY_MAX_PROBS; ++i) { tb.p[i] = tb.p[0]; tb.KS[i] = 1; tb.tiles_n[i] = 1; tb.ntiles[i] = 0; }
  hipLaunchKernelGGL(gemm_tiny_multi_kernel, dim3(blocks), dim3(256), 0, st, tb);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
