// SIMT (VALU fma) GEMM with the same descriptor as the MFMA kernel, templated on the storage type.
//
// Two uses: (1) storage = fp32: the exact-fp32 execution mode of the whole path (parity against the fp32 CPU
// oracle at 1e-3 / 1e-4) and the small fp32 Linears of the fusion head; (2) storage = bf16: an independent
// on-device cross-check of the MFMA kernel (same inputs, same fp32 accumulation, different code).
// 64x64 tile, 256 threads, 4x4 outputs per thread, K step 16. Addressing is per element and fully general
// (every gather mode of gemm.h), which is what makes it a useful checker; it is not a performance kernel.
#include "gemm.h"
#include "gemm_epilogue.h"
#include "gemm_generic.h"
#include <type_traits>

#define SBM 64
#define SBN 64
#define SBK 32

template <typename T>
__global__ __launch_bounds__(256) void gemm_simt_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float As[SBK][SBM + 4];
  __shared__ __attribute__((aligned(16))) float Bs[SBK][SBN + 4];
  const int tid = threadIdx.x;
  const int ntn = (p.N + SBN - 1) / SBN;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
  const int m0 = tm * SBM, n0 = tn * SBN;
  int kbeg = 0, kend = p.K;
  if (p.split_k > 1) {
    const int steps = (p.K + SBK - 1) / SBK;
    const int per = (steps + p.split_k - 1) / p.split_k;
    kbeg = blockIdx.y * per * SBK;
    kend = min(p.K, kbeg + per * SBK);
  }
  const int ty = tid >> 4, tx = tid & 15;  // thread computes rows ty*4.., cols tx*4..
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  // fp32 plain (non-gather) operands with 16-byte aligned rows take 4 elements per load along their contiguous
  // dimension (the fusion head's Linears: 57 launches per step, each a short serial K loop)
  const bool vec = std::is_same<T, float>::value && p.gather == 0 && !(p.lda & 3) && !(p.ldb & 3) && !(p.K & 3) &&
                   !(p.M & 3) && !(p.N & 3) && !(((size_t)p.A | (size_t)p.B) & 15);
  // fp32 fast path: register double buffering — the global loads of step k+1 are in flight while step k is computed
  // (the head's split launches walk 2 steps per block: their two HBM round trips used to run back to back)
  f32x4 va[2], vb[2];
  auto vec_load = [&](int k0) __attribute__((always_inline)) {
    const float* A = (const float*)p.A;
    const float* B = (const float*)p.B;
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // 64 x 32 elements = 512 float4 per operand, 2 per thread
      const int e = tid + 256 * i;
      va[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      vb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.a_kmajor) {  // [K][M]: 16 float4 per k-row
        const int kk = e >> 4, mm = (e & 15) * 4;
        if (k0 + kk < kend && m0 + mm < p.M) va[i] = *(const f32x4*)(A + (long)(k0 + kk) * p.lda + m0 + mm);
      } else {           // [M][K]: 8 float4 per row
        const int mm = e >> 3, kk = (e & 7) * 4;
        if (k0 + kk < kend && m0 + mm < p.M) va[i] = *(const f32x4*)(A + (long)(m0 + mm) * p.lda + k0 + kk);
      }
      if (p.b_kmajor) {
        const int kk = e >> 4, nn = (e & 15) * 4;
        if (k0 + kk < kend && n0 + nn < p.N) vb[i] = *(const f32x4*)(B + (long)(k0 + kk) * p.ldb + n0 + nn);
      } else {
        const int nn = e >> 3, kk = (e & 7) * 4;
        if (k0 + kk < kend && n0 + nn < p.N) vb[i] = *(const f32x4*)(B + (long)(n0 + nn) * p.ldb + k0 + kk);
      }
    }
  };
  if (vec && kbeg < kend) vec_load(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += SBK) {
    if (vec) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;
        if (p.a_kmajor) {
          const int kk = e >> 4, mm = (e & 15) * 4;
          *(f32x4*)&As[kk][mm] = va[i];
        } else {
          const int mm = e >> 3, kk = (e & 7) * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) As[kk + r][mm] = va[i][r];
        }
        if (p.b_kmajor) {
          const int kk = e >> 4, nn = (e & 15) * 4;
          *(f32x4*)&Bs[kk][nn] = vb[i];
        } else {
          const int nn = e >> 3, kk = (e & 7) * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) Bs[kk + r][nn] = vb[i][r];
        }
      }
      if (k0 + SBK < kend) vec_load(k0 + SBK);  // prefetch: consumed after this step's compute
    } else {

    // 64x16 elements per operand, 4 per thread. Pick the thread->element map so global reads are coalesced
    // along the contiguous dimension of each storage form.
    float ta[SBK / 4], tb[SBK / 4];
#pragma unroll
    for (int i = 0; i < SBK / 4; ++i) {  // issue all global loads of the step before any LDS write
      const int e = tid + 256 * i;
      int mm, kk;
      if (p.a_kmajor) { mm = e & 63; kk = e >> 6; } else { kk = e & (SBK - 1); mm = e / SBK; }
      ta[i] = simt_load_a<T>(p, m0 + mm, k0 + kk, kend);
      int nn, kb;
      if (p.b_kmajor) { nn = e & 63; kb = e >> 6; } else { kb = e & (SBK - 1); nn = e / SBK; }
      tb[i] = simt_load_b<T>(p, k0 + kb, n0 + nn, kend);
    }
#pragma unroll
    for (int i = 0; i < SBK / 4; ++i) {
      const int e = tid + 256 * i;
      int mm, kk;
      if (p.a_kmajor) { mm = e & 63; kk = e >> 6; } else { kk = e & (SBK - 1); mm = e / SBK; }
      As[kk][mm] = ta[i];
      int nn, kb;
      if (p.b_kmajor) { nn = e & 63; kb = e >> 6; } else { kb = e & (SBK - 1); nn = e / SBK; }
      Bs[kb][nn] = tb[i];
    }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < SBK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  const bool vec_ok = !(p.N & 3) && !(p.ldc & 3) && !(p.ldc2 & 3) && !(p.ldmul & 3) && !(p.ldadd & 3);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    const int n = n0 + tx * 4;
    if (m >= p.M || n >= p.N) continue;
    if (vec_ok) {
      f32x4 v = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
      if (p.split_k > 1) *(f32x4*)(p.ws + ((long)blockIdx.y * p.M + m) * p.N + n) = v;
      else gemm_epilogue4<T>(p, m, n, v);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n + j < p.N) gemm_epilogue1<T>(p, m, n + j, acc[i][j]);
    }
  }
}

template <typename T>
static int simt_launch(const GemmParams& pin, hipStream_t st) {
  GemmParams p = pin;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return MMSA_ERR_ARG;
  if (p.c_gw > 0) return MMSA_ERR_UNSUPPORTED;
  if (p.split_k < 1) p.split_k = 1;
  if (p.split_k > 1 && (!p.ws || (p.N % 4))) return MMSA_ERR_ARG;
  dim3 grid(cdiv(p.M, SBM) * cdiv(p.N, SBN), p.split_k, 1);
  hipLaunchKernelGGL(gemm_simt_kernel<T>, grid, dim3(256), 0, st, p);
  MMSA_CHECK_LAUNCH();
  if (p.split_k > 1) {
    launch_splitk_reduce<T>(p, st);
    MMSA_CHECK_LAUNCH();
  }
  return MMSA_OK;
}

int gemm_f32_valu_launch(const GemmParams& p, hipStream_t st) { return simt_launch<float>(p, st); }
int gemm_f32_launch(const GemmParams& pin, hipStream_t st) {
  if (gemm_f32_tiny_eligible(pin)) return gemm_f32_tiny_launch(pin, st);
  if (!gemm_f32_mfma_eligible(pin)) return simt_launch<float>(pin, st);
  GemmParams p = pin;
  const int slot = gemm_prof_open(p, st);
  const int rc = gemm_f32_mfma_launch(p, st);
  gemm_prof_close(slot, st);
  return rc;
}
int gemm_bf16_simt_launch(const GemmParams& p, hipStream_t st) { return simt_launch<bf16>(p, st); }
