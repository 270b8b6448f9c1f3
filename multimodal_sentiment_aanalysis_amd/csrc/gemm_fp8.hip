// fp8 (OCP e4m3, the CDNA4 format) GEMM for the encoder's forward Linears — BASELINE.json configs[4] "fp8 MFMA encoder GEMMs
// (CDNA4 fp8), bf16 accumulate... " (accumulation here is fp32 in the MFMA, outputs are stored bf16).
//
//   C[M,N] = epilogue( sA * sB * sum_k Aq[m][k] * Bq[n][k] )      Aq, Bq: e4m3 bytes, k-contiguous rows ("NT")
//
// with per-tensor scales sA, sB (device floats) written by the quantizer below from the tensor's own absolute maximum
// ("current scaling": amax pass, then x / (amax / 448) rounded to e4m3). v_mfma_f32_16x16x32_fp8_fp8 has the bf16 MFMA's
// rate on gfx950 (only the MX-scaled K = 128 forms are faster), so what fp8 buys this kernel family is BYTES: a K step of 128
// values is the same 128-byte LDS row as 64 bf16 values — the second-generation bf16 kernel is L2->LDS-rate bound on the
// BERT shapes, fp8 halves that traffic per FLOP.
//
// Structure (first-generation shape, on purpose simple): 128x128 tile per 256-thread workgroup (4 waves, 2x2, 4x4 MFMA tiles
// each), K step = 128 bytes, two LDS stages of 32 KiB filled by global_load_lds (16 B per lane, lane-linear 1-KiB pieces, the
// 16-byte XOR swizzle of gemm_tile.h applied to the per-lane SOURCE address and to the fragment reads), one
// vmcnt(0) + barrier per step. A lane's ds_read_b128 of chunk (g + 4h) feeds TWO MFMAs (bytes 0-7 and 8-15): A and B use the
// same chunk -> k-set assignment, so every k meets its partner exactly once, in a permuted order that the sum does not see.
// Epilogue: the shared float4 epilogue of the other GEMMs (bias, GELU with gelu' side output, residual add, bf16 / fp32 out).
#include <stdlib.h>
#include "gemm.h"
#include "gemm_epilogue.h"
#include "gemm_tile.h"
#include "ops.h"

#define Q_BM 128
#define Q_BN 128
#define Q_BKB 128  // bytes (= fp8 values) per K step
#define Q_TILE 16384
#define Q_STAGE 32768
#define FP8_MAX 448.0f

struct Fp8Params {
  GemmParams g;           // M, N, K (in fp8 values), lda, ldb (bytes per row), C / epilogue fields as in gemm.h
  const float* scale_a;   // device scalar, or one scale per row of A (g.scale_a_rows)
  const float* scale_b;   // device scalar
};

__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(Fp8Params q) {
  const GemmParams& p = q.g;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + Q_BN - 1) / Q_BN, ntm = (p.M + Q_BM - 1) / Q_BM;
  const int nblk = ntm * ntn;
  int bid = blockIdx.x;
  {  // blocks b and b + 8 share an XCD (L2): give each XCD a contiguous run of tiles (bijective for any block count)
    const int qq = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + loc;
  }
  const int tm = bid / ntn, tn = bid - tm * ntn;
  const int m0 = tm * Q_BM, n0 = tn * Q_BN;
  stamp_begin(p.stamp);
  const unsigned char* __restrict__ A = (const unsigned char*)p.A;
  const unsigned char* __restrict__ B = (const unsigned char*)p.B;
  const int nk = p.K / Q_BKB;

  // staging map: wave-instruction i of wave w fills the 1-KiB piece 4 w + i = rows 8 (4w + i) .. + 7; lane -> row + (l >> 3),
  // physical 16-byte chunk l & 7 holds logical chunk (l & 7) ^ (row & 7). Rows past the end are clamped (computed, never stored).
  const unsigned char* srcA[4];
  const unsigned char* srcB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ (row & 7);
    const int ma = min(m0 + row, p.M - 1), nb = min(n0 + row, p.N - 1);
    srcA[i] = A + (long)ma * p.lda + ch * 16;
    srcB[i] = B + (long)nb * p.ldb + ch * 16;
  }
  auto issue = [&](int stage, int kt) __attribute__((always_inline)) {
    const long koff = (long)kt * Q_BKB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned char* da = smem + stage * Q_STAGE + (wave * 4 + i) * 1024;
      unsigned char* db = da + Q_TILE;
      __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + koff), (lptr_t)da, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(srcB[i] + koff), (lptr_t)db, 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int r16 = lane & 15, g = lane >> 4;
  typedef __attribute__((ext_vector_type(2))) long i64x2;

  if (nk > 0) issue(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < nk) issue(st ^ 1, kt + 1);
    const unsigned char* ta = smem + st * Q_STAGE;
    const unsigned char* tb = ta + Q_TILE;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      i64x2 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *(const i64x2*)(ta + kc_off(wm * 64 + i * 16 + r16, g + 4 * h));
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *(const i64x2*)(tb + kc_off(wn * 64 + j * 16 + r16, g + 4 * h));
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fb[j][e], fa[i][e], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }
  // lane holds C[m = .. + r16][n = .. + 4 g + (0..3)] (swapped operands: D rows = n, columns = m)
  const float sb = q.scale_b[0];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + r16;
    const float alpha = (p.scale_a_rows ? (m < p.M ? q.scale_a[m] : 0.f) : q.scale_a[0]) * sb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * g;
      if (m < p.M && n < p.N) gemm_epilogue4<bf16>(p, m, n, acc[i][j] * alpha);
    }
  }
  stamp_end(p.stamp);
}

// ---- quantizer: bf16 -> e4m3 with the tensor's own scale ------------------------------------------------------------------
// Two launches, no atomics: (1) per-workgroup maxima of |x| into part[block] (one same-address atomicMax per wave — 8192 of
// them for a [16384][768] activation — serialised at ~90 per microsecond: the first version spent 90 us per tensor there,
// +6.4 ms on the B = 128 forward); (2) every workgroup of the converter re-reduces the <= FP8_PARTS partial maxima itself.
#define FP8_PARTS 512
__device__ __forceinline__ float block_max256(float m, float* red) {
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(256) void fp8_amax_kernel(const bf16* __restrict__ x, long n8, float* __restrict__ part) {
  __shared__ float red[4];
  float m = 0.f;
  bool bad = false;  // a NaN / Inf element: fmaxf would drop the NaN and the clamp would turn it into a finite value, hiding a
                     // fault from the trainer's NaN rule (Trainer.py:74-76). The maximum becomes +Inf instead: scale = Inf,
                     // every product 0 * Inf or finite * Inf, so the Linear's output is non-finite and the step is skipped.
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < n8; i += 2 * stride) {  // two 16-byte loads in flight per lane
    const bf16x8 v = *(const bf16x8*)(x + i * 8), w = *(const bf16x8*)(x + (i + stride) * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = fabsf((float)v[e]), b = fabsf((float)w[e]);
      bad |= !(a <= 3.0e38f) | !(b <= 3.0e38f);
      m = fmaxf(m, fmaxf(a, b));
    }
  }
  if (i < n8) {
    const bf16x8 v = *(const bf16x8*)(x + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = fabsf((float)v[e]);
      bad |= !(a <= 3.0e38f);
      m = fmaxf(m, a);
    }
  }
  if (bad) m = __builtin_inff();
  m = block_max256(m, red);
  if (threadIdx.x == 0) part[blockIdx.x] = m;
}
__global__ __launch_bounds__(256) void fp8_quant_kernel(const bf16* __restrict__ x, long n8, const float* __restrict__ part, int nparts,
                                                        unsigned char* __restrict__ out, float* __restrict__ scale_out) {
  __shared__ float red[4];
  float am = 0.f;
  for (int k = threadIdx.x; k < nparts; k += 256) am = fmaxf(am, part[k]);
  const float amax = fmaxf(block_max256(am, red), 1e-20f);
  const float scale = amax * (1.0f / FP8_MAX), inv = FP8_MAX / amax;
  if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const bf16x8 v = *(const bf16x8*)(x + i * 8);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf((float)v[e] * inv, -FP8_MAX), FP8_MAX);
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    typedef __attribute__((ext_vector_type(2))) int i32x2_;
    *(i32x2_*)(out + i * 8) = i32x2_{lo, hi};
  }
}

// ---- per-row ("per-token") quantizer: ONE pass -----------------------------------------------------------------------------
// A wave owns a row (K <= 64 * 8 * FP8_ROW_CHUNKS values): its lanes keep the row's 16-byte chunks in registers, take the row
// maximum with DPP / shuffles and convert: 2 B read + 1 B written per element and one launch, against 5 B and two launches of the
// per-tensor form (which has to know the whole tensor's maximum before it can convert anything). scales[m] = amax_m / 448.
#define FP8_ROW_CHUNKS 8
// ROWS rows per wave at a time, all of their loads issued before the first maximum is taken (K = 768 is 1.5 chunks per lane: with
// one row per wave a lane had 1-2 loads in flight and the kernel ran at 1.9 TB/s)
template <int NCH, int ROWS>
__global__ __launch_bounds__(256) void fp8_quant_rows_kernel(const bf16* __restrict__ x, long ldx, int M, int K8,
                                                             unsigned char* __restrict__ out, float* __restrict__ scales) {
  const int lane = threadIdx.x & 63;
  const long m0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
  if (m0 >= M) return;
  bf16x8 v[ROWS][NCH];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // unconditional loads (clamped chunk / row; masked where they are used): a branch around each load makes hipcc wait
      // for every chunk separately — serial memory round trips (30 us instead of ~16 for a [16384][3072] activation)
      const int ch = min(lane + 64 * c, K8 - 1);
      const long row = min(m0 + r, (long)M - 1);
      v[r][c] = *(const bf16x8*)(x + row * ldx + (long)ch * 8);
    }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    if (m0 + r >= M) break;
    float am = 0.f;
    bool bad = false;  // (NaN / Inf: as fp8_amax_kernel, the row's scale becomes Inf and the Linear's output non-finite)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (lane + 64 * c < K8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = fabsf((float)v[r][c][e]);
          bad |= !(a <= 3.0e38f);
          am = fmaxf(am, a);
        }
      }
    }
    if (bad) am = __builtin_inff();
    am = fmaxf(wave_max(am), 1e-20f);
    const float inv = FP8_MAX / am;
    if (lane == 0) scales[m0 + r] = am * (1.0f / FP8_MAX);
    unsigned char* o = out + (m0 + r) * (long)K8 * 8;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < K8) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf((float)v[r][c][e] * inv, -FP8_MAX), FP8_MAX);
        int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        typedef __attribute__((ext_vector_type(2))) int i32x2_;
        *(i32x2_*)(o + (long)ch * 8) = i32x2_{lo, hi};
      }
    }
  }
}
// x[M][K] bf16 (row stride ldx elements; K % 8 == 0, K <= 4096) -> out[M][K] e4m3 bytes (contiguous rows), scales[M]
int fp8_quantize_rows(const void* x, long ldx, int M, int K, void* out, float* scales, hipStream_t st) {
  if (!x || !out || !scales || M <= 0 || K <= 0 || (K % 8) || (ldx % 8) || K > 64 * 8 * FP8_ROW_CHUNKS) return MMSA_ERR_ARG;
  const int K8 = K / 8, nch = cdiv(K8, 64);
  const dim3 block(256);
  const bf16* xp = (const bf16*)x;
  unsigned char* op = (unsigned char*)out;
  if (nch <= 2) hipLaunchKernelGGL((fp8_quant_rows_kernel<2, 4>), dim3(cdiv(M, 16)), block, 0, st, xp, ldx, M, K8, op, scales);
  else if (nch <= 4) hipLaunchKernelGGL((fp8_quant_rows_kernel<4, 2>), dim3(cdiv(M, 8)), block, 0, st, xp, ldx, M, K8, op, scales);
  else if (nch <= 6) hipLaunchKernelGGL((fp8_quant_rows_kernel<6, 2>), dim3(cdiv(M, 8)), block, 0, st, xp, ldx, M, K8, op, scales);
  else hipLaunchKernelGGL((fp8_quant_rows_kernel<8, 1>), dim3(cdiv(M, 4)), block, 0, st, xp, ldx, M, K8, op, scales);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

// ---- per-tensor quantization of MANY tensors in two launches (the weights of every quantized Linear, once per forward) ------
// Tensor t = n8[t] 16-byte chunks at element offset off[t] of one bf16 buffer; its e4m3 bytes go to the same element offset of
// `out`, its scale to scales[t]. Same arithmetic as fp8_amax_kernel / fp8_quant_kernel; blockIdx.y = tensor.
#define FP8_BATCH_PARTS 64
struct Fp8Batch {
  int n;
  long off[FP8_BATCH_MAX], n8[FP8_BATCH_MAX];
};
__global__ __launch_bounds__(256) void fp8_amax_batch_kernel(const bf16* __restrict__ base, Fp8Batch tb, float* __restrict__ part) {
  __shared__ float red[4];
  const int t = blockIdx.y;
  const bf16* x = base + tb.off[t];
  const long n8 = tb.n8[t];
  float m = 0.f;
  bool bad = false;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const bf16x8 v = *(const bf16x8*)(x + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = fabsf((float)v[e]);
      bad |= !(a <= 3.0e38f);
      m = fmaxf(m, a);
    }
  }
  if (bad) m = __builtin_inff();
  m = block_max256(m, red);
  if (threadIdx.x == 0) part[(long)t * FP8_BATCH_PARTS + blockIdx.x] = m;
}
__global__ __launch_bounds__(256) void fp8_quant_batch_kernel(const bf16* __restrict__ base, Fp8Batch tb, const float* __restrict__ part,
                                                              unsigned char* __restrict__ out, float* __restrict__ scales) {
  __shared__ float red[4];
  const int t = blockIdx.y;
  float am = threadIdx.x < FP8_BATCH_PARTS ? part[(long)t * FP8_BATCH_PARTS + threadIdx.x] : 0.f;
  const float amax = fmaxf(block_max256(am, red), 1e-20f);
  const float inv = FP8_MAX / amax;
  if (blockIdx.x == 0 && threadIdx.x == 0) scales[t] = amax * (1.0f / FP8_MAX);
  const bf16* x = base + tb.off[t];
  unsigned char* o = out + tb.off[t];
  const long n8 = tb.n8[t];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const bf16x8 v = *(const bf16x8*)(x + i * 8);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf((float)v[e] * inv, -FP8_MAX), FP8_MAX);
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    typedef __attribute__((ext_vector_type(2))) int i32x2_;
    *(i32x2_*)(o + i * 8) = i32x2_{lo, hi};
  }
}
size_t fp8_quantize_batch_ws_bytes(int n) { return (size_t)(n > 0 ? n : 1) * FP8_BATCH_PARTS * sizeof(float); }
// n <= FP8_BATCH_MAX tensors of `base` (bf16): offsets / sizes in elements (both % 8 == 0); out: e4m3 bytes at the same offsets
int fp8_quantize_batch(const void* base, const long* off, const long* numel, int n, void* out, float* scales, float* ws,
                       hipStream_t st) {
  if (!base || !off || !numel || !out || !scales || !ws || n <= 0 || n > FP8_BATCH_MAX) return MMSA_ERR_ARG;
  Fp8Batch tb;
  tb.n = n;
  for (int t = 0; t < n; ++t) {
    if (numel[t] <= 0 || (numel[t] % 8) || (off[t] % 8)) return MMSA_ERR_ARG;
    tb.off[t] = off[t];
    tb.n8[t] = numel[t] / 8;
  }
  hipLaunchKernelGGL(fp8_amax_batch_kernel, dim3(FP8_BATCH_PARTS, n), dim3(256), 0, st, (const bf16*)base, tb, ws);
  hipLaunchKernelGGL(fp8_quant_batch_kernel, dim3(FP8_BATCH_PARTS, n), dim3(256), 0, st, (const bf16*)base, tb, (const float*)ws,
                     (unsigned char*)out, scales);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

size_t fp8_quantize_ws_bytes() { return FP8_PARTS * sizeof(float); }

// x[n] bf16 (n % 8 == 0) -> out[n] e4m3 bytes, *scale = amax / 448; `amax_ws`: fp8_quantize_ws_bytes() of device scratch
int fp8_quantize(const void* x, long n, void* out, float* scale, unsigned* amax_ws, hipStream_t st) {
  if (!x || !out || !scale || !amax_ws || n <= 0 || (n % 8)) return MMSA_ERR_ARG;
  const long n8 = n / 8;
  const int parts = (int)min((n8 + 511) / 512, (long)FP8_PARTS);
  const int blocks = (int)min((n8 + 255) / 256, 4096L);
  hipLaunchKernelGGL(fp8_amax_kernel, dim3(parts), dim3(256), 0, st, (const bf16*)x, n8, (float*)amax_ws);
  hipLaunchKernelGGL(fp8_quant_kernel, dim3(blocks), dim3(256), 0, st, (const bf16*)x, n8, (const float*)amax_ws, parts,
                     (unsigned char*)out, scale);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

bool gemm_fp8_eligible(const GemmParams& p) {
  return p.gather == 0 && !p.a_kmajor && !p.b_kmajor && p.split_k <= 1 && !(p.K % Q_BKB) && !(p.N % 4) && !(p.lda % 16) &&
         !(p.ldb % 16) && p.M > 0 && p.c_gw == 0 && !(p.out_f32 && p.accumulate);
}

// p.A / p.B: e4m3 bytes (k-contiguous rows, lda / ldb in BYTES = fp8 elements); everything else as a bf16 NT GEMM of gemm.h
// pin.scale_a_rows: scale_a holds one scale per row of A (fp8_quantize_rows) instead of one for the tensor
int gemm_fp8_launch(const GemmParams& pin, const float* scale_a, const float* scale_b, hipStream_t st) {
  if (!gemm_fp8_eligible(pin) || !scale_a || !scale_b) return MMSA_ERR_UNSUPPORTED;
  Fp8Params q;
  q.g = pin;
  q.scale_a = scale_a;
  q.scale_b = scale_b;
  const int slot = gemm_prof_open(q.g, st);
  {  // the persistent kernel's fp8 instantiation (gemm_mfma2.hip) when the shape suits it: K / lda / ldb in 2-byte units
    static const bool simple = [] { const char* v = MMSA_EXP_ENV("MMSA_FP8_SIMPLE"); return v && atoi(v) != 0; }();  // A/B hook
    GemmParams p2 = q.g;
    p2.K = pin.K / 2; p2.lda = pin.lda / 2; p2.ldb = pin.ldb / 2;
    p2.scale_a = scale_a; p2.scale_b = scale_b;
    p2.split_k = 1; p2.ws = nullptr; p2.ws_bytes = 0;
    if (!simple && pin.M >= 256 && gemm2_eligible(p2)) {
      const int rc = gemm2_launch(p2, 0, st);
      gemm_prof_close(slot, st);
      return rc;
    }
  }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_fp8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Q_STAGE);
    attr = true;
  }
  const int blocks = cdiv(pin.M, Q_BM) * cdiv(pin.N, Q_BN);
  hipLaunchKernelGGL(gemm_fp8_kernel, dim3(blocks), dim3(256), 2 * Q_STAGE, st, q);
  gemm_prof_close(slot, st);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
