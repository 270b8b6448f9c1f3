// GEMM descriptor shared by the bf16 MFMA kernel (gemm_mfma.hip) and the fp32 SIMT kernel (gemm_f32.hip).
//
//   C[M,N] = epilogue( sum_k opA[m][k] * opB[k][n] )
//
// Operand storage:
//   a_kmajor = 0 : A stored [M][K], k contiguous, row stride lda        (activations in a forward Linear)
//   a_kmajor = 1 : A stored [K][M], m contiguous, k-row stride lda      (dY^T in a weight-gradient GEMM)
//   b_kmajor = 0 : B stored [N][K], k contiguous, row stride ldb        (nn.Linear weight [out,in])
//   b_kmajor = 1 : B stored [K][N], n contiguous, k-row stride ldb      (weight in a data-gradient GEMM, X in wgrad)
//
// Implicit-GEMM convolution (NHWC activations, weights [Cout][KH][KW][Cin]) is expressed as a row gather:
//   gather = 1 : A (k-contiguous) rows are pixels of a GH x GW grid; k = tap*cper + c; source pixel of (row, tap) is
//                sy = y*mul + ky*kmul + off (same in x); when div > 1 the tap is valid only if sy % div == 0 and
//                sy /= div. Invalid taps (padding, stride holes, rows >= M) read zeros.
//                forward conv: mul = stride, kmul = +1, off = -pad, div = 1  (source = input activation)
//                data grad   : mul = 1, kmul = -1, off = +pad, div = stride  (source = dY)
//                With gather = 1 and b_kmajor = 1 the B row of k is (k % cper) * ldb + (k / cper) * b_tap_stride
//                (weight [Cout][taps][Cin] read as [k = (tap, cout)][n = cin]).
//                parity class of a stride-2 data grad: see conv_dgrad (resnet_engine.hip): a stride-1 gather over the
//                class's own taps (off / offx = the class parities, tap strides 2x the whole filter's).
//   gather = 2 : B (k-major) rows (k) are output pixels, columns n = tap*cper + c; source as forward conv above
//                (weight gradient: dW[cout][tap][cin] = sum_pixels dY[pixel][cout] * X[shifted pixel][cin]).
#pragma once
#include "common.h"

struct ConvGeom {
  int SH, SW;           // source spatial size
  int GH, GW;           // row-space grid (rows -> (img, y, x))
  int KH, KW;
  int mul, kmul, off, div;
  int offx;             // the x offset (off is the y offset; equal for whole convolutions, different for parity classes)
  int cper;             // channels per tap along K (gather=1) or along N (gather=2)
  long src_pix_stride;  // elements between consecutive source pixels
  FastDiv fd_gw, fd_ghw, fd_kw, fd_cper;
};

struct GemmParams {
  const void* A;
  const void* B;
  void* C;
  int M, N, K;
  long lda, ldb, ldc;
  int a_kmajor, b_kmajor;
  int gather;
  long b_tap_stride;
  long b_tap_stride_y;    // element stride of a tap row (0: KW * b_tap_stride); tap (ky, kx) sits at ky * this + kx * b_tap_stride
  ConvGeom g;
  // epilogue: v = acc (+ bias[n]); C2 = v (optional, storage type); v = act(v); v *= gelu'(mul[m][n]) (optional);
  //           v += add[m][n] (optional); (v = act(v) here instead when act_after_add;) C = v (storage type, or fp32 when out_f32;
  //           += when accumulate)
  const float* bias;
  void* C2;
  long ldc2;
  int act;
  const void* mul;
  long ldmul;
  const void* add;
  long ldadd;
  // act_after_add: the activation is applied AFTER `add` (a bottleneck's output in inference mode, with its BatchNorm folded into
  // the convolution's weights and bias, is relu(conv3'(x) + b' + identity)) instead of before it.
  int act_after_add;
  // engine-internal variants of the two GELU operands (0 = the documented forms): c2_gelu_grad: C2 receives gelu'(v)
  // instead of v (act must be GELU); mul_is_factor: `mul` already holds that factor (v *= mul, no gelu' evaluation).
  // The forward has exp(-v^2/2) and erf in registers anyway, so the backward GEMM's epilogue loses ~25 VALU per element.
  int c2_gelu_grad, mul_is_factor;
  int out_f32;
  int accumulate;
  // split-K: when split_k > 1 raw fp32 partials go to ws[split][M][N] and a second kernel reduces them into C
  int split_k;
  float* ws;
  long ws_bytes;          // capacity of ws (0: exactly split_k slabs); lets the persistent kernel pick its own K split
  const void* zero_page;  // >= 64 bytes of zeros in device memory (source for padded taps)
  // filled by the MFMA launcher: byte extents of the A / B views for the buffer-descriptor staging path
  unsigned a_bytes, b_bytes;
  int use_srd;
  // fp8 (e4m3) operands (gemm_fp8.hip; the persistent kernel's NT instantiation): A / B hold e4m3 BYTES, K / lda / ldb are given
  // in 2-byte units (K = values / 2), the result is scaled by scale_a[0] * scale_b[0] (device scalars). null: bf16 operands.
  const float* scale_a;
  const float* scale_b;
  int scale_a_rows;  // scale_a holds one scale per ROW of A (per-token quantization, fp8_quantize_rows) instead of scale_a[0]
  // batch-row kernel only (gemm_f32_tiny.hip): B is a [K][1] column of ones that is never read (N must be 1): the product
  // is the column sum of the k-major A, i.e. a bias gradient as one more problem of a grouped launch
  int b_ones;
  // optional per-column statistics of the STORED output (the BatchNorm batch statistics of a convolution's output, taken in the
  // conv GEMM's own epilogue instead of a separate pass over z: resnet_engine.hip): colstat[row][2][N] receives, for every 64-row
  // slice `row` of the output, sum_m q(C[m][n]) and sum_m q(C[m][n])^2 (q = the rounding to the storage type). Only the persistent
  // kernel's plain bf16 store epilogue without a K split produces them; *colstat_rows (host) is set to the number of slices
  // written (cdiv(M, 64)), or 0 when this launch did not (the caller then runs its own statistics pass).
  float* colstat;
  long colstat_cap;       // capacity of colstat in floats
  int* colstat_rows;
  // optional in-kernel timing record {min start, max end} in s_memrealtime ticks (100 MHz), filled by the MFMA kernels
  // and the split-K reducer when non-null (bench.py roofline: HIP event pairs add ~12 us of queue drain per launch)
  unsigned long long* stamp;
  // optional output row map (bf16 plain-store epilogue of the persistent kernel only): row m = (img, y, x) of a
  // c_gh x c_gw grid is stored at element offset img * c_imgpitch + y * c_rowpitch + x * c_colpitch (+ n) instead of
  // m * ldc. Used by the data gradient of a strided 1x1 convolution: a dense GEMM over the (4x fewer) output-gradient
  // rows scattered to the even pixels of a zero-filled input gradient. c_gw == 0: no map.
  int c_gw, c_gh;
  long c_imgpitch, c_rowpitch, c_colpitch;
  FastDiv fd_c_ghw, fd_c_gw;
  // c_rmw (with a row map; NN problems): the mapped rows are ADDED to what is stored there (fp32 add of the accumulator and the
  // stored bf16 value, rounded once) instead of overwriting a zero-filled buffer: the strided 1x1 data gradient lands on top of the
  // block's main-path data gradient, which was written first — no zero fill, and the main path's GEMM reads no side operand
  // that is 3/4 zeros.
  int c_rmw;
};

// element offset of filter tap (ky, kx) in the k-major weight view of a gather = 1 data gradient
__host__ __device__ __forceinline__ long b_tap_offset(const GemmParams& p, int ky, int kx) {
  return (long)ky * (p.b_tap_stride_y ? p.b_tap_stride_y : (long)p.g.KW * p.b_tap_stride) + (long)kx * p.b_tap_stride;
}

int gemm_bf16_launch(const GemmParams& p, hipStream_t st);
bool gemm2_eligible(const GemmParams& p);
int gemm2_launch(const GemmParams& p, size_t ws_bytes_avail, hipStream_t st);
extern thread_local int g2_last_plan[3];  // (thread_local: the two encoders may be enqueued from two host threads)
#define GEMM_MAX_GROUPS 12  // problems of one grouped weight-gradient launch
int gemm2_launch_group(const GemmParams* probs, float* const* colsum, int n, hipStream_t st);
int gemm_bf16_launch_group(const GemmParams* probs, float* const* colsum, int n, hipStream_t st);  // + roofline record
// streaming kernel for the HBM-bound 1x1 convolutions (gemm_stream.hip); gemm_bf16_launch routes eligible problems to it
bool gemm_stream_eligible(const GemmParams& p);
int gemm_stream_launch(const GemmParams& p, hipStream_t st);
int gemm_f32_launch(const GemmParams& p, hipStream_t st);
// exact-fp32 GEMM on the matrix cores (gemm_f32_mfma.hip); gemm_f32_launch routes eligible problems to it
bool gemm_f32_mfma_eligible(const GemmParams& p);
int gemm_f32_valu_launch(const GemmParams& p, hipStream_t st);  // the VALU-fma kernel, unconditionally
// one-launch fp32-MFMA kernel for batch-row problems (gemm_f32_tiny.hip); gemm_f32_launch routes eligible problems to it
bool gemm_f32_tiny_eligible(const GemmParams& p);
int gemm_f32_tiny_launch(const GemmParams& p, hipStream_t st);
// up to 3 independent fp32 batch-row problems (weight gradient + bias gradient + data gradient of one Linear) as ONE launch;
// MMSA_ERR_UNSUPPORTED when one of them is not eligible (nothing launched)
int gemm_f32_tiny_launch_multi(const GemmParams* ps, int n, hipStream_t st);
// the same kernel on bf16 operands (<= 128 rows: BERT pooler, the two projections into the fusion width)
bool gemm_bf16_tiny_eligible(const GemmParams& p);
int gemm_bf16_tiny_launch(const GemmParams& p, hipStream_t st);
int gemm_f32_mfma_launch(const GemmParams& p, hipStream_t st);
// roofline record of one matrix-core launch made outside gemm_mfma.hip: open returns a slot (or -1: not recording) and sets
// p.stamp / records the start event; close records the end event
int gemm_prof_open(GemmParams& p, hipStream_t st);
void gemm_prof_close(int slot, hipStream_t st);
size_t gemm_splitk_ws_bytes(int M, int N, int split_k);
const void* mmsa_zero_page();
