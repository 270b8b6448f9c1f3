// Exact-fp32 GEMM on the matrix cores: v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate; MI355X: 157 TFLOP/s peak,
// = the fp32 vector peak, but one VGPR per operand and no VALU work in the inner loop). This is the accuracy mode of
// the whole path (`precision="fp32"`: fp32 storage, the north-star's 1e-3 logits / 1e-4 loss tolerances against the fp32
// CPU oracle) — round 1 ran that mode on the SIMT kernel (gemm_simt.hip), which stays as the checker and as the
// small-problem / odd-shape fallback.
//
// Numerics: the hardware's result is bit for bit a k-ordered fmaf chain (one rounding per product), and this kernel walks
// k in the same order as the SIMT kernel: for the SAME K split the two agree BITWISE (tests/test_kernels_gpu.py::
// test_gemm_f32_mfma_matches_simt_bitwise passes the split explicitly). Inside the engines the launcher below plans its OWN
// split when it is lent a slab workspace (more slices than the SIMT kernel would take): the results then differ from the SIMT
// path by fp32 summation order only (the exact-mode tests bound that at 1e-5 relative).
//
// Structure: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA blocks of 32x32),
// K step 16. Operands are staged through registers (fp32 views come with every addressing mode of gemm.h, so there is
// no lane-linear LDS-DMA image to exploit): the global loads of step k+1 are issued before the MFMAs of step k and
// written to LDS after them. LDS tiles are k-major ([16][130] floats: lanes of an MFMA operand read 32 consecutive
// floats of one k-row, conflict-free; the 130 stride makes the transposing writes of k-contiguous operands
// conflict-free as well). Two workgroups per CU hide each other's barriers. Fast loaders exist for the layouts the
// engines use (k-contiguous rows, m/n-contiguous k-rows, both convolution gathers); anything else — tile edges, odd
// strides, K tails — goes through the generic per-element loaders (gemm_generic.h).
#include <type_traits>
#include "gemm.h"
#include "gemm_epilogue.h"
#include "gemm_generic.h"
#include "gemm_tile.h"
#include <stdlib.h>

#define FBM 128
#define FBN 128
#define FBK 16
#define FLD 130
#define F_LDS_FLOATS (4 * 32 * 36 > 2 * FBK * FLD ? 4 * 32 * 36 : 2 * FBK * FLD)
#ifndef F32_WPS
#define F32_WPS 4  // waves per SIMD the register budget is set for (4 = 4 workgroups per CU; measured against 2 and 3)
#endif
#define FSPLIT_GRAN 32  // split-K partition granularity = the SIMT kernel's K step (identical slabs -> identical sums)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum { F_GENERIC = 0, F_KCONTIG = 1, F_XCONTIG = 2, F_GATHER1 = 3, F_GATHER1_B = 4, F_GATHER2 = 5 };

struct F32Plan {
  int amode, bmode;
};

// Which fast loader fits operand A / B of this problem (host side; the kernel re-checks nothing).
static F32Plan f32_plan(const GemmParams& p) {
  F32Plan pl{F_GENERIC, F_GENERIC};
  const bool k16 = !(p.K % FBK);
  const bool taps_ok = p.gather == 0 || (!(p.g.cper % FBK) && !(p.g.src_pix_stride % 4));
  if (k16 && taps_ok && !((size_t)p.A & 15)) {
    if (p.gather == 1 && !p.a_kmajor) pl.amode = F_GATHER1;
    else if (p.gather != 1 && !p.a_kmajor && !(p.lda % 4)) pl.amode = F_KCONTIG;
    else if (p.a_kmajor && !(p.lda % 4) && !(p.M % 4)) pl.amode = F_XCONTIG;
  }
  if (k16 && taps_ok && !((size_t)p.B & 15)) {
    if (p.gather == 2) { if (!(p.g.cper % 4) && !(p.N % 4)) pl.bmode = F_GATHER2; }
    else if (!p.b_kmajor && !(p.ldb % 4)) pl.bmode = F_KCONTIG;
    else if (p.b_kmajor && p.gather == 1) { if (!(p.ldb % 4) && !(p.N % 4) && !(p.b_tap_stride % 4) && !(p.b_tap_stride_y % 4)) pl.bmode = F_GATHER1_B; }
    else if (p.b_kmajor && !(p.ldb % 4) && !(p.N % 4)) pl.bmode = F_XCONTIG;
  }
  return pl;
}

template <int AMODE, int BMODE>
__global__ __launch_bounds__(256, F32_WPS) void gemm_f32_mfma_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float lds[F_LDS_FLOATS];  // A tile | B tile; reused by the epilogue images
  float* const As = lds;
  float* const Bs = lds + FBK * FLD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int ntn = (p.N + FBN - 1) / FBN;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
  const int m0 = tm * FBM, n0 = tn * FBN;
  stamp_begin(p.stamp);
  int kbeg = 0, kend = p.K;
  if (p.split_k > 1) {
    const int steps = (p.K + FSPLIT_GRAN - 1) / FSPLIT_GRAN;
    const int per = (steps + p.split_k - 1) / p.split_k;
    kbeg = blockIdx.y * per * FSPLIT_GRAN;
    kend = min(p.K, kbeg + per * FSPLIT_GRAN);
  }
  const float* __restrict__ A = (const float*)p.A;
  const float* __restrict__ B = (const float*)p.B;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- per-thread staging maps -------------------------------------------------------------------------------------
  // k-contiguous storage: thread -> rows (tid >> 2) + 64 j (j = 0, 1), 4 consecutive k at (tid & 3) * 4: the 4 lanes of a
  //   row read its 64 contiguous bytes; the LDS writes (4 scalars per row into 4 k-rows) are conflict-free at stride 130.
  // x-contiguous storage (m or n along memory): thread -> k-row tid >> 4, columns 4 (tid & 15) + 64 j (j = 0, 1): 16 lanes
  //   read 256 contiguous bytes per load; two ds_write_b64 per float4 (rows are 8-byte aligned at stride 130).
  const int kc_row = tid >> 2, kc_k = (tid & 3) * 4;
  const int xc_k = tid >> 4, xc_x = 4 * (tid & 15);  // x-contiguous: 16 lanes x 16 B = 256 contiguous bytes of a k-row, twice (+64)
  float ra[8], rb[8];
  RowPix apix[2];
  if constexpr (AMODE == F_GATHER1) {
#pragma unroll
    for (int j = 0; j < 2; ++j) apix[j] = decompose_pixel(p.g, m0 + kc_row + 64 * j, p.M);
  }

  auto load_a = [&](int k0) __attribute__((always_inline)) {
    if constexpr (AMODE == F_KCONTIG) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = m0 + kc_row + 64 * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < p.M) v = *(const f32x4*)(A + (long)m * p.lda + k0 + kc_k);
        ra[4 * j] = v[0]; ra[4 * j + 1] = v[1]; ra[4 * j + 2] = v[2]; ra[4 * j + 3] = v[3];
      }
    } else if constexpr (AMODE == F_GATHER1) {
      const int tap = k0 / p.g.cper, c0 = k0 - tap * p.g.cper;  // a K step never straddles a tap (cper % 16 == 0)
      const int ky = tap / p.g.KW, kx = tap - ky * p.g.KW;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const long s = tap_src(p.g, apix[j], ky, kx);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (s >= 0) v = *(const f32x4*)(A + s + c0 + kc_k);
        ra[4 * j] = v[0]; ra[4 * j + 1] = v[1]; ra[4 * j + 2] = v[2]; ra[4 * j + 3] = v[3];
      }
    } else if constexpr (AMODE == F_XCONTIG) {
      const int k = k0 + xc_k;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = m0 + xc_x + 64 * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < p.M && k < kend) v = *(const f32x4*)(A + (long)k * p.lda + m);
        ra[4 * j] = v[0]; ra[4 * j + 1] = v[1]; ra[4 * j + 2] = v[2]; ra[4 * j + 3] = v[3];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {  // element e = tid + 256 i of the [16 k][128 m] tile image
        const int e = tid + 256 * i;
        ra[i] = simt_load_a<float>(p, m0 + (e & 127), k0 + (e >> 7), kend);
      }
    }
  };
  auto store_a = [&]() __attribute__((always_inline)) {
    if constexpr (AMODE == F_KCONTIG || AMODE == F_GATHER1) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) As[(kc_k + r) * FLD + kc_row + 64 * j] = ra[4 * j + r];
    } else if constexpr (AMODE == F_XCONTIG) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        *(f32x2*)&As[xc_k * FLD + xc_x + 64 * j] = f32x2{ra[4 * j], ra[4 * j + 1]};
        *(f32x2*)&As[xc_k * FLD + xc_x + 64 * j + 2] = f32x2{ra[4 * j + 2], ra[4 * j + 3]};
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i;
        As[(e >> 7) * FLD + (e & 127)] = ra[i];
      }
    }
  };
  auto load_b = [&](int k0) __attribute__((always_inline)) {
    if constexpr (BMODE == F_KCONTIG) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + kc_row + 64 * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < p.N) v = *(const f32x4*)(B + (long)n * p.ldb + k0 + kc_k);
        rb[4 * j] = v[0]; rb[4 * j + 1] = v[1]; rb[4 * j + 2] = v[2]; rb[4 * j + 3] = v[3];
      }
    } else if constexpr (BMODE == F_XCONTIG || BMODE == F_GATHER1_B) {
      const int k = k0 + xc_k;
      long row = (long)k * p.ldb;
      if constexpr (BMODE == F_GATHER1_B) {  // weight [Cout][taps][Cin] read as [k = (tap, cout)][n = cin]
        const int tap = k0 / p.g.cper, c = k - tap * p.g.cper;
        const int ky = tap / p.g.KW, kx = tap - ky * p.g.KW;
        row = (long)c * p.ldb + b_tap_offset(p, ky, kx);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + xc_x + 64 * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < p.N && k < kend) v = *(const f32x4*)(B + row + n);
        rb[4 * j] = v[0]; rb[4 * j + 1] = v[1]; rb[4 * j + 2] = v[2]; rb[4 * j + 3] = v[3];
      }
    } else if constexpr (BMODE == F_GATHER2) {  // k-rows are output pixels, columns n = tap * cper + c of the shifted input pixel
      const int k = k0 + xc_k;
      const RowPix px = decompose_pixel(p.g, k, kend);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + xc_x + 64 * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < p.N) {
          const int tap = (int)fd_div((uint32_t)n, p.g.fd_cper), c = n - tap * p.g.cper;
          const int ky = (int)fd_div((uint32_t)tap, p.g.fd_kw), kx = tap - ky * p.g.KW;
          const long s = tap_src(p.g, px, ky, kx);
          if (s >= 0) v = *(const f32x4*)(B + s + c);
        }
        rb[4 * j] = v[0]; rb[4 * j + 1] = v[1]; rb[4 * j + 2] = v[2]; rb[4 * j + 3] = v[3];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i;
        rb[i] = simt_load_b<float>(p, k0 + (e >> 7), n0 + (e & 127), kend);
      }
    }
  };
  auto store_b = [&]() __attribute__((always_inline)) {
    if constexpr (BMODE == F_KCONTIG) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Bs[(kc_k + r) * FLD + kc_row + 64 * j] = rb[4 * j + r];
    } else if constexpr (BMODE == F_XCONTIG || BMODE == F_GATHER1_B || BMODE == F_GATHER2) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        *(f32x2*)&Bs[xc_k * FLD + xc_x + 64 * j] = f32x2{rb[4 * j], rb[4 * j + 1]};
        *(f32x2*)&Bs[xc_k * FLD + xc_x + 64 * j + 2] = f32x2{rb[4 * j + 2], rb[4 * j + 3]};
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i;
        Bs[(e >> 7) * FLD + (e & 127)] = rb[i];
      }
    }
  };

  // ---- main loop: registers hold step k+1 while the MFMAs of step k run ----------------------------------------------
  const int arow = wr * 64 + (lane & 31), brow = wc * 64 + (lane & 31), khalf = lane >> 5;
  if (kbeg < kend) { load_a(kbeg); load_b(kbeg); }
  for (int k0 = kbeg; k0 < kend; k0 += FBK) {
    store_a();
    store_b();
    __syncthreads();
    if (k0 + FBK < kend) { load_a(k0 + FBK); load_b(k0 + FBK); }
#pragma unroll
    for (int kk = 0; kk < FBK / 2; ++kk) {
      const float* ak = As + (2 * kk + khalf) * FLD + arow;
      const float* bk = Bs + (2 * kk + khalf) * FLD + brow;
      const float a0 = ak[0], a1 = ak[32], b0 = bk[0], b1 = bk[32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D map of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5). Each wave
  // turns its four 32x32 blocks, one at a time, through a private [32][36] LDS image (the staging tiles are free by now)
  // into rows of 4 consecutive columns per lane, so the shared float4 epilogue (bias / activation / side operands / fp32
  // accumulate) applies; shapes whose N or leading dimensions are not multiples of 4 take the scalar form from the same image.
  float* img = lds + wave * (32 * 36);
  const bool vec_ok = !(p.N & 3) && !(p.ldc & 3) && !(p.ldc2 & 3) && !(p.ldmul & 3) && !(p.ldadd & 3);
  // The four blocks as four compile-time instantiations, NOT a `#pragma unroll` loop: when the shared epilogue grew by two
  // branches (act_after_add, round 3) the unroller gave up on this loop ("loop not unrolled"), acc[bi][bj] became runtime-indexed,
  // the 64 accumulators + staging registers moved to scratch (320 B per lane) and the exact mode ran 2.3x slower (73 -> 170 ms).
  auto block = [&](auto bi_c, auto bj_c) __attribute__((always_inline)) {
    constexpr int bi = decltype(bi_c)::value, bj = decltype(bj_c)::value;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r)  // (every accumulator index compile-time: a runtime-indexed one would live in scratch)
      img[((r & 3) + 8 * (r >> 2) + 4 * khalf) * 36 + (lane & 31)] = acc[bi][bj][r];
    __syncthreads();
    const int mb = m0 + wr * 64 + bi * 32, nb = n0 + wc * 64 + bj * 32;
    // (both loops stay rolled: unrolled, the kernel carried 16 copies of the float4 epilogue and 64 of the scalar one —
    //  115-159 KiB of code per instantiation against a 64 KiB instruction cache shared by the workgroups of two CUs)
    if (p.split_k > 1 || vec_ok) {
#pragma unroll 1
      for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i, c4 = (lane & 7) * 4;
        const int m = mb + row, n = nb + c4;
        if (m < p.M && n < p.N) {
          const f32x4 v = *(const f32x4*)&img[row * 36 + c4];
          if (p.split_k > 1) *(f32x4*)(p.ws + ((long)blockIdx.y * p.M + m) * p.N + n) = v;
          else gemm_epilogue4<float>(p, m, n, v);
        }
      }
    } else {
#pragma unroll 1
      for (int i = 0; i < 16; ++i) {
        const int row = (lane >> 5) + 2 * i, col = lane & 31;
        const int m = mb + row, n = nb + col;
        if (m < p.M && n < p.N) gemm_epilogue1<float>(p, m, n, img[row * 36 + col]);
      }
    }
  };
  block(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  block(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  block(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  block(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  stamp_end(p.stamp);
}

template <int AMODE>
static void f32_launch_b(const GemmParams& p, int bmode, dim3 grid, hipStream_t st) {
  switch (bmode) {
    case F_KCONTIG: hipLaunchKernelGGL((gemm_f32_mfma_kernel<AMODE, F_KCONTIG>), grid, dim3(256), 0, st, p); break;
    case F_XCONTIG: hipLaunchKernelGGL((gemm_f32_mfma_kernel<AMODE, F_XCONTIG>), grid, dim3(256), 0, st, p); break;
    case F_GATHER1_B: hipLaunchKernelGGL((gemm_f32_mfma_kernel<AMODE, F_GATHER1_B>), grid, dim3(256), 0, st, p); break;
    case F_GATHER2: hipLaunchKernelGGL((gemm_f32_mfma_kernel<AMODE, F_GATHER2>), grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((gemm_f32_mfma_kernel<AMODE, F_GENERIC>), grid, dim3(256), 0, st, p); break;
  }
}

// Is this problem worth a 128x128-tile matrix-core launch? The fusion head's B x 256 Linears (M = batch rows) and the
// 3-class heads stay on the SIMT kernel with its 64x64 tiles and float4 path.
bool gemm_f32_mfma_eligible(const GemmParams& p) {
  static const bool off = [] { const char* v = getenv("MMSA_F32_SIMT"); return v && atoi(v) != 0; }();
  if (off || p.c_gw > 0) return false;
  if (p.M < 64 || p.N < 64) return false;
  if (p.split_k > 1 && (!p.ws || (p.N % 4))) return false;
  // >= 2^27 multiply-adds: every encoder GEMM of the exact mode, none of the fusion head's B x 256 Linears (measured: the
  // head's 64 x 768 x 768-class problems took 11-15 us here against 10 us on the VALU kernel with its K split)
  return (double)p.M * p.N * p.K >= (double)(1 << 27);
}

int gemm_f32_mfma_launch(const GemmParams& pin, hipStream_t st) {
  GemmParams p = pin;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return MMSA_ERR_ARG;
  if (p.split_k < 1) p.split_k = 1;
  // Own K split when the caller lent a slab workspace (ws_bytes > 0): the engines size their request for the VALU kernel's
  // 64x64 tiles; with 128x128 tiles a BERT weight gradient (768 x 3072, K = 8192) is 144 workgroups for 256 CUs x 2 slots
  // and every forward / data gradient with N = 768 is 384. Aim at >= 2 workgroups per CU, slabs of >= 256 K values each.
  if (p.ws && p.ws_bytes > 0 && !(p.N % 4)) {
    const long tiles = (long)cdiv(p.M, FBM) * cdiv(p.N, FBN);
    int split = 1;
    static const int target = [] { const char* v = MMSA_EXP_ENV("MMSA_F32_TARGET_WGS"); const int x = v ? atoi(v) : 0; return x > 0 ? x : 768; }();
    if (tiles < target) {
      split = (int)((target + tiles - 1) / tiles);
      const int maxs = p.K / 256;
      if (split > maxs) split = maxs;
      if (split > 256) split = 256;
      while (split > 1 && (long)split * p.M * p.N * (long)sizeof(float) > p.ws_bytes) --split;
      if (split < 1) split = 1;
    }
    p.split_k = split;
  }
  const F32Plan pl = f32_plan(p);
  dim3 grid(cdiv(p.M, FBM) * cdiv(p.N, FBN), p.split_k, 1);
  switch (pl.amode) {
    case F_KCONTIG: f32_launch_b<F_KCONTIG>(p, pl.bmode, grid, st); break;
    case F_XCONTIG: f32_launch_b<F_XCONTIG>(p, pl.bmode, grid, st); break;
    case F_GATHER1: f32_launch_b<F_GATHER1>(p, pl.bmode, grid, st); break;
    default: f32_launch_b<F_GENERIC>(p, pl.bmode, grid, st); break;
  }
  MMSA_CHECK_LAUNCH();
  if (p.split_k > 1) {
    launch_splitk_reduce<float>(p, st);
    MMSA_CHECK_LAUNCH();
  }
  return MMSA_OK;
}
