// bf16 MFMA GEMM for gfx950 (v_mfma_f32_16x16x32_bf16), fp32 accumulate.
//
// One 256-thread workgroup (4 waves, 2x2) computes a 128x128 tile of C; each wave owns 64x64 = 4x4 MFMA tiles.
// K is consumed in steps of 64 through a double-buffered LDS image (2 x 32 KiB). Staging is LDS-DMA:
// `global_load_lds_dwordx4` (16 B per lane, 1 KiB per wave-instruction, no VGPR destination, no ds_write) for tile
// t+1 is issued before the MFMAs of tile t; the barrier that ends the step waits for it (2-phase structure of the
// CDNA4 guide, T3+T4 minimum form). The LDS image is lane-linear per wave-instruction, so the bank-conflict swizzle
// is applied to the per-lane SOURCE address and, identically, to the fragment reads.
// (GLDS = false keeps the register-staged variant: global_load_dwordx4 -> ds_write_b128, for A/B comparison;
//  select with MMSA_GEMM_REGSTAGE=1.)
//
// Operands come in two storage forms (gemm.h): k-contiguous tiles are [128 rows][64 k] read with ds_read_b128
// through a 16-byte XOR swizzle; k-major tiles are [64 k][128 cols] read with ds_read_b64_tr_b16 (the LDS
// transpose read of CDNA4) through a 32-byte XOR swizzle, so NN (data gradient) and TN (weight gradient) GEMMs
// need no transposed copies in HBM. The MFMA is issued with swapped operands (D = Bfrag x Afrag) so that each
// lane ends up with 4 consecutive columns of one C row and the epilogue stores 8/16 bytes per lane.
//
// Implicit-GEMM convolution gathers are resolved per 16-byte chunk at staging time: an invalid chunk (padding,
// stride hole, row/col/k past the end) is redirected to a page of zeros, so the main loop is branch-free.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <mutex>
#include <vector>
#include "gemm.h"
#include "gemm_epilogue.h"
#include "gemm_tile.h"

#define TBM 128
#define TBN 128
#define TBK 64
#define STAGE_BYTES 32768
#define TILE_BYTES 16384

__device__ uint4 g_mmsa_zero_page[32];  // 512 B of zeros

const void* mmsa_zero_page() {
  static const void* ptr = nullptr;
  if (!ptr) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_mmsa_zero_page)) != hipSuccess) return nullptr;
    ptr = p;
  }
  return ptr;
}

template <bool A_KM, bool B_KM, int GATHER, bool GLDS>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile mapping: XCD-aware (blocks b and b+8 share an L2) + n-fastest order inside an XCD's contiguous chunk
  const int ntn = (p.N + TBN - 1) / TBN, ntm = (p.M + TBM - 1) / TBM;
  const int nblk = ntm * ntn;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = bid / ntn, tn = bid - tm * ntn;
  const int m0 = tm * TBM, n0 = tn * TBN;

  // ---- K range of this split
  int kbeg = 0, kend = p.K;
  if (p.split_k > 1) {
    const int steps = (p.K + TBK - 1) / TBK;
    const int per = (steps + p.split_k - 1) / p.split_k;
    kbeg = blockIdx.y * per * TBK;
    kend = min(p.K, kbeg + per * TBK);
  }
  const int nk = kend > kbeg ? (kend - kbeg + TBK - 1) / TBK : 0;

  const bf16* __restrict__ Ab = (const bf16*)p.A;
  const bf16* __restrict__ Bb = (const bf16*)p.B;
  const bf16* zp = (const bf16*)p.zero_page;

  // ---- per-thread staging map (fixed over the K loop). Each thread moves 4 x 16 bytes per operand per K step.
  // GLDS: wave-instruction `i` of wave w writes the 1-KiB LDS piece pi = 4w + i linearly (lane l -> byte 16 l):
  //   k-contiguous tile: piece = 8 rows; lane -> row 8 pi + (l>>3), physical chunk l&7 = logical chunk ^ (row&7)
  //   k-major tile     : piece = 4 k-rows; lane -> k-row 4 pi + (l>>4), physical chunk l&15 (32-byte XOR swizzle)
  // register staging: k-contiguous chunk i -> row (tid>>3)+32i, chunk tid&7; k-major -> k-row (tid>>4)+16i, chunk tid&15
  int rowi[4], krowi[4], cci[4];
  int kc;
  if constexpr (GLDS) {
    kc = (lane & 7) ^ (lane >> 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pi = wave * 4 + i;
      rowi[i] = pi * 8 + (lane >> 3);
      krowi[i] = pi * 4 + (lane >> 4);
      const int pc = lane & 15;
      cci[i] = ((((pc >> 1) ^ kmajor_swz(krowi[i])) << 1) | (pc & 1));
    }
  } else {
    kc = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rowi[i] = (tid >> 3) + 32 * i;
      krowi[i] = (tid >> 4) + 16 * i;
      cci[i] = tid & 15;
    }
  }

  long a_rowoff[4];   // !A_KM plain: element offset of the row (or -1)
  RowPix a_pix[4];    // GATHER==1
  long a_coloff[4];   // A_KM: column (element) of this thread's chunk (or -1)
  long b_rowoff[4];
  long b_coloff[4];
  int b_ky[4], b_kx[4];  // GATHER==2: tap of the thread's column chunk
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (!A_KM) {
      const int m = m0 + rowi[i];
      if constexpr (GATHER == 1) a_pix[i] = decompose_pixel(p.g, m, p.M);
      else a_rowoff[i] = m < p.M ? (long)m * p.lda : -1;
    } else {
      const int col = m0 + cci[i] * 8;
      a_coloff[i] = col < p.M ? col : -1;
    }
    if constexpr (!B_KM) {
      const int n = n0 + rowi[i];
      b_rowoff[i] = n < p.N ? (long)n * p.ldb : -1;
    } else {
      const int col = n0 + cci[i] * 8;
      b_coloff[i] = -1;
      b_ky[i] = 0; b_kx[i] = 0;
      if constexpr (GATHER == 2) {
        if (col < p.N) {
          const uint32_t tap = fd_div((uint32_t)col, p.g.fd_cper);
          b_coloff[i] = col - (long)tap * p.g.cper;
          b_ky[i] = (int)fd_div(tap, p.g.fd_kw);
          b_kx[i] = (int)tap - b_ky[i] * p.g.KW;
        }
      } else {
        b_coloff[i] = col < p.N ? col : -1;
      }
    }
  }

  // source address of chunk i of the A / B tile for the K step starting at k0
  auto src_a = [&](int i, int k0) -> const bf16* {
    if constexpr (!A_KM) {
      const int kcol = k0 + kc * 8;
      const bool kvalid = kcol < kend;
      if constexpr (GATHER == 1) {
        const uint32_t tap = fd_div((uint32_t)k0, p.g.fd_cper);
        const int c0 = k0 - (int)tap * p.g.cper;
        const int ky = (int)fd_div(tap, p.g.fd_kw), kx = (int)tap - ky * p.g.KW;
        const long s = tap_src(p.g, a_pix[i], ky, kx);
        return (s >= 0 && kvalid) ? Ab + s + c0 + kc * 8 : zp;
      } else {
        return (a_rowoff[i] >= 0 && kvalid) ? Ab + a_rowoff[i] + kcol : zp;
      }
    } else {
      const int k = k0 + krowi[i];
      return (a_coloff[i] >= 0 && k < kend) ? Ab + (long)k * p.lda + a_coloff[i] : zp;
    }
  };
  auto src_b = [&](int i, int k0) -> const bf16* {
    if constexpr (!B_KM) {
      const int kcol = k0 + kc * 8;
      return (b_rowoff[i] >= 0 && kcol < kend) ? Bb + b_rowoff[i] + kcol : zp;
    } else if constexpr (GATHER == 2) {
      const RowPix px = decompose_pixel(p.g, k0 + krowi[i], kend);
      const long s = tap_src(p.g, px, b_ky[i], b_kx[i]);
      return (s >= 0 && b_coloff[i] >= 0) ? Bb + s + b_coloff[i] : zp;
    } else {
      long tapoff = 0;
      int kbase = k0;
      if constexpr (GATHER == 1) {  // weight [cout][tap][cin] read as rows k = (tap, cout)
        const uint32_t tap = fd_div((uint32_t)k0, p.g.fd_cper);
        kbase = k0 - (int)tap * p.g.cper;
        const int tky = (int)fd_div(tap, p.g.fd_kw);
        tapoff = b_tap_offset(p, tky, (int)tap - tky * p.g.KW);
      }
      const int k = k0 + krowi[i];
      return (b_coloff[i] >= 0 && k < kend) ? Bb + (long)(kbase + krowi[i]) * p.ldb + tapoff + b_coloff[i] : zp;
    }
  };

  bf16x8 ra[4], rb[4];  // register staging only

  // Plain (non-gather) operands go through buffer descriptors: the per-lane byte offset (voffset) is fixed for the
  // whole K loop and the K-step advance is a scalar (soffset), so a K step costs no VALU address arithmetic at all
  // (measured before: ~120 VALU per wave per K step for the 64-bit pointer selects, which bounded the kernel).
  // Rows / columns past the edge get voffset 0x80000000: out of the descriptor's range -> the DMA writes zeros.
  constexpr int OOB = (int)0x80000000;
  int voffA[4], voffB[4];
  __amdgpu_buffer_rsrc_t rsrcA, rsrcB;
  const bool use_srd = GLDS && GATHER == 0 && p.use_srd;
  if constexpr (GLDS && GATHER == 0) {
    rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
    rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (!A_KM) {
        const int m = m0 + rowi[i];
        voffA[i] = m < p.M ? (int)(((long)m * p.lda + kc * 8) * 2) : OOB;
      } else {
        const int col = m0 + cci[i] * 8;
        voffA[i] = col < p.M ? (int)(((long)krowi[i] * p.lda + col) * 2) : OOB;
      }
      if constexpr (!B_KM) {
        const int n = n0 + rowi[i];
        voffB[i] = n < p.N ? (int)(((long)n * p.ldb + kc * 8) * 2) : OOB;
      } else {
        const int col = n0 + cci[i] * 8;
        voffB[i] = col < p.N ? (int)(((long)krowi[i] * p.ldb + col) * 2) : OOB;
      }
    }
  }

  // GLDS: issue the 8 LDS-DMA loads of a K step into LDS stage `stage`
  auto stage_glds = [&](int stage, int k0) {
    unsigned char* sa = smem + stage * STAGE_BYTES + wave * 4096;
    unsigned char* sb = sa + TILE_BYTES;
    if constexpr (GLDS && GATHER == 0) {
      if (use_srd && k0 + TBK <= kend) {  // full K step: descriptor path (a partial last step takes the pointer path)
        const int soffA = A_KM ? (int)((long)k0 * p.lda * 2) : k0 * 2;
        const int soffB = B_KM ? (int)((long)k0 * p.ldb * 2) : k0 * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lptr_t)(sa + i * 1024), 16, voffA[i], soffA, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lptr_t)(sb + i * 1024), 16, voffB[i], soffB, 0, 0);
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)src_a(i, k0), (lptr_t)(sa + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)src_b(i, k0), (lptr_t)(sb + i * 1024), 16, 0, 0);
    }
  };
  auto load_regs = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *(const bf16x8*)src_a(i, k0);
      rb[i] = *(const bf16x8*)src_b(i, k0);
    }
  };
  auto store_regs = [&](int stage) {
    unsigned char* sa = smem + stage * STAGE_BYTES;
    unsigned char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (!A_KM) *(bf16x8*)(sa + kc_off(rowi[i], kc)) = ra[i];
      else *(bf16x8*)(sa + km_off(krowi[i], cci[i])) = ra[i];
      if constexpr (!B_KM) *(bf16x8*)(sb + kc_off(rowi[i], kc)) = rb[i];
      else *(bf16x8*)(sb + km_off(krowi[i], cci[i])) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    if constexpr (GLDS) stage_glds(0, kbeg);
    else { load_regs(kbeg); store_regs(0); }
  }
  __syncthreads();  // (hipcc drains the outstanding LDS-DMA with vmcnt(0) ahead of the barrier)

  // One K step on the compile-time LDS stage ST (the loop is unrolled by two so every fragment address is a loop-
  // invariant VGPR plus an immediate): issue all 16 fragment reads, then the LDS-DMA of the next tile into the other
  // stage (it must not sit in front of the reads in the memory queues), then the 32 MFMAs at raised priority.
  // Before this form the compiler recomputed ~100 VALU of addresses per step and drained lgkmcnt(0) three times.
  auto kstep = [&](auto st_c, int kt) {
    constexpr int ST = decltype(st_c)::value;
    const bool more = kt + 1 < nk;
    const unsigned char* sa = smem + ST * STAGE_BYTES;
    const unsigned char* sb = sa + TILE_BYTES;
    bf16x8 fa[2][4], fb[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[kk][i] = read_frag<A_KM>(sa, wm * 64 + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[kk][j] = read_frag<B_KM>(sb, wn * 64 + j * 16, kk, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      if constexpr (GLDS) stage_glds(ST ^ 1, kbeg + (kt + 1) * TBK);
      else load_regs(kbeg + (kt + 1) * TBK);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if constexpr (!GLDS) {
      if (more) store_regs(ST ^ 1);
    }
    __syncthreads();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    kstep(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) kstep(std::integral_constant<int, 1>{}, kt + 1);
  }

  // ---- epilogue: lane holds C[m = ..+(lane&15)][n = ..+4*(lane>>4)+r]
  const int r16 = lane & 15, g = lane >> 4;
  f32x4 bias4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + 4 * g;
    bias4[j] = (p.bias && p.split_k <= 1 && n < p.N) ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // fast path: plain bf16 store (+ bias) — most launches (dgrads, convolutions) — without the option branches
  if (p.split_k <= 1 && !p.C2 && p.act == MMSA_ACT_NONE && !p.mul && !p.add && !p.out_f32) {
    bf16* Cb = (bf16*)p.C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + r16;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + 4 * g;
        if (n < p.N) Vec4<bf16>::store(Cb + (long)m * p.ldc + n, acc[i][j] + bias4[j]);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + r16;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * g;
      if (n >= p.N) continue;
      if (p.split_k > 1) {
        *(f32x4*)(p.ws + ((long)blockIdx.y * p.M + m) * p.N + n) = acc[i][j];
      } else {
        gemm_epilogue4b<bf16>(p, m, n, acc[i][j], bias4[j]);
      }
    }
  }
}

size_t gemm_splitk_ws_bytes(int M, int N, int split_k) { return split_k > 1 ? (size_t)split_k * M * N * sizeof(float) : 0; }

template <bool A_KM, bool B_KM, int GATHER, bool GLDS>
static int launch_variant2(const GemmParams& p, hipStream_t st) {
  const int ntm = cdiv(p.M, TBM), ntn = cdiv(p.N, TBN);
  dim3 grid(ntm * ntn, p.split_k > 1 ? p.split_k : 1, 1);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<A_KM, B_KM, GATHER, GLDS>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_bf16_kernel<A_KM, B_KM, GATHER, GLDS>), grid, dim3(256), 2 * STAGE_BYTES, st, p);
  MMSA_CHECK_LAUNCH();
  if (p.split_k > 1) {
    launch_splitk_reduce<bf16>(p, st);
    MMSA_CHECK_LAUNCH();
  }
  return MMSA_OK;
}

static bool use_regstage() {
  const char* v = MMSA_EXP_ENV("MMSA_GEMM_REGSTAGE");
  return v && atoi(v) != 0;
}

template <bool A_KM, bool B_KM, int GATHER>
static int launch_variant(const GemmParams& p, hipStream_t st) {
  return use_regstage() ? launch_variant2<A_KM, B_KM, GATHER, false>(p, st) : launch_variant2<A_KM, B_KM, GATHER, true>(p, st);
}

// ---- optional live timing of the MFMA GEMM launches (bench.py roofline): HIP events on the launch stream ----------
struct ProfRec { hipEvent_t a, b; double flop, bytes; int M, N, K, akm, bkm, gather, split, wm, nj, ksplit, epi; };
// algorithmic HBM bytes of one problem: every distinct operand read once (an implicit-GEMM gather counts each source pixel
// once), the output written once, epilogue side operands / side outputs included
static double alg_bytes_of(const GemmParams& p) {
  const int taps = p.gather ? (p.g.KH * p.g.KW > 0 ? p.g.KH * p.g.KW : 1) : 1;
  double a = (double)p.M * p.K * 2, b = (double)p.N * p.K * 2;
  if (p.gather == 1) a = (double)p.M * (p.K / taps) * 2;
  if (p.gather == 2) b = (double)p.K * (p.N / taps) * 2;
  double t = a + b + (double)p.M * p.N * (p.out_f32 ? 4 : 2);
  if (p.mul) t += (double)p.M * p.N * 2;
  if (p.add) t += (double)p.M * p.N * 2;
  if (p.C2) t += (double)p.M * p.N * 2;
  if (p.out_f32 && p.accumulate) t += (double)p.M * p.N * 4;
  return t;
}
static unsigned long long* g_stamp_dev = nullptr;  // [record][2]: in-kernel {min start, max end} ticks
static size_t g_stamp_cap = 0;
static int g_prof_mode = 0;  // 0: HIP events around each launch; 1: in-kernel clock stamps (no events)
static std::vector<ProfRec> g_prof;
static size_t g_prof_used = 0;
static bool g_prof_on = false;
static std::recursive_mutex g_prof_mu;  // profiled launches are serialised (the encoders may be enqueued from two host threads)
static int g_prof_stride = 1, g_prof_phase = 0;
static long g_prof_index = 0;  // running launch index since the last gemm_prof_sample()

int gemm_prof_mode(int mode) {
  if (mode != 0 && mode != 1) return MMSA_ERR_ARG;
  g_prof_mode = mode;
  return MMSA_OK;
}
int gemm_prof_sample(int stride, int phase) {
  if (stride < 1 || phase < 0 || phase >= stride) return MMSA_ERR_ARG;
  g_prof_stride = stride; g_prof_phase = phase; g_prof_index = 0;
  return MMSA_OK;
}
static bool prof_take() {  // is this launch one of the sampled ones?
  const bool take = (g_prof_index % g_prof_stride) == g_prof_phase;
  ++g_prof_index;
  return take;
}

int gemm_prof_begin(int max_records) {
  if (max_records < 0) return MMSA_ERR_ARG;
  while (g_prof.size() < (size_t)max_records) {
    ProfRec r;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return MMSA_ERR_LAUNCH;
    r.flop = 0;
    g_prof.push_back(r);
  }
  g_prof_used = 0;
  g_prof_on = true;
  g_prof_stride = 1; g_prof_phase = 0; g_prof_index = 0;
  if (g_stamp_cap < (size_t)max_records) {
    if (g_stamp_dev) (void)hipFree(g_stamp_dev);
    g_stamp_dev = nullptr;
    if (hipMalloc((void**)&g_stamp_dev, (size_t)max_records * 2 * sizeof(unsigned long long)) != hipSuccess) return MMSA_ERR_LAUNCH;
    g_stamp_cap = (size_t)max_records;
  }
  {
    std::vector<unsigned long long> init((size_t)max_records * 2);
    for (int i = 0; i < max_records; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0ull; }
    if (max_records > 0 &&
        hipMemcpy(g_stamp_dev, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) != hipSuccess)
      return MMSA_ERR_LAUNCH;
  }
  return MMSA_OK;
}
static double g_prof_last_bytes = 0;  // algorithmic HBM bytes of the records of the latest gemm_prof_end
double gemm_prof_last_bytes() { return g_prof_last_bytes; }
// caller must have synchronized the stream(s); returns summed kernel time, algorithmic flop and launch count
int gemm_prof_end(double* total_ms, double* total_flop, long* launches) {
  g_prof_on = false;
  std::vector<double> dur(g_prof_used, 0.0);  // ms per record
  if (g_prof_mode == 1) {
    std::vector<unsigned long long> st(g_prof_used * 2);
    if (g_prof_used &&
        hipMemcpy(st.data(), g_stamp_dev, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
      return MMSA_ERR_LAUNCH;
    for (size_t i = 0; i < g_prof_used; ++i)  // s_memrealtime ticks at 100 MHz
      dur[i] = st[2 * i + 1] > st[2 * i] ? (double)(st[2 * i + 1] - st[2 * i]) * 1e-5 : 0.0;
  } else {
    for (size_t i = 0; i < g_prof_used; ++i) {
      float t = 0;
      if (hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b) != hipSuccess) return MMSA_ERR_LAUNCH;
      dur[i] = t;
    }
  }
  double ms = 0, fl = 0;
  g_prof_last_bytes = 0;
  for (size_t i = 0; i < g_prof_used; ++i) {
    ms += dur[i];
    fl += g_prof[i].flop;
    g_prof_last_bytes += g_prof[i].bytes;
  }
  *total_ms = ms; *total_flop = fl; *launches = (long)g_prof_used;
  if (const char* path = getenv("MMSA_PROF_DUMP")) {  // per-launch table for the profiles/ directory
    if (FILE* f = fopen(path, "w")) {
      fprintf(f, "M,N,K,a_kmajor,b_kmajor,gather,split_k,us,tflops,tile,ksplit,epilogue,alg_bytes\n");
      for (size_t i = 0; i < g_prof_used; ++i) {
        const ProfRec& r = g_prof[i];
        fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%.2f,%.1f,%dx%d,%d,%s,%.0f\n", r.M, r.N, r.K, r.akm, r.bkm, r.gather, r.split, dur[i] * 1e3,
                dur[i] > 0 ? r.flop / (dur[i] * 1e-3) / 1e12 : 0.0, r.wm * 64, r.nj * 16 * (r.wm ? 8 / r.wm : 0), r.ksplit,
                r.epi == 0 ? "store" : r.epi == 1 ? "bias/act" : "side", r.bytes);
      }
      fclose(f);
    }
  }
  g_prof_used = 0;
  return MMSA_OK;
}

static int gemm_bf16_launch_inner(const GemmParams& pin, hipStream_t st);

int gemm_prof_open(GemmParams& p, hipStream_t st) {
  if (!g_prof_on) return -1;
  std::lock_guard<std::recursive_mutex> lock(g_prof_mu);
  if (!g_prof_on || g_prof_used >= g_prof.size() || !prof_take()) return -1;
  const int slot = (int)g_prof_used++;
  ProfRec& r = g_prof[slot];
  r.flop = 2.0 * p.M * p.N * (double)p.K;
  r.bytes = alg_bytes_of(p);
  r.M = p.M; r.N = p.N; r.K = p.K; r.akm = p.a_kmajor; r.bkm = p.b_kmajor; r.gather = p.gather; r.split = p.split_k;
  r.epi = (p.mul || p.add) ? 2 : (p.bias || p.C2 || p.act != MMSA_ACT_NONE) ? 1 : 0;
  r.wm = 2; r.nj = 2; r.ksplit = p.split_k;
  if (g_prof_mode == 1) p.stamp = g_stamp_dev + 2 * slot;
  else (void)hipEventRecord(r.a, st);
  return slot;
}
void gemm_prof_close(int slot, hipStream_t st) {
  if (slot >= 0 && g_prof_mode == 0) (void)hipEventRecord(g_prof[slot].b, st);
}

int gemm_bf16_launch(const GemmParams& pin, hipStream_t st) {
  // batch-row problems: latency, not throughput (gemm_f32_tiny.hip); not part of the MFMA roofline record set
  if (gemm_bf16_tiny_eligible(pin)) return gemm_bf16_tiny_launch(pin, st);
  if (!g_prof_on) return gemm_bf16_launch_inner(pin, st);
  std::lock_guard<std::recursive_mutex> lock(g_prof_mu);
  if (!g_prof_on || g_prof_used >= g_prof.size() || !prof_take()) return gemm_bf16_launch_inner(pin, st);
  ProfRec& r = g_prof[g_prof_used];
  r.flop = 2.0 * pin.M * pin.N * (double)pin.K;
  r.bytes = alg_bytes_of(pin);
  r.M = pin.M; r.N = pin.N; r.K = pin.K; r.akm = pin.a_kmajor; r.bkm = pin.b_kmajor; r.gather = pin.gather; r.split = pin.split_k;
  r.epi = (pin.mul || pin.add) ? 2 : (pin.bias || pin.C2 || pin.act != MMSA_ACT_NONE) ? 1 : 0;
  g2_last_plan[0] = g2_last_plan[1] = g2_last_plan[2] = 0;
  if (g_prof_mode == 1) {  // in-kernel clock stamps: nothing is put into the queue around the launch
    GemmParams p = pin;
    p.stamp = g_stamp_dev + 2 * g_prof_used;
    ++g_prof_used;
    const int rc = gemm_bf16_launch_inner(p, st);
    r.wm = g2_last_plan[0]; r.nj = g2_last_plan[1]; r.ksplit = g2_last_plan[2];
    return rc;
  }
  (void)hipEventRecord(r.a, st);
  const int rc = gemm_bf16_launch_inner(pin, st);
  (void)hipEventRecord(r.b, st);
  r.wm = g2_last_plan[0]; r.nj = g2_last_plan[1]; r.ksplit = g2_last_plan[2];
  ++g_prof_used;
  return rc;
}

// grouped weight-gradient launch (gemm_mfma2.hip) with one roofline record for the whole group
int gemm_bf16_launch_group(const GemmParams* probs, float* const* colsum, int n, hipStream_t st) {
  static const bool v1_only = [] { const char* v = getenv("MMSA_GEMM_V1"); return v && atoi(v) != 0; }();
  static const bool no_group = mmsa_disabled("layer_group");
  if (v1_only || no_group || use_regstage()) return MMSA_ERR_UNSUPPORTED;
  if (!g_prof_on) return gemm2_launch_group(probs, colsum, n, st);
  std::lock_guard<std::recursive_mutex> lock(g_prof_mu);
  if (!g_prof_on || g_prof_used >= g_prof.size() || !prof_take()) return gemm2_launch_group(probs, colsum, n, st);
  ProfRec& r = g_prof[g_prof_used];
  r.flop = 0;
  r.bytes = 0;
  for (int g = 0; g < n; ++g) {
    r.flop += 2.0 * probs[g].M * probs[g].N * (double)probs[g].K;
    r.bytes += alg_bytes_of(probs[g]);
  }
  r.M = -n; r.N = 0; r.K = probs[0].K; r.akm = 1; r.bkm = 1; r.gather = probs[0].gather; r.split = 1; r.epi = 0;
  for (int g = 0; g < n; ++g) r.N += probs[g].M + probs[g].N;  // (grouped rows: N = the summed operand widths)
  if (g_prof_mode == 1) {
    GemmParams ps[GEMM_MAX_GROUPS];
    if (n > GEMM_MAX_GROUPS) return MMSA_ERR_UNSUPPORTED;
    for (int g = 0; g < n; ++g) ps[g] = probs[g];
    ps[0].stamp = g_stamp_dev + 2 * g_prof_used;
    const int rc = gemm2_launch_group(ps, colsum, n, st);
    r.wm = g2_last_plan[0]; r.nj = g2_last_plan[1]; r.ksplit = g2_last_plan[2];
    if (rc != MMSA_ERR_UNSUPPORTED) ++g_prof_used;
    return rc;
  }
  (void)hipEventRecord(r.a, st);
  const int rc = gemm2_launch_group(probs, colsum, n, st);
  if (rc == MMSA_ERR_UNSUPPORTED) return rc;  // nothing was launched: the record is reused by the fallback launches
  r.wm = g2_last_plan[0]; r.nj = g2_last_plan[1]; r.ksplit = g2_last_plan[2];
  (void)hipEventRecord(r.b, st);
  ++g_prof_used;
  return rc;
}

static int gemm_bf16_launch_inner(const GemmParams& pin, hipStream_t st) {
  GemmParams p = pin;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return MMSA_ERR_ARG;
  if (p.N % 4) return MMSA_ERR_ARG;
  // every 16-byte chunk must be wholly valid or wholly out of range
  if (p.K % 8 && !(p.a_kmajor && p.b_kmajor)) return MMSA_ERR_ARG;
  if (p.a_kmajor && (p.M % 8)) return MMSA_ERR_ARG;
  if (p.b_kmajor && (p.N % 8)) return MMSA_ERR_ARG;
  if ((p.lda % 8) || (p.ldb % 8)) return MMSA_ERR_ARG;
  if (p.gather && (p.g.cper % 8)) return MMSA_ERR_ARG;
  if (p.gather == 1 && (p.g.cper % TBK)) return MMSA_ERR_ARG;  // a K step must stay inside one tap
  if (p.split_k > 1 && !p.ws) return MMSA_ERR_ARG;
  if (p.split_k < 1) p.split_k = 1;
  {  // second-generation persistent kernel (gemm_mfma2.hip) for every plain (non-gather) shape it supports
    static const bool v1_only = [] { const char* v = getenv("MMSA_GEMM_V1"); return v && atoi(v) != 0; }();
    // ... except the HBM-bound 1x1 convolutions of the image encoder's first stages: independent per-wave streams (gemm_stream.hip)
    if (!v1_only && !use_regstage() && gemm_stream_eligible(p)) {
      g2_last_plan[0] = 1; g2_last_plan[1] = 0; g2_last_plan[2] = 1;  // (profile tables: tile "64x0" = the streaming kernel)
      return gemm_stream_launch(p, st);
    }
    if (!v1_only && !use_regstage() && gemm2_eligible(p)) {
      const size_t avail = p.ws ? (p.ws_bytes > 0 ? (size_t)p.ws_bytes : (size_t)p.split_k * p.M * p.N * sizeof(float)) : 0;
      return gemm2_launch(p, p.split_k > 1 || p.ws_bytes > 0 ? avail : 0, st);
    }
  }
  if (p.c_gw > 0) return MMSA_ERR_UNSUPPORTED;  // output row maps exist in the persistent kernel only
  p.zero_page = mmsa_zero_page();
  if (!p.zero_page) return MMSA_ERR_LAUNCH;
  {
    // byte extent of each operand view; the descriptor path needs every offset to fit a positive 32-bit int
    const long ea = p.a_kmajor ? ((long)(p.K - 1) * p.lda + p.M) * 2 : ((long)(p.M - 1) * p.lda + p.K) * 2;
    const long eb = p.b_kmajor ? ((long)(p.K - 1) * p.ldb + p.N) * 2 : ((long)(p.N - 1) * p.ldb + p.K) * 2;
    const char* off = MMSA_EXP_ENV("MMSA_GEMM_NO_SRD");
    p.use_srd = (p.gather == 0 && ea < 0x7FFFFFF0L && eb < 0x7FFFFFF0L && !(off && atoi(off))) ? 1 : 0;
    p.a_bytes = p.use_srd ? (unsigned)ea : 0;
    p.b_bytes = p.use_srd ? (unsigned)eb : 0;
    // timing-only diagnostic (results are wrong): zero-record descriptors make the range check drop every staging
    // load while the instruction stream, waits and barriers stay (cdna guide §7) — prices the memory side of the loop.
    if (const char* nl = MMSA_EXP_ENV("MMSA_GEMM_DBG_NOLOAD"))
      if (atoi(nl) && p.use_srd) { p.a_bytes = 0; p.b_bytes = 0; }
  }
  if (p.gather == 0) {
    if (!p.a_kmajor && !p.b_kmajor) return launch_variant<false, false, 0>(p, st);
    if (!p.a_kmajor && p.b_kmajor) return launch_variant<false, true, 0>(p, st);
    if (p.a_kmajor && p.b_kmajor) return launch_variant<true, true, 0>(p, st);
    return launch_variant<true, false, 0>(p, st);
  }
  if (p.gather == 1) {
    if (p.a_kmajor) return MMSA_ERR_ARG;
    if (!p.b_kmajor) return launch_variant<false, false, 1>(p, st);
    return launch_variant<false, true, 1>(p, st);
  }
  if (p.gather == 2) {
    if (!(p.a_kmajor && p.b_kmajor)) return MMSA_ERR_ARG;
    return launch_variant<true, true, 2>(p, st);
  }
  return MMSA_ERR_ARG;
}
