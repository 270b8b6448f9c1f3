// Streaming bf16 MFMA GEMM for the HBM-bound 1x1 convolutions of the image encoder's first two stages (round 4).
//
//   C[M][N] = A[M][K] · op(B)  (+ add[M][N])        K = 64 .. 512, N = 64 .. 512, N · K <= 65536, M = B·H·W = 50176 .. 200704
//
// Why a second kernel family: with K this small a tile of the persistent kernel (gemm_mfma2.hip) is ONE K step — load 48 KiB, 32
// MFMAs per wave, store 64 KiB — and its eight waves walk that sequence in lock step behind one barrier per step, so the loads of
// the next tiles, the matrix work and the store issue of a workgroup never overlap each other: these launches ran at 2.7-4.5
// TB/s of algorithmic bytes (profiles/r03_gemm_shapes_isolated.csv), the residual-add data gradients at 3.2-4.0 because their
// epilogue drains the queue once per tile. They are pure streams: every A row is read once, every C row written once, the
// weight matrix (<= 128 KiB) is the only reused operand.
// Here the weight matrix is staged into LDS ONCE per workgroup (swizzled images of gemm_tile.h), and after that single barrier
// the waves are independent streams: a wave owns 16-row tiles (all N columns, or a column half), loads the A fragments of its
// next tiles straight from global memory into registers in MFMA operand layout (16 rows x 64 B per instruction; no LDS round
// trip for an operand nobody shares), and multiplies with swapped operands — D = W_frag (A slot) x A_frag (B slot) — so that a
// lane ends up with ONE output row and 4 consecutive columns per 16-column tile: two v_permlane16_swap per tile pair give every
// lane 8 consecutive columns = one 16-byte store (CDNA4 guide T21), each store instruction covering 16 rows x 64 B. No barrier,
// no counted wait, no shared ring: loads (2-4 tiles ahead per wave), MFMAs and stores of different waves overlap by themselves.
//   EPI 0  plain bf16 store
//   EPI 1  + BatchNorm batch statistics of the STORED values (GemmParams::colstat): per-lane fp32 sums over all of a wave's
//          tiles, one 16-lane DPP reduction per kernel, waves combined through LDS: ONE partial row per workgroup
//          (256-512 rows for the finalize instead of M / 64 = 3136)
//   EPI 2  + add[M][N] (the skip connection's gradient in a 1x1 data gradient): the side operand's 16-byte loads are issued before
//          the tile's MFMAs; the sum is taken in fp32 (four fp32 lane swaps per tile pair), rounded once
// Reference slot: the encoder position of MML_ZYC/MultimodalModel.py:264-266 (ResNet-50 is not in the reference).
#include <mutex>
#include "gemm.h"
#include "gemm_tile.h"

#define GS_THREADS 512

typedef __attribute__((ext_vector_type(2))) int gs_i32x2;
typedef __attribute__((ext_vector_type(4))) int gs_i32x4;

__device__ __forceinline__ float gs_row16_sum(float v) {  // sum over the 16 lanes of a DPP row, left in every lane of the row
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
  return v;
}

struct GsParams {
  const bf16* A; const bf16* B; bf16* C; const bf16* add;
  int M, N, K;
  long lda, ldb, ldc, ldadd;
  int wn;            // column groups per workgroup (1, 2 or 4): a wave owns N / wn columns
  int ntiles;        // M / 16
  float* colstat;    // EPI 1: [gridDim.x][2][N]
  unsigned long long* stamp;
};

// NT = 16-column tiles per wave, KF = K / 32, B_KM: B stored [K][N] (data gradient) instead of [N][K] (forward)
template <int NT, int KF, bool B_KM, int EPI>
__global__ __launch_bounds__(GS_THREADS) void gemm_stream_kernel(GsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g4 = lane >> 4;
  const bool stamped = (blockIdx.x & 7) == 0 || blockIdx.x == gridDim.x - 1;  // (few same-address atomics: common.h stamp_*)
  if (stamped) stamp_begin(p.stamp);
  const int N = p.N, K = p.K;
  // ---- the weight matrix into LDS, once: k-contiguous [K/64][N][64] images (kc_off) or k-major [K/64][N/128][64][128] (km_off)
  if constexpr (!B_KM) {
    const int cpr = K >> 3;  // 16-byte chunks per row
    for (int c = tid; c < N * cpr; c += GS_THREADS) {
      const int row = c / cpr, kc = c - row * cpr;
      *(bf16x8*)(smem + (kc >> 3) * N * 128 + kc_off(row, kc & 7)) = *(const bf16x8*)(p.B + (long)row * p.ldb + kc * 8);
    }
  } else {
    const int cpr = N >> 3, nb = (N + 127) >> 7;
    for (int c = tid; c < K * cpr; c += GS_THREADS) {
      const int k = c / cpr, nc = c - k * cpr;
      *(bf16x8*)(smem + ((k >> 6) * nb + (nc >> 4)) * 16384 + km_off(k & 63, nc & 15)) = *(const bf16x8*)(p.B + (long)k * p.ldb + nc * 8);
    }
  }
  __syncthreads();  // the only barrier before the statistics hand-off: from here on every wave is its own stream

  const int wm_n = 8 / p.wn;                      // row-waves per workgroup
  const int wm = wave / p.wn, wn = wave - wm * p.wn;
  const int n_base = wn * NT * 16;                 // first column this wave owns
  const int rw = blockIdx.x * wm_n + wm, nrw = gridDim.x * wm_n;  // row-wave index / count: tiles rw, rw + nrw, ...
  const int koff = 8 * g4;                         // this lane's k offset inside a 32-wide fragment

  auto load_a = [&](bf16x8 (&af)[KF], int tile) __attribute__((always_inline)) {
    const bf16* src = p.A + (long)(tile * 16 + r16) * p.lda + koff;
#pragma unroll
    for (int kk = 0; kk < KF; ++kk) af[kk] = *(const bf16x8*)(src + kk * 32);
  };
  // The weight image never changes after the barrier, so hipcc hoists the k-contiguous image's plain ds_read_b128 fragment reads
  // out of the tile loop and keeps ALL NT x KF fragments in registers: fine (and faster: no LDS read in the stream) up to 16
  // fragments = 64 VGPRs, a spill disaster beyond (the 16 x 8 instantiation: 2.5 KB of scratch per lane). Larger images are read
  // through an offset the compiler cannot see through (re-made opaque once per tile): fragments are then read where they are used.
  constexpr bool HOIST = NT * KF <= 16;
  unsigned opaque = 0;
  auto b_frag = [&](int j, int kk) __attribute__((always_inline)) -> bf16x8 {
    const int n0 = n_base + j * 16;
    if constexpr (!B_KM) return read_frag<false>(smem + opaque + (kk >> 1) * N * 128, n0, kk & 1, lane);
    else return read_frag<true>(smem + ((kk >> 1) * ((N + 127) >> 7) + (n0 >> 7)) * 16384, n0 & 127, kk & 1, lane);
  };

  f32x4 cs[EPI == 1 ? NT : 1], cq[EPI == 1 ? NT : 1];  // EPI 1: per-lane column sums / sums of squares over all of this wave's tiles
  if constexpr (EPI == 1) {
#pragma unroll
    for (int j = 0; j < NT; ++j) { cs[j] = f32x4{0.f, 0.f, 0.f, 0.f}; cq[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }

  auto compute = [&](const bf16x8 (&af)[KF], int tile) __attribute__((always_inline)) {
    if constexpr (!B_KM && !HOIST) asm volatile("" : "+v"(opaque));
    const int m = tile * 16 + r16;
    // lane (r16, g4) after the swaps: even g4 -> tile j, columns 4 g4 .. 4 g4 + 7; odd g4 -> tile j + 1, columns 4 (g4 - 1) .. + 7
    const int col_in_pair = (g4 & 1) * 16 + ((g4 >> 1) << 3);
    bf16x8 side[EPI == 2 ? NT / 2 : 1];
    if constexpr (EPI == 2) {
      const bf16* ar = p.add + (long)m * p.ldadd + n_base + col_in_pair;
#pragma unroll
      for (int jp = 0; jp < NT / 2; ++jp) side[jp] = *(const bf16x8*)(ar + jp * 32);
    }
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KF; ++kk) {
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_frag(j, kk), af[kk], acc[j], 0, 0, 0);
    }
    bf16* cr = p.C + (long)m * p.ldc + n_base + col_in_pair;
#pragma unroll
    for (int jp = 0; jp < NT / 2; ++jp) {
      const f32x4 x = acc[2 * jp], y = acc[2 * jp + 1];
      if constexpr (EPI == 2) {
        f32x4 lo, hi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto s = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x[r]), __builtin_bit_cast(unsigned, y[r]), false, false);
          lo[r] = __builtin_bit_cast(float, s[0]);
          hi[r] = __builtin_bit_cast(float, s[1]);
        }
        const bf16x8 sv = side[jp];
        const bf16x8 o = {(bf16)(lo[0] + (float)sv[0]), (bf16)(lo[1] + (float)sv[1]), (bf16)(lo[2] + (float)sv[2]), (bf16)(lo[3] + (float)sv[3]),
                          (bf16)(hi[0] + (float)sv[4]), (bf16)(hi[1] + (float)sv[5]), (bf16)(hi[2] + (float)sv[6]), (bf16)(hi[3] + (float)sv[7])};
        *(bf16x8*)(cr + jp * 32) = o;
      } else {
        const bf16x4 xb = {(bf16)x[0], (bf16)x[1], (bf16)x[2], (bf16)x[3]};
        const bf16x4 yb = {(bf16)y[0], (bf16)y[1], (bf16)y[2], (bf16)y[3]};
        if constexpr (EPI == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float qx = (float)xb[r], qy = (float)yb[r];
            cs[2 * jp][r] += qx; cq[2 * jp][r] += qx * qx;
            cs[2 * jp + 1][r] += qy; cq[2 * jp + 1][r] += qy * qy;
          }
        }
        const gs_i32x2 xi = __builtin_bit_cast(gs_i32x2, xb), yi = __builtin_bit_cast(gs_i32x2, yb);
        const auto s0 = __builtin_amdgcn_permlane16_swap((unsigned)xi[0], (unsigned)yi[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap((unsigned)xi[1], (unsigned)yi[1], false, false);
        const gs_i32x4 d = {(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
        *(gs_i32x4*)(cr + jp * 32) = d;
      }
    }
  };

  // ---- the wave's stream: tiles rw, rw + nrw, ... with the A fragments of the next tile(s) in flight (two register sets; a
  // wave of the K = 64 problems keeps two tiles ahead through the second pair)
  {
    bf16x8 a0[KF], a1[KF];
    int t = rw;
    if (t < p.ntiles) load_a(a0, t);
    while (t < p.ntiles) {
      const int t1 = t + nrw;
      if (t1 < p.ntiles) load_a(a1, t1);
      compute(a0, t);
      if (t1 >= p.ntiles) break;
      const int t2 = t1 + nrw;
      if (t2 < p.ntiles) load_a(a0, t2);
      compute(a1, t1);
      t = t2;
    }
  }

  if constexpr (EPI == 1) {
    // column statistics: 16-lane row reduction once per kernel, then the row-waves of the workgroup through LDS (the weight
    // image is dead once every wave has left its stream), one partial row [2][N] per workgroup
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { cs[j][r] = gs_row16_sum(cs[j][r]); cq[j][r] = gs_row16_sum(cq[j][r]); }
    __syncthreads();
    float* red = (float*)smem;  // [wm][2][N]
    if (r16 == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n_base + j * 16 + 4 * g4;
        *(f32x4*)(red + ((long)wm * 2 + 0) * N + n) = cs[j];
        *(f32x4*)(red + ((long)wm * 2 + 1) * N + n) = cq[j];
      }
    }
    __syncthreads();
    for (int i = tid; i < 2 * N; i += GS_THREADS) {
      float s = 0.f;
      for (int w = 0; w < wm_n; ++w) s += red[(long)w * 2 * N + i];
      p.colstat[(long)blockIdx.x * 2 * N + i] = s;
    }
  }
  if (stamped) stamp_end(p.stamp);
}

// ---- host side -----------------------------------------------------------------------------------------------------------
static int gs_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

static size_t gs_lds_bytes(const GemmParams& p) {
  const size_t img = p.b_kmajor ? (size_t)(p.K / 64) * ((p.N + 127) / 128) * 16384 : (size_t)p.N * p.K * 2;
  const size_t red = (size_t)8 * 2 * p.N * sizeof(float);  // statistics hand-off (reuses the image's space)
  return img > red ? img : red;
}

// Which problems take this kernel: plain (no gather, no row map, no K split) bf16 NT / NN problems with K in {64, 128, 256},
// N in {64, 128, 256, 512}, N · K <= 65536 (the weight image fits 128 KiB of LDS), M a multiple of 16 and >= 4096 rows, epilogue = plain store (+ statistics) or + add. MMSA_DISABLE=stream1x1 turns
// it off (A/B and parity switch: the persistent kernel then takes these launches as before).
bool gemm_stream_eligible(const GemmParams& p) {
  if (mmsa_disabled("stream1x1") || getenv("MMSA_G2_NJ")) return false;  // (a forced tile shape addresses the persistent kernel)
  if (p.gather || p.a_kmajor || p.c_gw > 0 || p.split_k > 1 || p.scale_a || p.out_f32) return false;
  if (p.bias || p.C2 || p.mul || p.act != MMSA_ACT_NONE || p.act_after_add) return false;
  if (p.add && p.colstat) return false;
  if (p.M < 4096 || (p.M % 16)) return false;
  if (!(p.K == 64 || p.K == 128 || p.K == 256)) return false;  // (K = 512 streams at 4.1-4.5 TB/s on the persistent kernel already)
  if (!(p.N == 64 || p.N == 128 || p.N == 256 || p.N == 512)) return false;
  if ((long)p.N * p.K > 65536 || (p.N == 256 && p.K == 256)) return false;
  if ((p.lda % 8) || (p.ldb % 8) || (p.ldc % 8) || (p.add && (p.ldadd % 8))) return false;
  if (((long)p.M * p.lda) >= 0x3FFFFFFFL * 2 || ((long)p.M * p.ldc) >= 0x3FFFFFFFL * 2) return false;
  if (p.colstat && (!p.colstat_rows || p.colstat_cap < (long)2 * 1024 * p.N)) return false;  // (room for up to 1024 partial rows)
  return true;
}

template <int NT, int KF, bool B_KM, int EPI>
static int gs_launch_t(const GsParams& gp, size_t lds, int* rows_out, hipStream_t st) {
  static int occ = 0;
  static std::mutex mu;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!occ) {
      (void)hipFuncSetAttribute((const void*)gemm_stream_kernel<NT, KF, B_KM, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      int o = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)gemm_stream_kernel<NT, KF, B_KM, EPI>, GS_THREADS, lds) != hipSuccess || o < 1) o = 1;
      occ = o > 2 ? 2 : o;  // 8 or 16 waves per CU: every resident workgroup must be running for the strided tile deal to balance
    }
  }
  GsParams p = gp;
  int grid = gs_cus() * occ;
  const int wm_n = 8 / p.wn;
  if ((long)grid * wm_n > p.ntiles) grid = (p.ntiles + wm_n - 1) / wm_n;
  if (rows_out) *rows_out = grid;
  hipLaunchKernelGGL((gemm_stream_kernel<NT, KF, B_KM, EPI>), dim3(grid), dim3(GS_THREADS), lds, st, p);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

template <int NT, int KF>
static int gs_launch_nk(const GsParams& gp, const GemmParams& p, size_t lds, hipStream_t st) {
  if (!p.b_kmajor) {
    if (p.colstat) {
      if constexpr (NT <= 8) {
        int rows = 0;
        const int rc = gs_launch_t<NT, KF, false, 1>(gp, lds, &rows, st);
        if (rc == MMSA_OK) *p.colstat_rows = rows;
        return rc;
      } else {
        return MMSA_ERR_UNSUPPORTED;
      }
    }
    if (p.add) return gs_launch_t<NT, KF, false, 2>(gp, lds, nullptr, st);
    return gs_launch_t<NT, KF, false, 0>(gp, lds, nullptr, st);
  }
  if (p.colstat) return MMSA_ERR_UNSUPPORTED;  // (statistics are a forward feature: NT only)
  if (p.add) return gs_launch_t<NT, KF, true, 2>(gp, lds, nullptr, st);
  return gs_launch_t<NT, KF, true, 0>(gp, lds, nullptr, st);
}

template <int NT>
static int gs_launch_n(const GsParams& gp, const GemmParams& p, size_t lds, hipStream_t st) {
  switch (p.K / 32) {
    case 2: return gs_launch_nk<NT, 2>(gp, p, lds, st);
    case 4: return gs_launch_nk<NT, 4>(gp, p, lds, st);
    case 8: return gs_launch_nk<NT, 8>(gp, p, lds, st);
    default: return MMSA_ERR_UNSUPPORTED;
  }
}

int gemm_stream_launch(const GemmParams& p, hipStream_t st) {
  if (!gemm_stream_eligible(p)) return MMSA_ERR_UNSUPPORTED;
  if (p.colstat_rows) *p.colstat_rows = 0;
  GsParams gp;
  gp.A = (const bf16*)p.A; gp.B = (const bf16*)p.B; gp.C = (bf16*)p.C; gp.add = (const bf16*)p.add;
  gp.M = p.M; gp.N = p.N; gp.K = p.K; gp.lda = p.lda; gp.ldb = p.ldb; gp.ldc = p.ldc; gp.ldadd = p.ldadd;
  gp.ntiles = p.M / 16;
  gp.colstat = p.colstat;
  gp.stamp = p.stamp;
  const size_t lds = gs_lds_bytes(p);
  // a wave owns at most 16 column tiles (64 accumulator registers), 8 when it also keeps the column statistics (another 16
  // registers per tile): wider outputs are split over 2 or 4 column groups of waves (each re-reads the narrow A operand)
  if (p.colstat) {
    if (p.N == 512) { gp.wn = 4; return gs_launch_n<8>(gp, p, lds, st); }
    if (p.N == 256) { gp.wn = 2; return gs_launch_n<8>(gp, p, lds, st); }
  }
  if (p.N == 512) { gp.wn = 2; return gs_launch_n<16>(gp, p, lds, st); }
  gp.wn = 1;
  if (p.N == 256) return gs_launch_n<16>(gp, p, lds, st);
  if (p.N == 128) return gs_launch_n<8>(gp, p, lds, st);
  return gs_launch_n<4>(gp, p, lds, st);
}
