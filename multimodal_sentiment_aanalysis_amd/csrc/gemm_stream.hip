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
//          the tile's MFMAs and lane-swapped into the accumulator layout; the sum is taken in fp32, rounded once
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
        // The side operand was loaded in the STORE layout (8 consecutive columns per lane); the lane swap is an involution, so
        // the same two swaps bring it into the accumulator layout (4 columns of tile 2jp and of tile 2jp + 1 per lane); the sum is
        // taken there in fp32 and rounded once, then stored like the plain form. (Swapping the fp32 accumulators instead — four
        // v_permlane16_swap in a loop over the register index — was miscompiled by hipcc 7.2: one swap per pair survived.)
        const gs_i32x4 sv = __builtin_bit_cast(gs_i32x4, side[jp]);
        const auto t0 = __builtin_amdgcn_permlane16_swap((unsigned)sv[0], (unsigned)sv[2], false, false);
        const auto t1 = __builtin_amdgcn_permlane16_swap((unsigned)sv[1], (unsigned)sv[3], false, false);
        const gs_i32x2 sxi = {(int)t0[0], (int)t1[0]}, syi = {(int)t0[1], (int)t1[1]};
        const bf16x4 sx = __builtin_bit_cast(bf16x4, sxi), sy = __builtin_bit_cast(bf16x4, syi);
        const bf16x4 xb = {(bf16)(x[0] + (float)sx[0]), (bf16)(x[1] + (float)sx[1]), (bf16)(x[2] + (float)sx[2]), (bf16)(x[3] + (float)sx[3])};
        const bf16x4 yb = {(bf16)(y[0] + (float)sy[0]), (bf16)(y[1] + (float)sy[1]), (bf16)(y[2] + (float)sy[2]), (bf16)(y[3] + (float)sy[3])};
        const gs_i32x2 xi = __builtin_bit_cast(gs_i32x2, xb), yi = __builtin_bit_cast(gs_i32x2, yb);
        const auto s0 = __builtin_amdgcn_permlane16_swap((unsigned)xi[0], (unsigned)yi[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap((unsigned)xi[1], (unsigned)yi[1], false, false);
        const gs_i32x4 d = {(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
        *(gs_i32x4*)(cr + jp * 32) = d;
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

// ---- 3 x 3 convolutions of stage 1 (64 -> 64 channels, stride 1) as streams ----------------------------------------------------
// The same structure for the implicit-GEMM 3 x 3 convolution with 64 input and 64 output channels (forward: conv2 of the three
// stage-1 bottlenecks; data gradient: the same geometry with flipped taps): K = 9 taps x 64, N = 64. On the persistent kernel
// these are 9-step tiles whose step time does not depend on the tile width (latency-bound: ~1.2 us per K step): 42 us = 350
// TFLOP/s for 14.8 GFLOP and 51 MB. Here all nine 64 x 64 weight blocks sit in LDS (74 KiB forward, 144 KiB k-major for the data
// gradient), a wave owns 16 consecutive output pixels and fetches, per tap row ky, the three shifted 16-pixel windows straight from
// global memory (each input pixel is one 128-byte line: the 9-fold re-reads are L1 / L2 hits of lines the neighbouring taps and
// waves already touched; padding taps are dropped by the buffer descriptor and read zeros); the three tap rows of the NEXT tile are
// requested as soon as the current tile has consumed theirs. K order = tap-major, as the persistent kernel: bit-identical results.
struct Gs3Params {
  const bf16* A; const bf16* B; bf16* C;
  int M, ntiles;
  long ldb, ldc;
  long tap_off[9];   // element offset of tap (ky, kx) inside a B row (forward: tap * 64; data gradient: b_tap_offset)
  unsigned a_bytes;
  ConvGeom g;
  float* colstat;
  unsigned long long* stamp;
};

template <bool B_KM, int EPI>
__global__ __launch_bounds__(GS_THREADS) void gemm_stream3x3_kernel(Gs3Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NT = 4, C = 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g4 = lane >> 4;
  const bool stamped = (blockIdx.x & 7) == 0 || blockIdx.x == gridDim.x - 1;
  if (stamped) stamp_begin(p.stamp);
  // weight blocks, one per tap: forward [64 n][64 k] k-contiguous (8 KiB each), data gradient [64 k][128 (64 used) n] k-major (16 KiB)
  if constexpr (!B_KM) {
    for (int c = tid; c < 9 * 64 * 8; c += GS_THREADS) {
      const int tap = c >> 9, row = (c >> 3) & 63, kc = c & 7;
      *(bf16x8*)(smem + tap * 8192 + kc_off(row, kc)) = *(const bf16x8*)(p.B + (long)row * p.ldb + p.tap_off[tap] + kc * 8);
    }
  } else {
    for (int c = tid; c < 9 * 64 * 8; c += GS_THREADS) {
      const int tap = c >> 9, k = (c >> 3) & 63, nc = c & 7;
      *(bf16x8*)(smem + tap * 16384 + km_off(k, nc)) = *(const bf16x8*)(p.B + (long)k * p.ldb + p.tap_off[tap] + nc * 8);
    }
  }
  __syncthreads();
  // A workgroup owns a CONTIGUOUS run of tiles and deals them to its waves round robin (128 consecutive pixels per step: the
  // neighbouring image rows its taps reach are the lines the same CU touched a step earlier — L1 / L2 hits). Dealing tiles across
  // the whole grid, as the 1 x 1 streams do, made every tap a miss in the XCD's 4 MiB L2 (each XCD walked the whole 25.7 MB input):
  // 462 MB of L2 fills per launch served by the Infinity Cache — 37 us instead of ~15.
  const int per_wg = (p.ntiles + gridDim.x - 1) / gridDim.x;
  const int t_end = min(p.ntiles, (int)(blockIdx.x + 1) * per_wg);
  const int rw = blockIdx.x * per_wg + wave, nrw = 8;
  __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  constexpr int OOB = (int)0x80000000;
  unsigned opaque = 0;

  // the three shifted windows of tap row ky for the tile whose lane pixel is px: 3 taps x 2 k-halves x 16 bytes per lane. The
  // geometry is stride 1 (div == 1, checked by the launcher): the source pixel of tap (ky, kx) is the lane's own base pixel plus a
  // per-tap constant, and only the two range checks are per lane (gemm_tile.h's general tap_src carries the strided / divided
  // forms as divergent branches: 150 instructions per tap, which made the first version of this kernel slower than the GEMM).
  const int kmul = p.g.kmul, SH = p.g.SH, SW = p.g.SW;
  auto load_row = [&](bf16x8 (&a)[6], const RowPix& px, int ky) __attribute__((always_inline)) {
    const int sy = px.yb + ky * kmul;
    const bool yok = px.img >= 0 && (unsigned)sy < (unsigned)SH;
    const int rowbase = ((px.img * SH + sy) * SW + px.xb) * C + 8 * g4;  // element offset of (img, sy, xb) + this lane's k chunk
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int sx = px.xb + kx * kmul;
      const bool ok = yok && (unsigned)sx < (unsigned)SW;
      const int vo = ok ? (rowbase + kx * kmul * C) * 2 : OOB;
      a[2 * kx] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, vo, 0, 0));
      a[2 * kx + 1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, vo, 64, 0));
    }
  };
  f32x4 acc[NT];
  // the 8 weight fragments of one tap (2 k-halves x 4 column tiles), read one tap AHEAD of the MFMAs that use them: left to itself
  // hipcc reads each fragment right before its MFMA (lgkmcnt(1) chains), and a wave then waits an LDS round trip (~100 cycles
  // under load) per 16-cycle MFMA — the first version of this kernel ran at 400 TFLOP/s for that reason
  auto read_tap = [&](bf16x8 (&b)[8], int tap) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (!B_KM) b[kk * NT + j] = read_frag<false>(smem + opaque + tap * 8192, j * 16, kk, lane);
        else b[kk * NT + j] = read_frag<true>(smem + tap * 16384, j * 16, kk, lane);
      }
  };
  auto mma_tap = [&](const bf16x8 (&b)[8], const bf16x8 (&a)[6], int kx) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[kk * NT + j], a[2 * kx + kk], acc[j], 0, 0, 0);
  };
  bf16x8 bq0[8], bq1[8];
  // tap row ky with its fragments double-buffered: on entry bq0 holds tap 3 ky; on exit bq0 holds tap 3 ky + 3 (when there is one)
  auto mma_row = [&](const bf16x8 (&a)[6], int ky) __attribute__((always_inline)) {
    read_tap(bq1, 3 * ky + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_tap(bq0, a, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_tap(bq0, 3 * ky + 2);
    __builtin_amdgcn_sched_barrier(0);
    mma_tap(bq1, a, 1);
    __builtin_amdgcn_sched_barrier(0);
    read_tap(bq1, ky < 2 ? 3 * ky + 3 : 0);  // (after the last row: tap 0 again, for the next tile)
    __builtin_amdgcn_sched_barrier(0);
    mma_tap(bq0, a, 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) bq0[i] = bq1[i];
  };
  f32x4 cs[EPI == 1 ? NT : 1], cq[EPI == 1 ? NT : 1];
  if constexpr (EPI == 1) {
#pragma unroll
    for (int j = 0; j < NT; ++j) { cs[j] = f32x4{0.f, 0.f, 0.f, 0.f}; cq[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }

  bf16x8 a0[6], a1[6], a2[6];
  int t = rw;
  if (t < t_end) {
    const RowPix px = decompose_pixel(p.g, t * 16 + r16, p.M);
    load_row(a0, px, 0); load_row(a1, px, 1); load_row(a2, px, 2);
  }
  read_tap(bq0, 0);
  for (; t < t_end; t += nrw) {
    if constexpr (!B_KM) asm volatile("" : "+v"(opaque));  // (keeps the weight-fragment reads inside the loop: see gemm_stream_kernel)
    const int tn = t + nrw;
    const bool more = tn < t_end;
    RowPix pn;
    if (more) pn = decompose_pixel(p.g, tn * 16 + r16, p.M);
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_row(a0, 0);
    if (more) load_row(a0, pn, 0);
    mma_row(a1, 1);
    if (more) load_row(a1, pn, 1);
    mma_row(a2, 2);
    if (more) load_row(a2, pn, 2);
    // epilogue: as gemm_stream_kernel (lane = one pixel, 8 consecutive channels after the lane swap)
    const int m = t * 16 + r16;
    const int col_in_pair = (g4 & 1) * 16 + ((g4 >> 1) << 3);
    bf16* cr = p.C + (long)m * p.ldc + col_in_pair;
#pragma unroll
    for (int jp = 0; jp < NT / 2; ++jp) {
      const f32x4 x = acc[2 * jp], y = acc[2 * jp + 1];
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
  if constexpr (EPI == 1) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { cs[j][r] = gs_row16_sum(cs[j][r]); cq[j][r] = gs_row16_sum(cq[j][r]); }
    __syncthreads();
    float* red = (float*)smem;  // [8 waves][2][64]
    if (r16 == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = j * 16 + 4 * g4;
        *(f32x4*)(red + ((long)wave * 2 + 0) * C + n) = cs[j];
        *(f32x4*)(red + ((long)wave * 2 + 1) * C + n) = cq[j];
      }
    }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += GS_THREADS) {
      float sum = 0.f;
      for (int w = 0; w < 8; ++w) sum += red[(long)w * 2 * C + i];
      p.colstat[(long)blockIdx.x * 2 * C + i] = sum;
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
// the 3 x 3 form: gather = 1 with a 3 x 3 stride-1 geometry over 64 channels, 64 output channels, plain store (+ statistics)
static bool gs3_eligible(const GemmParams& p) {
  if (mmsa_disabled("stream3x3") || getenv("MMSA_G2_NJ")) return false;
  if (p.gather != 1 || p.a_kmajor || p.c_gw > 0 || p.split_k > 1 || p.scale_a || p.out_f32) return false;
  if (p.bias || p.C2 || p.mul || p.add || p.act != MMSA_ACT_NONE || p.act_after_add) return false;
  if (p.N != 64 || p.K != 576 || p.g.cper != 64 || p.g.KH != 3 || p.g.KW != 3 || p.g.div != 1 || p.g.src_pix_stride != 64) return false;
  if (p.M < 4096 || (p.M % 16) || (p.ldc % 8) || (p.ldb % 8)) return false;
  if (p.colstat && (p.b_kmajor || !p.colstat_rows || p.colstat_cap < (long)2 * 1024 * p.N)) return false;
  const long ghw = (long)p.g.GH * p.g.GW;
  if (ghw <= 0 || (p.M % ghw)) return false;
  const long ea = (p.M / ghw) * p.g.SH * p.g.SW * p.g.src_pix_stride * 2;
  if (ea >= 0x7FFFFFF0L) return false;
  return true;
}

bool gemm_stream_eligible(const GemmParams& p) {
  if (gs3_eligible(p)) return true;
  if (mmsa_disabled("stream1x1") || getenv("MMSA_G2_NJ")) return false;  // (a forced tile shape addresses the persistent kernel)
  if (p.gather || p.a_kmajor || p.c_gw > 0 || p.split_k > 1 || p.scale_a || p.out_f32) return false;
  if (p.bias || p.C2 || p.mul || p.act != MMSA_ACT_NONE || p.act_after_add) return false;
  if (p.add && p.colstat) return false;
  if (p.M < 4096 || (p.M % 16)) return false;
  // (K = 512 streams at 4.1-4.5 TB/s on the persistent kernel already; K = 192 = the stem's padded im2col matrix, N = 64)
  if (!(p.K == 64 || p.K == 128 || p.K == 256 || (p.K == 192 && p.N == 64))) return false;
  if (!(p.N == 64 || p.N == 128 || p.N == 256 || p.N == 512)) return false;
  if ((long)p.N * p.K > 65536 || (p.N == 256 && p.K == 256)) return false;
  if ((p.lda % 8) || (p.ldb % 8) || (p.ldc % 8) || (p.add && (p.ldadd % 8))) return false;
  if (((long)p.M * p.lda) >= 0x3FFFFFFFL * 2 || ((long)p.M * p.ldc) >= 0x3FFFFFFFL * 2) return false;
  if (p.colstat && (!p.colstat_rows || p.colstat_cap < (long)2 * 1024 * p.N)) return false;  // (room for up to 1024 partial rows)
  // with the statistics a wave keeps 16 more registers per column tile: measured on MI355X (profiles/r04_*) the stream wins for the
  // narrow outputs and for N = 256 at K = 64 (25.6 against 31.2 us) and loses where the columns must be split over 2-4 wave groups
  // that each re-read A (N = 512: 24.6 against 22.6 us) or the register budget halves the occupancy (N = 128, K = 256: 41.9 / 40.6)
  if (p.colstat && !(p.N == 64 || (p.N == 256 && p.K == 64) || (p.N == 128 && p.K <= 128))) return false;
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
    case 6:  // K = 192: the stem convolution's zero-padded 7 x 7 x 3 = 147 im2col columns (N = 64 only)
      if constexpr (NT == 4) return gs_launch_nk<NT, 6>(gp, p, lds, st);
      else return MMSA_ERR_UNSUPPORTED;
    case 8: return gs_launch_nk<NT, 8>(gp, p, lds, st);
    default: return MMSA_ERR_UNSUPPORTED;
  }
}

template <bool B_KM, int EPI>
static int gs3_launch_t(const Gs3Params& gp, size_t lds, int* rows_out, hipStream_t st) {
  static int occ = 0;
  static std::mutex mu;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!occ) {
      (void)hipFuncSetAttribute((const void*)gemm_stream3x3_kernel<B_KM, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
      int o = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)gemm_stream3x3_kernel<B_KM, EPI>, GS_THREADS, lds) != hipSuccess || o < 1) o = 1;
      occ = o > 2 ? 2 : o;  // (74 KiB of weights: up to two workgroups per CU; 144 KiB: one) — every workgroup of the grid must be resident
    }
  }
  int grid = gs_cus() * occ;
  if ((long)grid * 8 > gp.ntiles) grid = (gp.ntiles + 7) / 8;
  if (rows_out) *rows_out = grid;
  hipLaunchKernelGGL((gemm_stream3x3_kernel<B_KM, EPI>), dim3(grid), dim3(GS_THREADS), lds, st, gp);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}

static int gs3_launch(const GemmParams& p, hipStream_t st) {
  if (p.colstat_rows) *p.colstat_rows = 0;
  Gs3Params gp;
  gp.A = (const bf16*)p.A; gp.B = (const bf16*)p.B; gp.C = (bf16*)p.C;
  gp.M = p.M; gp.ntiles = p.M / 16; gp.ldb = p.ldb; gp.ldc = p.ldc;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) gp.tap_off[ky * 3 + kx] = p.b_kmajor ? b_tap_offset(p, ky, kx) : (long)(ky * 3 + kx) * 64;
  gp.a_bytes = (unsigned)((p.M / ((long)p.g.GH * p.g.GW)) * p.g.SH * p.g.SW * p.g.src_pix_stride * 2);
  gp.g = p.g;
  gp.colstat = p.colstat;
  gp.stamp = p.stamp;
  if (!p.b_kmajor) {
    const size_t lds = 9 * 8192;
    if (p.colstat) {
      int rows = 0;
      const int rc = gs3_launch_t<false, 1>(gp, lds, &rows, st);
      if (rc == MMSA_OK) *p.colstat_rows = rows;
      return rc;
    }
    return gs3_launch_t<false, 0>(gp, lds, nullptr, st);
  }
  return gs3_launch_t<true, 0>(gp, 9 * 16384, nullptr, st);
}

int gemm_stream_launch(const GemmParams& p, hipStream_t st) {
  if (gs3_eligible(p)) return gs3_launch(p, st);
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
