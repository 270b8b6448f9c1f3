// bf16 MFMA GEMM for gfx950, second generation: persistent workgroups, 3-stage LDS-DMA ring, counted waits.
//
// Why: the 128x128 / 2-stage kernel (gemm_mfma.hip) drains every LDS-DMA at each K step (`vmcnt(0)` + barrier), so a
// step can never be shorter than one memory round trip: ~550 TF/s on the BERT shapes (K = 768: 12 steps), and on the
// ResNet shapes with K = 64..576 (1-9 steps per tile) the load of a tile is not overlapped with anything but the
// co-resident workgroup.
// Here one 512-thread workgroup per CU (8 waves = 4(M) x 2(N), 64 x (BN/2) outputs per wave, BN = 64 / 96 / 128)
// walks a list of work items (tile x K-split) and treats the K steps of ALL its items as one stream:
//   * 3 LDS stages of 48 KiB (A 256x64 + B 128x64 bf16); the DMA of step u+2 is issued during step u, and the wait at
//     the top of a step is a COUNTED `s_waitcnt vmcnt(N)` that leaves the newer step's DMA (and the stores of the last
//     two epilogues) in flight, followed by ONE raw `s_barrier` (never `__syncthreads()`, whose fence would drain
//     the DMA queue) — CDNA4 guide 'Pipelining across barriers';
//   * the prefetch runs across item boundaries, so the epilogue stores of tile t overlap the loads of tile t+1 and a
//     K = 64 convolution streams at the HBM rate instead of the load latency;
//   * epilogue stores go through a buffer descriptor (out-of-range lanes are dropped by the hardware, no branches),
//     so the number of vector-memory operations per epilogue is a compile-time constant and the counted waits stay
//     exact. Epilogues that read memory (bias / residual / gelu' operands) drain once per tile instead.
//   * BN = 96 exists because BERT-base has N in {768, 2304, 3072} at M = 8192: 32 row tiles x N/96 column tiles is an
//     exact multiple of the 256 CUs for all three (128-wide tiles leave a quarter of the chip idle on the last round).
// LDS images, swizzles and fragment reads are those of gemm_tile.h (shared with the first-generation kernel).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <mutex>
#include <vector>
#include "gemm.h"
#include "gemm_epilogue.h"
#include "gemm_tile.h"

#define G2_BM 256
#define G2_BK 64
#define G2_A_BYTES 32768
#define G2_B_BYTES 16384
#define G2_STAGE (G2_A_BYTES + G2_B_BYTES)
#define G2_LDS (3 * G2_STAGE)
#ifndef G2_GROUP_COLSUM
#define G2_GROUP_COLSUM 0
#endif

#define G2_MAX_GROUPS GEMM_MAX_GROUPS
#define G2_ORDER_MAX 256
// timing-only ablation switches of the main loop (MMSA_G2_DBG bitmask, results are wrong): run-time branches inside the K loop, so
// they are compiled in experiment builds only (-DMMSA_EXPERIMENTS); the shipped kernels carry none of them
#ifdef MMSA_EXPERIMENTS
#define G2_DBG (s.dbg)
#else
#define G2_DBG 0
#endif
#ifdef G2_NO_SETPRIO  // experiment (tools/microbench/build_variant.sh -DG2_NO_SETPRIO): no priority raise around the MFMA blocks
#define G2_SETPRIO(x) ((void)0)
#else
#define G2_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
struct G2Sched {
  int ntm, ntn, ntiles;  // tile grid
  int split_k, per;      // K splits; K steps per split
  int nsteps;            // K / 64
  int items;             // ntiles * split_k; item = split * ntiles + tile (tiles of one K slice are neighbours)
  int fast;              // epilogue is a plain store (no memory reads): exact store counting
  FastDiv fd_ntiles, fd_ntn;
  unsigned c_bytes;      // byte extent of the output view (C, or the split-K slabs)
  int rb, cb;            // 2-D blocking of the tile order (rb x cb tiles = one XCD's round), 0 = row-major
  FastDiv fd_cb, fd_sbc;  // divisors: cb, super-blocks per row of super-blocks
  // grouped launch (weight gradients of one BERT layer / of one ResNet stage): independent TN problems with the same K share
  // one launch. BERT: 4 + 2 problems of 18-72 tiles add up to ~216 work items without any K split (no slabs, no reducer).
  // ResNet: the 5-11 same-shape 1x1 weight gradients of a stage (or its 2-5 same-geometry 3x3 ones, GATHER 2) with ONE common
  // K split: launched one by one each needed 16-60 slices of 4-13 K steps to fill the chip; together they need 2-8 slices of
  // 25-100 steps (C of a group then points at its slab region, and one grouped reducer launch follows).
  int ngroups;
  struct Group {
    const void* A; const void* B; void* C; float* colsum;  // colsum: optional sum_k A[k][m] (the bias gradient)
    int M, N; long lda, ldb, ldc;
    int tile_begin, ntn;
    unsigned a_bytes, b_bytes, c_bytes;
  } grp[G2_MAX_GROUPS];
  // grouped launches with <= G2_ORDER_MAX tiles: the walk order of the tiles (position -> tile index over all problems). The
  // host orders them so that the contiguous run of tiles one XCD works on at a time is a compact block of one problem (few
  // distinct A / B panels per K step in that XCD's L2) and a bias-gradient tile sits beside the weight-gradient tiles that
  // read the same dY panel. 0 = the tiles are walked in index order.
  int use_order;
  unsigned short order[G2_ORDER_MAX];
  unsigned long long* stamp;  // in-kernel timing record or null
  unsigned colstat_bytes;
  float* colstat;        // per-64-row-slice column sums / sums of squares of the stored bf16 output (GemmParams::colstat) or null
  int dbg;               // timing-only ablations (MMSA_G2_DBG bitmask; results are wrong): 1 no DMA, 2 no LDS reads, 4 no barrier, 8 no epilogue
};

typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;

// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt = simm16[15:14|3:0], expcnt [6:4] = 7 and lgkmcnt [11:8] = 15 mean
// "do not wait"). The builtin, not inline asm: hipcc's own wait-insertion pass then knows which vector-memory
// operations have completed (with asm it kept a stale "bias load pending" state across the loop back-edge and put a
// vmcnt(0) in front of a fragment read of every K step).
template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
}

// s_waitcnt lgkmcnt(N) only (vmcnt = 63, expcnt = 7: no wait); lgkmcnt is 4 bits
template <int N>
__device__ __forceinline__ void wait_lgkm() {
  static_assert(N >= 0 && N < 16, "lgkmcnt is 4 bits");
  __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8));
}

// Fragment reads as inline asm: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16
// builtin while an LDS-DMA is outstanding (it cannot prove the transpose read does not alias the DMA's LDS write),
// which drained the 3-stage ring twice per K step in every k-major variant. The asm forms carry no memory operand;
// ordering against the DMA is by the counted vmcnt + barrier, and their results are awaited with explicit
// lgkmcnt waits (each followed by sched_barrier(0), CDNA4 guide rule 18).
template <int IMM>
__device__ __forceinline__ bf16x8 lds_rd128(unsigned addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(IMM));
  return r;
}
template <int IMM>
__device__ __forceinline__ bf16x8 lds_rd_tr(unsigned addr) {  // k rows kb..kb+3 (lo) and kb+4..kb+7 (hi = +1024 B)
  typedef __attribute__((ext_vector_type(2))) int i32x2_;
  i32x2_ lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(IMM));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(IMM + 1024));
  typedef __attribute__((ext_vector_type(4))) int i32x4_;
  i32x4_ r = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, r);
}

// sum of v over the 16 lanes of a DPP row (lanes 16g .. 16g+15), left in every lane of the row: xor-1 and xor-2 exchanges inside
// the quads, then the mirror of each 8-lane half (quad 0 <-> quad 1) and of the whole row (lower 8 <-> upper 8)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
  return v;
}

// WM = wave rows: 4 -> 256-row tile, waves 4(M) x 2(N); 2 -> 128-row tile, waves 2(M) x 4(N) (twice the tiles for
// the mid-size convolutions: M = 12544 / 3136 gives only 49 / 13 row tiles of 256). NJ = 16-column MFMA tiles per wave:
// BN = NJ * 16 * (8 / WM).
// FP8: the operands are e4m3 bytes (NT only): a K step of 64 two-byte slots is 128 fp8 values in the SAME LDS image, every
// 16-byte fragment read feeds two v_mfma_f32_16x16x32_fp8_fp8 (bytes 0-7, 8-15: A and B share the slot -> k assignment, so
// every k meets its partner once), and the epilogue scales by scale_a * scale_b. Half the L2->LDS bytes per FLOP.
// GEN: the general epilogue (bias / activation / side output / gelu' and residual operands / fp32 read-modify-write) is compiled in.
// Launches whose epilogue is a plain store (s.fast) or a split-K slab store take the GEN = false instantiation: the general
// epilogue's ~60 extra live registers made the allocator spill LOOP-CARRIED values of kernels that never execute it — the grouped
// weight-gradient kernel went from 177 to 492 us when two more arrays were added to an epilogue it does not use — and every
// scratch reload in the K loop waits vmcnt(0), i.e. drains the LDS-DMA ring.
// EPI (the GEN parameter refined): 0 = no general epilogue (plain / slab stores only); 1 = the general epilogue with every
// option decided at run time; 2-5 = the same code with the options fixed at compile time for the launches that dominate the step:
//   2 bias (+ nothing else), 3 bias + GELU + gelu' side output (FFN-up), 4 (bias +) residual add, 5 (bias +) x gelu' factor.
// The fully general instantiation is ~27 000 instructions = 156 KiB of code against 12 KiB for EPI 0 (the shared instruction
// cache holds 64 KiB): every tile's epilogue streamed tens of KiB of mostly not-taken branches through it and evicted the K loop.
template <int WM, int NJ, bool A_KM, bool B_KM, int GATHER, bool FP8 = false, int EPI = 1>
__global__ __launch_bounds__(512, 2) void gemm2_kernel(GemmParams p, G2Sched s) {
  // The body is compiled in the device pass only: hipcc's HOST pass (ROCm 7.2) silently fails to instantiate this
  // template when it sees the body (no diagnostic, the launch stub stays an undefined symbol of the .so).
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WN = 8 / WM, BM = WM * 64, BN = NJ * 16 * WN;
  // NJ = 8 (WM = 4): the 256 x 256 tile. Half the L2->LDS bytes and 3/4 of the LDS fragment bytes per FLOP of the 256 x 128
  // tile; a stage is A 256x64 + B 256x64 = 64 KiB, so the ring has TWO stages (128 KiB) and the refill of a stage is issued in
  // one go right after the barrier that frees it (see the NJ == 8 main loop below).
  constexpr bool BIG = NJ == 8;
  static_assert(!BIG || (WM == 4 && GATHER == 0 && !FP8), "the 256 x 256 tile: plain NT / NN / TN problems only");
  constexpr int NSTG = BIG ? 2 : 3;
  constexpr int B_BYTES = BIG ? 32768 : G2_B_BYTES;
  constexpr int STAGE = G2_A_BYTES + B_BYTES;
  constexpr int NPA = BM / 64;  // A LDS-DMA pieces per wave per step (a piece = 1 KiB = 8 rows)
  constexpr int NLB = BIG ? 4 : ((BN <= 64 && !B_KM) ? 1 : 2);  // B LDS-DMA pieces per wave per step
  constexpr int NL = NPA + NLB;                      // LDS-DMA instructions per wave per step
  // store instructions per wave per plain epilogue, in units of 2 NJ: bf16 output = 1 unit (16 B per lane, row tiles
  // paired), fp32 output / split-K slabs = 2 units. The counted waits need the exact number.
  constexpr int NSU = BIG ? NJ : 2 * NJ;  // (the 256 x 256 tile counts its stores in units of NJ = 8 instructions)
  constexpr int OOB = (int)0x80000000;
  const int tid = threadIdx.x, lane = tid & 63;
  int lane_ = lane;  // the 256 x 256 loop makes this copy opaque once per K step: what is derived from it is recomputed, not kept
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WM == 4 ? wave >> 1 : wave >> 2, wn = WM == 4 ? wave & 1 : wave & 3;

  // ---- logical workgroup id: blocks b and b+8 share an XCD (L2); give each XCD a contiguous run of items
  const int G = gridDim.x;
  int lb;
  {
    const int q = G >> 3, r = G & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  if (lb >= s.items) return;
  stamp_begin(s.stamp);

  // ---- item decode (all scalar)
  constexpr bool GROUPS = A_KM && B_KM && (GATHER == 0 || GATHER == 2);  // only the TN instantiations carry the group code
  constexpr bool GCOLSUM = GROUPS && GATHER == 0 && G2_GROUP_COLSUM;     // ... and (optionally) the fused column sums
  struct Item { int m0, n0, kb, nk, sp, grp, tn; };
  auto decode = [&](int item) __attribute__((always_inline)) -> Item {
    Item it;
    it.grp = 0;
    if constexpr (GROUPS) {
      if (s.ngroups > 0) {
        int sp = 0, gt = item;  // K slice, tile index over all problems
        if (s.split_k > 1) {
          sp = (int)fd_div((uint32_t)item, s.fd_ntiles);
          gt = item - sp * s.ntiles;
        }
        if (s.use_order) gt = s.order[gt];
        int g = 0;
#pragma unroll
        for (int q = 1; q < G2_MAX_GROUPS; ++q)
          if (q < s.ngroups && gt >= s.grp[q].tile_begin) g = q;
        const int tile = gt - s.grp[g].tile_begin;
        const int tm = tile / s.grp[g].ntn, tn = tile - tm * s.grp[g].ntn;
        it.m0 = tm * BM; it.n0 = tn * BN; it.sp = sp; it.grp = g; it.tn = tn;
        it.kb = sp * s.per;
        it.nk = min(s.per, s.nsteps - it.kb);
        return it;
      }
    }
    const int sp = (int)fd_div((uint32_t)item, s.fd_ntiles);
    const int tile = item - sp * s.ntiles;
    int tm, tn;
    if (s.rb > 0) {  // super-blocks of rb x cb tiles (32 tiles = the run one XCD works on in a round): the XCD's L2 then
                     // holds rb A panels + cb B panels per K step instead of ~1 + ntn (B re-fetched by every XCD)
      const int sb = tile >> 5, within = tile & 31;
      const int r = (int)fd_div((uint32_t)within, s.fd_cb), c = within - r * s.cb;
      const int sbr = (int)fd_div((uint32_t)sb, s.fd_sbc), sbc = sb - sbr * (int)s.fd_sbc.d;
      tm = sbr * s.rb + r;
      tn = sbc * s.cb + c;
    } else {
      tm = (int)fd_div((uint32_t)tile, s.fd_ntn);
      tn = tile - tm * s.ntn;
    }
    it.m0 = tm * BM; it.n0 = tn * BN; it.sp = sp; it.tn = tn;
    it.kb = sp * s.per;
    it.nk = min(s.per, s.nsteps - it.kb);
    return it;
  };
  int total = 0;
  if (s.split_k == 1) total = ((s.items - lb + G - 1) / G) * s.nsteps;
  else
    for (int it = lb; it < s.items; it += G) total += decode(it).nk;

  // ---- per-lane staging map: relative byte offsets inside a tile (the tile origin and the K advance are scalar)
  // Implicit-GEMM gathers (gemm.h): GATHER 1 = the k-contiguous A rows are pixels and K = (tap, channel): the per-lane
  // source offset of a row changes with the tap, so it is recomputed per K step from the row's decomposed pixel;
  // GATHER 2 = the k-major B rows (k) are output pixels and the columns are (tap, channel): the pixel of a lane's k-row
  // is decomposed per K step. A K step never straddles a tap (cper % 64 == 0 is checked by the launcher).
  // Per-lane staging map, recomputed from the lane id whenever the loader enters an item (a handful of VALU per
  // item) instead of living in ~18 VGPRs through the main loop (the 256 x 128 variants were spilling):
  //   rc  = row (k-contiguous operand) or first column (k-major) of the lane's 16-byte chunk inside the tile
  //   rel = its byte offset inside the tile for the current leading dimension (tile origin and K advance are scalar)
  const int kc8 = ((lane & 7) ^ (lane >> 3)) * 8;
  // the loader's view of the problem (changes with the group of the item being loaded)
  int lM = p.M, lN = p.N;
  long llda = p.lda, lldb = p.ldb;
  auto krow_km = [&](int pi) __attribute__((always_inline)) -> int { return 4 * (pi & 15) + (lane >> 4); };
  auto col_km = [&](int krow) __attribute__((always_inline)) -> int {
    const int pc = lane & 15;
    return ((((pc >> 1) ^ kmajor_swz(krow)) << 1) | (pc & 1)) * 8;
  };
  auto rcA_of = [&](int i) __attribute__((always_inline)) -> int {
    const int pi = wave * NPA + i;
    if constexpr (!A_KM) return pi * 8 + (lane >> 3);
    else return (pi >> 4) * 128 + col_km(krow_km(pi));  // two [64][128] halves; piece = 4 k-rows of one half
  };
  auto relA_of = [&](int i) __attribute__((always_inline)) -> int {
    if constexpr (!A_KM) return (int)(((long)rcA_of(i) * llda + kc8) * 2);
    else return (int)(((long)krow_km(wave * NPA + i) * llda + rcA_of(i)) * 2);
  };
  auto krowB_of = [&](int i) __attribute__((always_inline)) -> int {
    if constexpr (BIG) return krow_km(wave * NLB + i);  // two [64][128] halves, piece = 4 k-rows of one half (as the A image)
    else return 4 * (wave * NLB + i) + (lane >> 4);
  };
  auto rcB_of = [&](int i) __attribute__((always_inline)) -> int {
    if constexpr (!B_KM) return (wave * NLB + i) * 8 + (lane >> 3);
    else if constexpr (BIG) return ((wave * NLB + i) >> 4) * 128 + col_km(krowB_of(i));
    else return col_km(krowB_of(i));
  };
  auto relB_of = [&](int i) __attribute__((always_inline)) -> int {
    if constexpr (!B_KM) return (int)(((long)rcB_of(i) * lldb + kc8) * 2);
    else return (int)(((long)krowB_of(i) * lldb + rcB_of(i)) * 2);
  };
  __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrcC =
      __builtin_amdgcn_make_buffer_rsrc(s.split_k > 1 ? (void*)p.ws : p.C, 0, s.c_bytes, 0x00020000);

  // ---- loader cursor
  int l_item = lb, l_kt = 0;
  Item L = decode(l_item);
  int voffA[NPA], voffB[NLB];
  RowPix a_pix[NPA];                       // GATHER 1: decomposed pixel of the lane's 4 A rows of the current tile
  int a_base[NPA];                         // GATHER 1, div == 1: element offset of (img, yb, xb) in the source (may be out of range)
  int b_ky[NLB], b_kx[NLB], b_coff[NLB];  // GATHER 2: tap and channel offset of the lane's B column chunks
  int l_grp = -1;
  // The 256 x 256 tile keeps ONE per-lane source offset per operand (two for a k-major operand: the column swizzle of a piece
  // depends on bit 3 of its k row) and adds the piece's own offset as a scalar (soffset): 8 per-lane offsets were 8 more
  // loop-carried VGPRs than the allocator had (spilled, and each reload drains the DMA ring). No range masks: a row / column
  // past the problem only feeds outputs the epilogue does not store, and the descriptors bound the reads.
  int vA[2] = {0, 0}, vB[2] = {0, 0};
  auto big_lane_offset = [&](bool km, long ld, int bit) __attribute__((always_inline)) -> int {
    const int ln = lane_;
    if (!km) return ((ln >> 3) * (int)ld + (((ln & 7) ^ (ln >> 3)) * 8)) * 2;
    const int pc = ln & 15, swz = (ln >> 4) | (bit << 2);
    const int colkm = ((((pc >> 1) ^ swz) << 1) | (pc & 1)) * 8;
    return ((ln >> 4) * (int)ld + colkm) * 2;
  };
  auto big_piece_soff = [&](bool km, long ld, int pi) __attribute__((always_inline)) -> int {  // scalar: pi is wave-uniform
    if (!km) return (int)((long)pi * 8 * ld * 2);
    return (int)(((long)(4 * (pi & 15)) * ld + (pi >> 4) * 128) * 2);
  };
  auto big_offsets = [&]() __attribute__((always_inline)) {  // (per DMA burst: the offsets are not kept across the K loop)
    vA[0] = big_lane_offset(A_KM, llda, 0);
    if constexpr (A_KM) vA[1] = big_lane_offset(true, llda, 1);
    vB[0] = big_lane_offset(B_KM, lldb, 0);
    if constexpr (B_KM) vB[1] = big_lane_offset(true, lldb, 1);
  };
  auto loader_setup = [&]() __attribute__((always_inline)) {
    if constexpr (GROUPS) {
      if (s.ngroups > 0 && L.grp != l_grp) {  // the loader enters another problem of the group
        l_grp = L.grp;
        lM = s.grp[l_grp].M; lN = s.grp[l_grp].N; llda = s.grp[l_grp].lda; lldb = s.grp[l_grp].ldb;
        rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)s.grp[l_grp].A, 0, s.grp[l_grp].a_bytes, 0x00020000);
        rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)s.grp[l_grp].B, 0, s.grp[l_grp].b_bytes, 0x00020000);
      }
    }
    if constexpr (BIG) return;  // big_offsets() at every DMA burst
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      if constexpr (GATHER == 1) {
        a_pix[i] = decompose_pixel(p.g, L.m0 + rcA_of(i), lM);
        a_base[i] = ((a_pix[i].img * p.g.SH + a_pix[i].yb) * p.g.SW + a_pix[i].xb) * (int)p.g.src_pix_stride;
      } else voffA[i] = (L.m0 + rcA_of(i) < lM) ? relA_of(i) : OOB;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int rcb = rcB_of(i);
      const bool ok = rcb < BN && L.n0 + rcb < lN;
      if constexpr (GATHER == 2) {
        const int col = L.n0 + rcb;
        const uint32_t tap = fd_div((uint32_t)col, p.g.fd_cper);
        b_coff[i] = ok ? col - (int)tap * p.g.cper : -1;
        b_ky[i] = (int)fd_div(tap, p.g.fd_kw);
        b_kx[i] = (int)tap - b_ky[i] * p.g.KW;
      } else {
        voffB[i] = ok ? relB_of(i) : OOB;
      }
    }
  };
  loader_setup();
  // LDS-DMA of the loader cursor's K step into `stage`: dma_begin (offsets), NL x dma_piece, dma_advance
  int d_soffA = 0, d_soffB = 0;
  auto dma_begin = [&]() __attribute__((always_inline)) {
    const int k0 = (L.kb + l_kt) * G2_BK;
    if constexpr (GATHER == 1) {
      const uint32_t tap = fd_div((uint32_t)k0, p.g.fd_cper);
      const int c0 = k0 - (int)tap * p.g.cper;
      const int ky = (int)fd_div(tap, p.g.fd_kw), kx = (int)tap - ky * p.g.KW;
      if (p.g.div == 1) {
        // forward convolutions and stride-1 data gradients: the tap only adds a scalar pixel delta to the row's base
        // offset (a_base, computed once per item); validity = two unsigned range checks. ~8 VALU per row piece instead
        // of ~25 with 64-bit address arithmetic — this block sits in front of the second-half MFMAs of every K step.
        const int dy = ky * p.g.kmul, dx = kx * p.g.kmul;
        const int tap_delta = (dy * p.g.SW + dx) * (int)p.g.src_pix_stride;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
          const bool ok = a_pix[i].img >= 0 && (unsigned)(a_pix[i].yb + dy) < (unsigned)p.g.SH &&
                          (unsigned)(a_pix[i].xb + dx) < (unsigned)p.g.SW;
          voffA[i] = ok ? (a_base[i] + tap_delta) * 2 + kc8 * 2 : OOB;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
          const long src = tap_src(p.g, a_pix[i], ky, kx);
          voffA[i] = src >= 0 ? (int)((src + kc8) * 2) : OOB;
        }
      }
      d_soffA = c0 * 2;
      if constexpr (B_KM)  // weight [cout][tap][cin] read as k-major rows k = (tap, cout): row (k % cper), tap offset
        d_soffB = (int)(((long)c0 * lldb + b_tap_offset(p, ky, kx) + L.n0) * 2);
      else
        d_soffB = (int)(((long)L.n0 * lldb + k0) * 2);
    } else if constexpr (GATHER == 2) {
      d_soffA = (int)(((long)k0 * llda + L.m0) * 2);
#pragma unroll
      for (int i = 0; i < NLB; ++i) {
        const RowPix px = decompose_pixel(p.g, k0 + krowB_of(i), p.K);
        if (p.g.div == 1) {  // forward-geometry gathers (every weight gradient): 32-bit offsets, unsigned range checks
          const int sy = px.yb + b_ky[i] * p.g.kmul, sx = px.xb + b_kx[i] * p.g.kmul;
          const bool ok = b_coff[i] >= 0 && (unsigned)sy < (unsigned)p.g.SH && (unsigned)sx < (unsigned)p.g.SW;
          voffB[i] = ok ? (((px.img * p.g.SH + sy) * p.g.SW + sx) * (int)p.g.src_pix_stride + b_coff[i]) * 2 : OOB;
        } else {
          const long src = tap_src(p.g, px, b_ky[i], b_kx[i]);
          voffB[i] = (src >= 0 && b_coff[i] >= 0) ? (int)((src + b_coff[i]) * 2) : OOB;
        }
      }
      d_soffB = 0;
    } else {
      d_soffA = A_KM ? (int)(((long)k0 * llda + L.m0) * 2) : (int)(((long)L.m0 * llda + k0) * 2);
      d_soffB = B_KM ? (int)(((long)k0 * lldb + L.n0) * 2) : (int)(((long)L.n0 * lldb + k0) * 2);
    }
  };
  auto dma_piece = [&](int stage, auto pc_c) __attribute__((always_inline)) {
    constexpr int pc = decltype(pc_c)::value;
    if constexpr (pc < NPA) {
      unsigned char* sa = smem + stage * STAGE + wave * (NPA * 1024) + pc * 1024;
      if constexpr (BIG) {
        const int pi = wave * NPA + pc;
        const int vo = A_KM ? ((((pi & 15) >> 1) & 1) ? vA[1] : vA[0]) : vA[0];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lptr_t)sa, 16, vo, d_soffA + big_piece_soff(A_KM, llda, pi), 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lptr_t)sa, 16, voffA[pc], d_soffA, 0, 0);
      }
    } else if constexpr (pc < NL) {
      unsigned char* sb = smem + stage * STAGE + G2_A_BYTES + wave * (NLB * 1024) + (pc - NPA) * 1024;
      if constexpr (BIG) {
        const int pi = wave * NLB + (pc - NPA);
        const int vo = B_KM ? ((((pi & 15) >> 1) & 1) ? vB[1] : vB[0]) : vB[0];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lptr_t)sb, 16, vo, d_soffB + big_piece_soff(B_KM, lldb, pi), 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lptr_t)sb, 16, voffB[pc - NPA], d_soffB, 0, 0);
      }
    }
  };
  auto dma_advance = [&]() __attribute__((always_inline)) {
    if (++l_kt == L.nk) {
      l_kt = 0;
      l_item += G;
      if (l_item < s.items) {
        L = decode(l_item);
        loader_setup();
      }
    }
  };
  auto issue = [&](int stage) __attribute__((always_inline)) {
    dma_begin();
    if constexpr (BIG) big_offsets();
    dma_piece(stage, std::integral_constant<int, 0>{});
    dma_piece(stage, std::integral_constant<int, 1>{});
    dma_piece(stage, std::integral_constant<int, 2>{});
    dma_piece(stage, std::integral_constant<int, 3>{});
    dma_piece(stage, std::integral_constant<int, 4>{});
    dma_piece(stage, std::integral_constant<int, 5>{});
    dma_piece(stage, std::integral_constant<int, 6>{});
    dma_piece(stage, std::integral_constant<int, 7>{});
    dma_advance();
  };

  // ---- compute cursor
  int c_item = lb, c_kt = 0;
  Item C = decode(c_item);
  // the epilogue's view of the problem (the compute cursor's group) and the optional column-sum accumulators
  int eM = p.M, eN = p.N;
  long eldc = p.ldc;
  float* e_colsum = nullptr;
  f32x4 accb[4];  // GROUPS: sum_k A[k][m] for the wave's 64 rows m (every lane row n holds the same value)
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute_setup = [&]() __attribute__((always_inline)) {
    if constexpr (GROUPS) {
      if (s.ngroups > 0) {
        const int g = C.grp;
        eM = s.grp[g].M; eN = s.grp[g].N; eldc = s.grp[g].ldc;
        e_colsum = C.tn == 0 ? s.grp[g].colsum : nullptr;
        rsrcC = __builtin_amdgcn_make_buffer_rsrc(s.grp[g].C, 0, s.grp[g].c_bytes, 0x00020000);
      }
    }
  };
  compute_setup();
  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int r16 = lane & 15, g4 = lane >> 4;
  float fp8_alpha = 1.f;
  if constexpr (FP8) fp8_alpha = (p.scale_a_rows ? 1.f : p.scale_a[0]) * p.scale_b[0];
  auto epilogue = [&]() __attribute__((always_inline)) -> int {  // returns the store units it issued (0: it drained the queue)
    const int r16 = lane_ & 15, g4 = lane_ >> 4;  // (= the outer ones; re-derived so that they are not live across the K loop)
    const int mb = C.m0 + wm * 64 + r16, nb = C.n0 + wn * (NJ * 16) + 4 * g4;
    if constexpr (FP8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float al = fp8_alpha;  // scale_a[0] * scale_b[0], or (per-token scales) scale_b[0] alone times the row's own scale
        if (p.scale_a_rows) al *= (mb + i * 16 < p.M) ? p.scale_a[mb + i * 16] : 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] *= al;
      }
    }
    if (s.split_k > 1) {  // raw fp32 partials -> slab sp
      const long slab = (long)C.sp * eM * eN;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mb + i * 16;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int n = nb + j * 16;
          const int vo = (m < eM && n < eN) ? (int)((slab + (long)m * eN + n) * 4) : OOB;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, acc[i][j]), rsrcC, vo, 0, 0);
        }
      }
      return BIG ? 4 : 2;
    }
    if constexpr (EPI == 6) {
      // mapped read-modify-write (GemmParams::c_rmw): per pair of row tiles, load what is stored at the mapped rows in the store
      // layout (16 bytes = 8 columns of one row per lane), bring it into the accumulator layout with the same lane swap the
      // store uses (an involution), add in fp32, then the plain bf16 store. Not a counted epilogue (three launches per step).
      const int mrow = mb + ((g4 & 1) << 4), ncol = C.n0 + wn * (NJ * 16) + ((g4 >> 1) << 3);
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        const int m = mrow + i * 16;
        const uint32_t img = fd_div((uint32_t)m, p.fd_c_ghw);
        const uint32_t rem = (uint32_t)m - img * (uint32_t)(p.c_gh * p.c_gw);
        const uint32_t yy = fd_div(rem, p.fd_c_gw), xx = rem - yy * (uint32_t)p.c_gw;
        const long rowoff = (long)img * p.c_imgpitch + (long)yy * p.c_rowpitch + (long)xx * p.c_colpitch;
        i32x4 old[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int n = ncol + j * 16;
          old[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrcC, (m < eM && n < eN) ? (int)((rowoff + n) * 2) : OOB, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        wait_vm<0>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int n = ncol + j * 16;
          const auto o0 = __builtin_amdgcn_permlane16_swap((unsigned)old[j][0], (unsigned)old[j][2], false, false);
          const auto o1 = __builtin_amdgcn_permlane16_swap((unsigned)old[j][1], (unsigned)old[j][3], false, false);
          const bf16x4 ox = __builtin_bit_cast(bf16x4, i32x2{(int)o0[0], (int)o1[0]});
          const bf16x4 oy = __builtin_bit_cast(bf16x4, i32x2{(int)o0[1], (int)o1[1]});
          f32x4 x = acc[i][j], y = acc[i + 1][j];
#pragma unroll
          for (int e = 0; e < 4; ++e) { x[e] += (float)ox[e]; y[e] += (float)oy[e]; }
          const bf16x4 xb = {(bf16)x[0], (bf16)x[1], (bf16)x[2], (bf16)x[3]};
          const bf16x4 yb = {(bf16)y[0], (bf16)y[1], (bf16)y[2], (bf16)y[3]};
          const i32x2 xi = __builtin_bit_cast(i32x2, xb), yi = __builtin_bit_cast(i32x2, yb);
          const auto s0 = __builtin_amdgcn_permlane16_swap((unsigned)xi[0], (unsigned)yi[0], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap((unsigned)xi[1], (unsigned)yi[1], false, false);
          const i32x4 d = {(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
          __builtin_amdgcn_raw_buffer_store_b128(d, rsrcC, (m < eM && n < eN) ? (int)((rowoff + n) * 2) : OOB, 0, 0);
        }
      }
      wait_vm<0>();
      return 0;
    }
    if (s.fast) {
      if (p.out_f32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = mb + i * 16;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int n = nb + j * 16;
            const int vo = (m < eM && n < eN) ? (int)(((long)m * eldc + n) * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, acc[i][j]), rsrcC, vo, 0, 0);
          }
        }
      } else {
        // bf16: pair the row tiles (i, i+1) and exchange 4-column groups between neighbouring lane rows
        // (v_permlane16_swap: g <-> g^1), so every lane stores 8 consecutive columns = 16 bytes: half the store
        // instructions of the 8-byte form (the epilogue is store-ISSUE bound: CDNA4 guide T21).
        // even g: tile i, columns 4g..4g+7; odd g: tile i+1, columns 4(g-1)..4(g-1)+7
        const int mrow = mb + ((g4 & 1) << 4), ncol = C.n0 + wn * (NJ * 16) + ((g4 >> 1) << 3);
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          const int m = mrow + i * 16;
          long rowoff = (long)m * eldc;
          if (p.c_gw > 0) {  // output row map (strided 1x1 data gradient): (img, y, x) -> scattered row
            const uint32_t img = fd_div((uint32_t)m, p.fd_c_ghw);
            const uint32_t rem = (uint32_t)m - img * (uint32_t)(p.c_gh * p.c_gw);
            const uint32_t yy = fd_div(rem, p.fd_c_gw), xx = rem - yy * (uint32_t)p.c_gw;
            rowoff = (long)img * p.c_imgpitch + (long)yy * p.c_rowpitch + (long)xx * p.c_colpitch;
          }
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int n = ncol + j * 16;
            const f32x4 x = acc[i][j], y = acc[i + 1][j];
            const bf16x4 xb = {(bf16)x[0], (bf16)x[1], (bf16)x[2], (bf16)x[3]};
            const bf16x4 yb = {(bf16)y[0], (bf16)y[1], (bf16)y[2], (bf16)y[3]};
            const i32x2 xi = __builtin_bit_cast(i32x2, xb), yi = __builtin_bit_cast(i32x2, yb);
            const auto s0 = __builtin_amdgcn_permlane16_swap((unsigned)xi[0], (unsigned)yi[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap((unsigned)xi[1], (unsigned)yi[1], false, false);
            const i32x4 d = {(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
            const int vo = (m < eM && n < eN) ? (int)((rowoff + n) * 2) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(d, rsrcC, vo, 0, 0);
          }
        }
      }
      if (!BIG && s.colstat && !p.out_f32) {
        // BatchNorm batch statistics of this tile's stored values (GemmParams::colstat): per lane the sums over its 4 row tiles,
        // then over the 16 lanes of its lane row (= the wave's 64 rows); the lanes with r16 == 0 hold 4 consecutive columns per
        // column tile and store them: 2 NJ store instructions = one more unit of the counted waits
        f32x4 cs[NJ], cq[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          cs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
          cq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float q = (float)(bf16)acc[i][j][r];
              cs[j][r] += q;
              cq[j][r] += q * q;
            }
#pragma unroll
          for (int r = 0; r < 4; ++r) { cs[j][r] = row16_sum(cs[j][r]); cq[j][r] = row16_sum(cq[j][r]); }
        }
        __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc((void*)s.colstat, 0, s.colstat_bytes, 0x00020000);
        const long prow = (long)(C.m0 >> 6) + wm;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int n = nb + j * 16;
          const bool ok = r16 == 0 && n < eN;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, cs[j]), rst, ok ? (int)(((prow * 2 + 0) * eN + n) * 4) : OOB, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, cq[j]), rst, ok ? (int)(((prow * 2 + 1) * eN + n) * 4) : OOB, 0, 0);
        }
        return 2;
      }
      if constexpr (GCOLSUM) {
        if (e_colsum) {  // bias gradient of this row panel: lanes of lane-row 0 of the wn = 0 waves hold it
          __amdgpu_buffer_rsrc_t rcs = __builtin_amdgcn_make_buffer_rsrc((void*)e_colsum, 0, eM * 4, 0x00020000);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = C.m0 + wm * 64 + i * 16 + r16;
            const int vo = (wn == 0 && g4 == 0 && m < eM) ? m * 4 : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, accb[i][0]), rcs, vo, 0, 0);
          }
          wait_vm<0>();  // not a counted epilogue
          return 0;
        }
      }
      return (p.out_f32 ? 2 : 1) * (BIG ? 2 : 1);
    }
    if constexpr (EPI == 0) {  // (not reached: the launcher gives a non-plain epilogue an instantiation that has one)
      wait_vm<0>();
      return 0;
    } else {
    // general epilogue (reads bias / side operands), then a full drain so the counted waits of the following steps
    // see an empty queue. All side-operand loads of the tile are issued up front, branch-free through buffer
    // descriptors (out-of-range lanes read 0), and awaited once: as a per-sub-tile load -> use -> store chain it cost
    // 16 dependent memory round trips per tile (FFN GEMMs ran at 420-470 TF/s against 750 for the plain store).
    // the options: run-time values in the general instantiation, compile-time constants in the specialised ones
    constexpr bool EG = EPI == 1;
    const bool o_rmw = EG && p.out_f32 && p.accumulate;                 // fp32 read-modify-write (rare)
    const bool o_c2gelu = EG ? (p.C2 && p.c2_gelu_grad) : EPI == 3;      // side output = gelu'(pre-activation), output = gelu
    const bool o_c2plain = EG && p.C2 && !p.c2_gelu_grad;               // side output = pre-activation
    const bool o_act = EG && p.act != MMSA_ACT_NONE;                    // any other activation
    const bool o_mulfac = EG ? p.mul_is_factor != 0 : true;
    const bool o_outf32 = EG && p.out_f32;
    const bool o_act_after = EG && p.act_after_add;
    if (!o_rmw) {
      // The 256 x 256 tile (NJ = 8) runs this epilogue in two passes of four column tiles each: with all eight at once the side
      // operands (64 VGPRs) and biases (32) on top of 128 accumulators and the loop's prefetched fragments spilled ~500 VGPRs.
      constexpr int JW = BIG ? 4 : NJ;
      int units = 0;
      auto part = [&](auto j0_c) __attribute__((always_inline)) {
      constexpr int J0 = decltype(j0_c)::value;
      const long cext = (long)(p.M - 1);
      f32x4 bias4[JW];
      // gelu' operand, or the residual when there is no gelu' operand (both: a second batch). The 256 x 256 tile takes no side
      // operands (the planner keeps such problems on the smaller tiles): 128 accumulators leave no room for them
      i32x2 side[BIG ? 1 : 4][BIG ? 1 : JW];
      const bool has_mul = !BIG && (EG ? p.mul != nullptr : EPI == 5), has_add = !BIG && (EG ? p.add != nullptr : EPI == 4);
      auto load_side = [&](const void* ptr, long ld) __attribute__((always_inline)) {
        if constexpr (!BIG) {
          __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, (int)((cext * ld + p.N) * 2), 0x00020000);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < JW; ++j) {
              const int m = mb + i * 16, n = nb + (J0 + j) * 16;
              side[i][j] = __builtin_amdgcn_raw_buffer_load_b64(rs, (m < p.M && n < p.N) ? (int)(((long)m * ld + n) * 2) : OOB, 0, 0);
            }
        }
      };
      {
        __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? p.N * 4 : 0, 0x00020000);
#pragma unroll
        for (int j = 0; j < JW; ++j) {
          const int n = nb + (J0 + j) * 16;
          bias4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, n < p.N ? n * 4 : OOB, 0, 0));
        }
      }
      if (has_mul) load_side(p.mul, p.ldmul);
      else if (has_add) load_side(p.add, p.ldadd);
      __builtin_amdgcn_sched_barrier(0);
      wait_vm<0>();  // every load of this epilogue (and every older DMA) has landed: from here on only stores are issued,
                     // so the epilogue ends with a known number of outstanding operations and needs no trailing drain
                     // (the drain made each tile wait for its own stores: +10 us on a bias-only 8192x3072x768 GEMM)
      __builtin_amdgcn_sched_barrier(0);
      __amdgpu_buffer_rsrc_t rc2 = __builtin_amdgcn_make_buffer_rsrc(p.C2, 0, (o_c2gelu || o_c2plain) ? (int)((cext * p.ldc2 + p.N) * 2) : 0, 0x00020000);
      const int mrow = mb + ((g4 & 1) << 4), ncol = C.n0 + wn * (NJ * 16) + ((g4 >> 1) << 3);
      auto unpack = [&](i32x2 v) __attribute__((always_inline)) -> f32x4 {
        const bf16x4 t = __builtin_bit_cast(bf16x4, v);
        return f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
      };
      auto store_pair = [&](__amdgpu_buffer_rsrc_t rs, long ld, int m, int n, f32x4 x, f32x4 y) __attribute__((always_inline)) {
        const bf16x4 xb = {(bf16)x[0], (bf16)x[1], (bf16)x[2], (bf16)x[3]};
        const bf16x4 yb = {(bf16)y[0], (bf16)y[1], (bf16)y[2], (bf16)y[3]};
        const i32x2 xi = __builtin_bit_cast(i32x2, xb), yi = __builtin_bit_cast(i32x2, yb);
        const auto s0 = __builtin_amdgcn_permlane16_swap((unsigned)xi[0], (unsigned)yi[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap((unsigned)xi[1], (unsigned)yi[1], false, false);
        const i32x4 d = {(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
        __builtin_amdgcn_raw_buffer_store_b128(d, rs, (m < p.M && n < p.N) ? (int)(((long)m * ld + n) * 2) : OOB, 0, 0);
      };
      // pass 1 (in place): + bias, side output, activation, gelu' factor
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
#pragma unroll
        for (int jj = 0; jj < JW; ++jj) {
          const int j = J0 + jj;
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[i + h][j] += bias4[jj];
          if (o_c2gelu) {  // side output = gelu'(pre-activation), from the same exp / erf as the activation
            f32x4 d0, d1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const GeluPair g0 = gelu_erf_both(acc[i][j][r]), g1 = gelu_erf_both(acc[i + 1][j][r]);
              acc[i][j][r] = g0.y; d0[r] = g0.dy;
              acc[i + 1][j][r] = g1.y; d1[r] = g1.dy;
            }
            store_pair(rc2, p.ldc2, mrow + i * 16, ncol + j * 16, d0, d1);
          } else {
            if (o_c2plain) store_pair(rc2, p.ldc2, mrow + i * 16, ncol + j * 16, acc[i][j], acc[i + 1][j]);
            if (o_act && !o_act_after) {
#pragma unroll
              for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i + h][j][r] = apply_act(acc[i + h][j][r], p.act);
            }
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (has_mul) {
              const f32x4 x = unpack(side[BIG ? 0 : i + h][BIG ? 0 : jj]);
              if (o_mulfac) acc[i + h][j] *= x;
              else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i + h][j][r] *= gelu_erf_grad(x[r]);
              }
            }
          }
        }
      }
      if (has_mul && has_add) {
        __builtin_amdgcn_sched_barrier(0);
        load_side(p.add, p.ldadd);
        __builtin_amdgcn_sched_barrier(0);
        wait_vm<0>();  // second batch (and the side-output stores before it) complete
        __builtin_amdgcn_sched_barrier(0);
      }
      // pass 2: + residual, store
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
#pragma unroll
        for (int jj = 0; jj < JW; ++jj) {
          const int j = J0 + jj;
          if (has_add) {
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[i + h][j] += unpack(side[BIG ? 0 : i + h][BIG ? 0 : jj]);
          }
          if (o_act && o_act_after) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[i + h][j][r] = apply_act(acc[i + h][j][r], p.act);
          }
          if (o_outf32) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int m = mb + (i + h) * 16, n = nb + j * 16;
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, acc[i + h][j]), rsrcC,
                                                     (m < p.M && n < p.N) ? (int)(((long)m * p.ldc + n) * 4) : OOB, 0, 0);
            }
          } else {
            store_pair(rsrcC, p.ldc, mrow + i * 16, ncol + j * 16, acc[i][j], acc[i + 1][j]);
          }
        }
      }
      // outstanding now: the C stores (bf16 pairs: 2 JW instructions; fp32: 4 JW) and, unless a second load batch was
      // awaited after them, the side-output stores (2 JW). In units of NSU (2 NJ; the 256 x 256 tile counts in units of NJ = 2 JW):
      units = (o_outf32 ? 2 : 1) + (((o_c2gelu || o_c2plain) && !(has_mul && has_add)) ? 1 : 0);
      };
      part(std::integral_constant<int, 0>{});
      if constexpr (BIG) {
        __builtin_amdgcn_sched_barrier(0);
        part(std::integral_constant<int, 4>{});  // (its wait_vm<0> also retires the first pass's stores)
      }
      return units > 2 ? 2 : units;  // the wait table covers these; under-reporting only makes a wait stricter
    } else if constexpr (!BIG) {  // fp32 accumulate into C without a split (rare): per-element read-modify-write
      f32x4 bias4[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = nb + j * 16;
        bias4[j] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mb + i * 16;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int n = nb + j * 16;
          if (n < p.N) gemm_epilogue4b<bf16>(p, m, n, acc[i][j], bias4[j]);
        }
      }
    }
    wait_vm<0>();
    return 0;
    }
  };

  // ---- main loop. One iteration = one K step (64 deep) of the stream, in two halves of 32:
  //   first half : issue the fragment reads of the step's second half, run the MFMAs of its first half
  //   mid-step   : lgkmcnt(0) (all reads of this stage are done) + counted vmcnt (next step's DMA has landed) + barrier
  //   second half: issue the fragment reads of the NEXT step's first half and the DMA that refills the stage just
  //                freed (step u+3), run the MFMAs of the second half, then the epilogue if the item ends here.
  // So every LDS read is issued one half-step ahead of the MFMAs that consume it (all 8 waves leave the barrier
  // together: without this the LDS burst of a step sat in front of its MFMAs, 1/3 of the step), and three steps of
  // DMA are in flight. Stage of step u = u mod 3 (run-time scalar added to loop-invariant per-lane addresses).
  // per-lane fragment addresses inside a stage (loop invariant). k-contiguous: one per kk (row tiles i are +2048 B);
  // k-major: one per row tile (kk is +8192 B, the upper 4 k rows +1024 B): the XOR swizzles depend on exactly those.
  constexpr int NFA = A_KM ? 4 : 2, NFB = B_KM ? NJ : 2;
  unsigned offA[NFA], offB[NFB];
  {
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;
    const int q = r16 >> 2, pp = r16 & 3;
    if constexpr (!A_KM) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) offA[kk] = lds0 + kc_off(wm * 64 + r16, kk * 4 + g4);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = (wm & 1) * 64 + i * 16 + 4 * pp;
        offA[i] = lds0 + (wm >> 1) * 16384 + km_off(8 * g4 + q, col >> 3) + ((pp & 1) << 3);
      }
    }
    if constexpr (!B_KM) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) offB[kk] = lds0 + G2_A_BYTES + kc_off(wn * (NJ * 16) + r16, kk * 4 + g4);
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = wn * (NJ * 16) + j * 16 + 4 * pp;
        offB[j] = lds0 + G2_A_BYTES + (col >> 7) * 16384 + km_off(8 * g4 + q, (col & 127) >> 3) + ((pp & 1) << 3);
      }
    }
  }
  constexpr int RD_HALF = (A_KM ? 8 : 4) + (B_KM ? 2 * NJ : NJ);  // LDS instructions of one read_half
  auto read_half = [&](auto kk_c, int stage, bf16x8(&fa)[4], bf16x8(&fb)[NJ]) __attribute__((always_inline)) {
    constexpr int kk = decltype(kk_c)::value;
    const unsigned so = (unsigned)(stage * STAGE);
    if constexpr (!A_KM) {
      const unsigned a = offA[kk] + so;
      fa[0] = lds_rd128<0>(a); fa[1] = lds_rd128<2048>(a); fa[2] = lds_rd128<4096>(a); fa[3] = lds_rd128<6144>(a);
    } else {
      fa[0] = lds_rd_tr<kk * 8192>(offA[0] + so); fa[1] = lds_rd_tr<kk * 8192>(offA[1] + so);
      fa[2] = lds_rd_tr<kk * 8192>(offA[2] + so); fa[3] = lds_rd_tr<kk * 8192>(offA[3] + so);
    }
    if constexpr (!B_KM) {
      const unsigned b = offB[kk] + so;
      fb[0] = lds_rd128<0>(b);
      if constexpr (NJ > 1) fb[1] = lds_rd128<2048>(b);
      if constexpr (NJ > 2) fb[2] = lds_rd128<4096>(b);
      if constexpr (NJ > 3) fb[3] = lds_rd128<6144>(b);
    } else {
      fb[0] = lds_rd_tr<kk * 8192>(offB[0] + so);
      if constexpr (NJ > 1) fb[1] = lds_rd_tr<kk * 8192>(offB[1] + so);
      if constexpr (NJ > 2) fb[2] = lds_rd_tr<kk * 8192>(offB[2] + so);
      if constexpr (NJ > 3) fb[3] = lds_rd_tr<kk * 8192>(offB[3] + so);
    }
  };
  auto mma1 = [&](const bf16x8& fbj, const bf16x8& fai, f32x4 c) __attribute__((always_inline)) -> f32x4 {
#ifdef G2_ABLATE_NO_MFMA  // timing-only build (tools/microbench/ablate_nomfma.sh): the loop without its matrix instructions;
    asm volatile("" ::"v"(fbj), "v"(fai));  // the fragments stay live, so the LDS reads are not dropped with them
    return c;
#endif
    if constexpr (FP8) {
      typedef __attribute__((ext_vector_type(2))) long i64x2_;
      const i64x2_ b2 = __builtin_bit_cast(i64x2_, fbj), a2 = __builtin_bit_cast(i64x2_, fai);
      c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b2[0], a2[0], c, 0, 0, 0);
      return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b2[1], a2[1], c, 0, 0, 0);
    } else {
      return __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbj, fai, c, 0, 0, 0);
    }
  };
  // column sums of A on the matrix pipe: D[n][m] = sum_k 1 * A[m][k] with an all-ones operand in place of the B fragment
  auto mma_colsum = [&](const bf16x8(&fa)[4]) __attribute__((always_inline)) {
    if constexpr (GCOLSUM) {
      if (e_colsum) {
        const bf16 one = (bf16)1.0f;
        const bf16x8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
        for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i], accb[i], 0, 0, 0);
      }
    }
  };
  auto mma_half = [&](const bf16x8(&fa)[4], const bf16x8(&fb)[NJ]) __attribute__((always_inline)) {
    G2_SETPRIO(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = mma1(fb[j], fa[i], acc[i][j]);
    mma_colsum(fa);
    G2_SETPRIO(0);
  };
  // the same 4 x NJ MFMAs with half of the LDS-DMA instructions of a K step spread between them. The refill of the
  // stage freed by the mid-step barrier of step u (= the DMA of step u+3) is issued in two parts: pieces 0..2 in the
  // second half of step u (PART 0), the rest in the first half of step u+1 (PART 1). Issued as one burst right after
  // the barrier by all 8 waves they queue in the address unit and every wave's MFMA stream stands still behind them.
  constexpr int NLA = (NL + 1) / 2;  // pieces of part 0
  auto mma_half_dma = [&](const bf16x8(&fa)[4], const bf16x8(&fb)[NJ], int stage, auto part_c) __attribute__((always_inline)) {
    constexpr int PART = decltype(part_c)::value;
    constexpr int P0 = PART == 0 ? 0 : NLA, P1 = PART == 0 ? NLA : NL;  // pieces [P0, P1)
    constexpr int GAP = (4 * NJ) / (P1 - P0 > 0 ? (P1 - P0) : 1);        // MFMAs between two DMA instructions
    if constexpr (PART == 0) dma_begin();
    __builtin_amdgcn_sched_barrier(0);
    G2_SETPRIO(1);
    auto one = [&](auto idx_c) __attribute__((always_inline)) {
      constexpr int idx = decltype(idx_c)::value;
      constexpr int i = idx / NJ, j = idx % NJ;
      acc[i][j] = mma1(fb[j], fa[i], acc[i][j]);
      if constexpr (idx % GAP == 1 % GAP && P0 + idx / GAP < P1) {
        __builtin_amdgcn_sched_barrier(0);
        dma_piece(stage, std::integral_constant<int, P0 + idx / GAP>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{});
    one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
    if constexpr (NJ > 1) {
      one(std::integral_constant<int, 4>{}); one(std::integral_constant<int, 5>{});
      one(std::integral_constant<int, 6>{}); one(std::integral_constant<int, 7>{});
    }
    if constexpr (NJ > 2) {
      one(std::integral_constant<int, 8>{}); one(std::integral_constant<int, 9>{});
      one(std::integral_constant<int, 10>{}); one(std::integral_constant<int, 11>{});
    }
    if constexpr (NJ > 3) {
      one(std::integral_constant<int, 12>{}); one(std::integral_constant<int, 13>{});
      one(std::integral_constant<int, 14>{}); one(std::integral_constant<int, 15>{});
    }
    mma_colsum(fa);
    G2_SETPRIO(0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PART == 1) dma_advance();
  };

  if constexpr (BIG) {
    // ---- 256 x 256 tile. LDS = FIVE slots of 32 KiB (all 160 KiB): the A tile (256 x 64) and the B tile (256 x 64) of a K step
    // are separate slots, and the DMA stream A(0) B(0) A(1) B(1) A(2) B(2) ... runs 2.5 steps ahead of the MFMAs: while step u is
    // computed (2 slots), step u+1 and the A tile of step u+2 are landed or in flight (3 slots = 96 KiB, as much as the 3-stage
    // ring of the smaller tiles keeps in flight, for twice the FLOPs per byte). A two-stage ring of 64 KiB stages (the first
    // version) could only refill after a whole stage was free: every step waited ~2 us for its own refill (no faster than 256x128).
    // One K step = four sub-phases of 16 MFMAs, (kk, jh) = (k half, column half):
    //   (0,0): A(kk0) x B(kk0, j 0-3)      while the reads of B(kk0, j 4-7) are in flight
    //   (0,1): A(kk0) x B(kk0, j 4-7)      while A(kk1) and B(kk1, j 0-3) are read
    //   (1,0): A(kk1) x B(kk1, j 0-3)      while B(kk1, j 4-7) is read   -> every read of this step's two slots has been issued
    //   lgkmcnt(0) + counted vmcnt (B(u+1) has landed; A(u+2) and an epilogue's stores may stay in flight) + s_barrier
    //   (1,1): A(kk1) x B(kk1, j 4-7)      while A'(kk0), B'(kk0, j 0-3) of step u+1 are read and the two freed slots are
    //          refilled: B(u+2), then A(u+3) (4 + 4 LDS-DMA instructions per wave, one per two MFMAs); then the epilogue if the
    //          item ends here.
    constexpr int SLOT = 32768;
    constexpr int RD_A = A_KM ? 8 : 4, RD_B = B_KM ? 8 : 4;  // LDS instructions of one A read / one 4-tile B read
    // Fragment addresses are RE-DERIVED from the lane id (a handful of VALU per K step) instead of living in VGPRs across the
    // loop: with 128 accumulators + 64 fragment registers the allocator spilled exactly these loop-invariant addresses, and every
    // scratch reload is followed by s_waitcnt vmcnt(0) — it drains the LDS-DMA ring (CDNA4 guide, 4-wave attention pitfalls:
    // "recompute per block"). `lane_` is made opaque once per K step so that the recomputation is not hoisted out of the loop.
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned kmA = 0, kmQ = 0, kmB = 0, kmS = 0;  // k-major fragment address parts
    auto km_parts = [&]() __attribute__((always_inline)) {
      const unsigned r = lane_ & 15, g = lane_ >> 4, q = r >> 2, pp = r & 3;
      const unsigned row = (8 * g + q) * 256 + ((pp >> 1) << 4) + ((pp & 1) << 3);
      if constexpr (A_KM) {
        kmA = lds0 + (wm >> 1) * 16384 + row + ((((unsigned)wm & 1) ^ (g & 1)) << 7);
        kmQ = q << 5;
      }
      if constexpr (B_KM) {
        kmB = lds0 + wn * 16384 + row;
        kmS = (q | ((g & 1) << 2)) << 5;
      }
    };
    km_parts();
    auto readA = [&](auto kk_c, int slot, bf16x8(&fa)[4]) __attribute__((always_inline)) {
      constexpr int kk = decltype(kk_c)::value;
      const unsigned so = (unsigned)(slot * SLOT);
      if constexpr (!A_KM) {
        const unsigned a = lds0 + so + kc_off(wm * 64 + (lane_ & 15), kk * 4 + (lane_ >> 4));
        fa[0] = lds_rd128<0>(a); fa[1] = lds_rd128<2048>(a); fa[2] = lds_rd128<4096>(a); fa[3] = lds_rd128<6144>(a);
      } else {
        // k-major image: the address of row tile i is constA + ((i ^ q) << 5) (q = the lane's k row inside its group of four:
        // the 32-byte XOR swizzle of km_off acts on exactly these bits), so two values stand for the four addresses
        const unsigned a = kmA + so;
        fa[0] = lds_rd_tr<kk * 8192>(a + ((0u << 5) ^ kmQ)); fa[1] = lds_rd_tr<kk * 8192>(a + ((1u << 5) ^ kmQ));
        fa[2] = lds_rd_tr<kk * 8192>(a + ((2u << 5) ^ kmQ)); fa[3] = lds_rd_tr<kk * 8192>(a + ((3u << 5) ^ kmQ));
      }
    };
    auto readB = [&](auto kk_c, auto jh_c, int slot, bf16x8(&fb)[4]) __attribute__((always_inline)) {
      constexpr int kk = decltype(kk_c)::value, jh = decltype(jh_c)::value;
      const unsigned so = (unsigned)(slot * SLOT);
      if constexpr (!B_KM) {
        const unsigned b = lds0 + so + kc_off(wn * (NJ * 16) + (lane_ & 15), kk * 4 + (lane_ >> 4));
        fb[0] = lds_rd128<(jh * 4 + 0) * 2048>(b); fb[1] = lds_rd128<(jh * 4 + 1) * 2048>(b);
        fb[2] = lds_rd128<(jh * 4 + 2) * 2048>(b); fb[3] = lds_rd128<(jh * 4 + 3) * 2048>(b);
      } else {
        const unsigned b = kmB + so;  // column tile j sits at constB + ((j ^ (q | (g4 & 1) << 2)) << 5)
        fb[0] = lds_rd_tr<kk * 8192>(b + (((unsigned)(jh * 4 + 0) << 5) ^ kmS)); fb[1] = lds_rd_tr<kk * 8192>(b + (((unsigned)(jh * 4 + 1) << 5) ^ kmS));
        fb[2] = lds_rd_tr<kk * 8192>(b + (((unsigned)(jh * 4 + 2) << 5) ^ kmS)); fb[3] = lds_rd_tr<kk * 8192>(b + (((unsigned)(jh * 4 + 3) << 5) ^ kmS));
      }
    };
    // one LDS-DMA instruction (1 KiB) of the loader cursor's A / B tile into `slot`
    auto dmaA = [&](int slot, auto pc_c) __attribute__((always_inline)) {
      constexpr int pc = decltype(pc_c)::value;
      unsigned char* sa = smem + slot * SLOT + wave * 4096 + pc * 1024;
      const int pi = wave * 4 + pc;
      const int vo = A_KM ? ((((pi & 15) >> 1) & 1) ? vA[1] : vA[0]) : vA[0];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lptr_t)sa, 16, vo, d_soffA + big_piece_soff(A_KM, llda, pi), 0, 0);
    };
    auto dmaB = [&](int slot, auto pc_c) __attribute__((always_inline)) {
      constexpr int pc = decltype(pc_c)::value;
      unsigned char* sb = smem + slot * SLOT + wave * 4096 + pc * 1024;
      const int pi = wave * 4 + pc;
      const int vo = B_KM ? ((((pi & 15) >> 1) & 1) ? vB[1] : vB[0]) : vB[0];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lptr_t)sb, 16, vo, d_soffB + big_piece_soff(B_KM, lldb, pi), 0, 0);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    int dslot = 0;  // slot of the next unit of the DMA stream
    auto next_slot = [&](int sl) __attribute__((always_inline)) -> int { return sl == 4 ? 0 : sl + 1; };
    auto unitA = [&]() __attribute__((always_inline)) {  // the cursor's A tile (the cursor has just entered this K step)
      dma_begin();
      big_offsets();
      dmaA(dslot, K0{}); dmaA(dslot, K1{}); dmaA(dslot, K2{}); dmaA(dslot, K3{});
      dslot = next_slot(dslot);
    };
    auto unitB = [&]() __attribute__((always_inline)) {  // the cursor's B tile, then the cursor moves to the next K step
      dmaB(dslot, K0{}); dmaB(dslot, K1{}); dmaB(dslot, K2{}); dmaB(dslot, K3{});
      dslot = next_slot(dslot);
      dma_advance();
    };
    auto mma16 = [&](const bf16x8(&fa)[4], const bf16x8(&fb)[4], auto jh_c) __attribute__((always_inline)) {
      constexpr int jh = decltype(jh_c)::value;
      G2_SETPRIO(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][jh * 4 + j] = mma1(fb[j], fa[i], acc[i][jh * 4 + j]);
      G2_SETPRIO(0);
    };
    // eight MFMAs of sub-phase (1,1) (row tiles I0, I0 + 1) with the four LDS-DMA instructions of one unit between them
    auto mma8_dma = [&](const bf16x8(&fa)[4], const bf16x8(&fb)[4], auto i0_c, auto isB_c, int slot) __attribute__((always_inline)) {
      constexpr int I0 = decltype(i0_c)::value;
      constexpr bool ISB = decltype(isB_c)::value != 0;
      __builtin_amdgcn_sched_barrier(0);
      G2_SETPRIO(1);
      auto one = [&](auto idx_c) __attribute__((always_inline)) {
        constexpr int idx = decltype(idx_c)::value;
        constexpr int i = I0 + idx / 4, j = idx % 4;
        acc[i][4 + j] = mma1(fb[j], fa[i], acc[i][4 + j]);
        if constexpr (idx % 2 == 1) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ISB) dmaB(slot, std::integral_constant<int, idx / 2>{});
          else dmaA(slot, std::integral_constant<int, idx / 2>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{});
      one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
      one(std::integral_constant<int, 4>{}); one(std::integral_constant<int, 5>{});
      one(std::integral_constant<int, 6>{}); one(std::integral_constant<int, 7>{});
      G2_SETPRIO(0);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto mma8 = [&](const bf16x8(&fa)[4], const bf16x8(&fb)[4], auto i0_c) __attribute__((always_inline)) {
      constexpr int I0 = decltype(i0_c)::value;
      G2_SETPRIO(1);
#pragma unroll
      for (int i = I0; i < I0 + 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][4 + j] = mma1(fb[j], fa[i], acc[i][4 + j]);
      G2_SETPRIO(0);
    };
    // prologue: A(0) B(0) A(1) B(1) A(2)
    unitA(); unitB();
    if (total > 1) { unitA(); unitB(); }
    if (total > 2) unitA();
    if (total > 2) wait_vm<12>();
    else if (total > 1) wait_vm<8>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    bf16x8 a0[4], a1[4], b0[4], b1[4];
    int cA = 0, cB = 1;  // slots of the step being computed
    readA(K0{}, cA, a0);
    readB(K0{}, K0{}, cB, b0);
    int eu1 = 0;
    for (int u = 0; u < total; ++u) {
      asm volatile("" : "+v"(lane_));
      km_parts();
      // (0,0)
      readB(K0{}, K1{}, cB, b1);
      wait_lgkm<RD_B>();
      __builtin_amdgcn_sched_barrier(0);
      mma16(a0, b0, K0{});
      __builtin_amdgcn_sched_barrier(0);
      // (0,1)
      readA(K1{}, cA, a1);
      readB(K1{}, K0{}, cB, b0);
      wait_lgkm<(RD_A + RD_B < 15 ? RD_A + RD_B : 14)>();  // (lgkmcnt is 4 bits: a stricter wait for the TN variant)
      __builtin_amdgcn_sched_barrier(0);
      mma16(a0, b1, K1{});
      __builtin_amdgcn_sched_barrier(0);
      // (1,0)
      readB(K1{}, K1{}, cB, b1);
      wait_lgkm<RD_B>();
      __builtin_amdgcn_sched_barrier(0);
      mma16(a1, b0, K0{});
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm<0>();  // every read of this step's slots has completed: they may be refilled
      __builtin_amdgcn_sched_barrier(0);
      if (u + 1 < total) {
        // B(u+1) was issued one step ago. Newer operations: the 4 pieces of A(u+2) and the stores of the epilogue that followed
        switch (eu1 * 2 + (u + 2 < total ? 1 : 0)) {
          case 0: wait_vm<0>(); break;
          case 1: wait_vm<4>(); break;
          case 2: wait_vm<NSU>(); break;
          case 3: wait_vm<NSU + 4>(); break;
          case 4: wait_vm<2 * NSU>(); break;
          case 5: wait_vm<2 * NSU + 4>(); break;
          case 6: wait_vm<3 * NSU>(); break;
          case 7: wait_vm<3 * NSU + 4>(); break;
          case 8: wait_vm<4 * NSU>(); break;
          default: wait_vm<4 * NSU + 4>(); break;
        }
        __builtin_amdgcn_s_barrier();
      }
      const int nA = cA >= 3 ? cA - 3 : cA + 2, nB = cB >= 3 ? cB - 3 : cB + 2;
      // On the last K step of an item the read-ahead of step u+1's first fragments follows the epilogue (32 VGPRs the
      // epilogue can use: with them live the loop's address registers were spilled, and every scratch reload waits vmcnt(0))
      const bool item_ends = c_kt + 1 == C.nk;
      if (u + 1 < total && !item_ends) {
        readA(K0{}, nA, a0);
        readB(K0{}, K0{}, nB, b0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // (1,1), with the refill of the two freed slots: B(u+2) then A(u+3)
      if (u + 2 < total) {
        big_offsets();  // (re-derived here too: the per-lane source offsets are not carried across the K loop)
        mma8_dma(a1, b1, K0{}, K1{}, dslot);
        dslot = next_slot(dslot);
        dma_advance();
        if (u + 3 < total) {
          dma_begin();
          big_offsets();
          mma8_dma(a1, b1, K2{}, K0{}, dslot);
          dslot = next_slot(dslot);
        } else {
          mma8(a1, b1, K2{});
        }
      } else {
        mma16(a1, b1, K1{});
      }
      int e = 0;
      if (++c_kt == C.nk) {
        e = epilogue();
        __builtin_amdgcn_sched_barrier(0);
        if (u + 1 < total) {
          readA(K0{}, nA, a0);
          readB(K0{}, K0{}, nB, b0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        c_kt = 0;
        c_item += G;
        if (c_item < s.items) {
          C = decode(c_item);
          compute_setup();
        }
      }
      eu1 = e;
      cA = nA;
      cB = nB;
    }
  } else {
  // prologue: up to three steps of DMA in flight, then wait for step 0 and read its first-half fragments
  issue(0);
  if (total > 1) issue(1);
  if (total > 2) issue(2);
  if (total > 2) wait_vm<2 * NL>();
  else if (total > 1) wait_vm<NL>();
  else wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  bf16x8 a0[4], b0[NJ], a1[4], b1[NJ];
  read_half(std::integral_constant<int, 0>{}, 0, a0, b0);
  if (G2_DBG & 2) read_half(std::integral_constant<int, 1>{}, 0, a1, b1);

  int eu1 = 0;    // store units of the counted epilogue that ended the previous step
  int st = 0;     // u % 3
  for (int u = 0; u < total; ++u) {
    if (!(G2_DBG & 2)) read_half(std::integral_constant<int, 1>{}, st, a1, b1);
    wait_lgkm<(RD_HALF < 14 ? RD_HALF : 14)>();  // first-half fragments (issued half a step ago) are in; 15 = "no wait"
    __builtin_amdgcn_sched_barrier(0);
    // second part of the refill started in the previous step (the DMA of step u+2, into the stage of step u-1)
    if (u >= 1 && u + 2 < total && !(G2_DBG & 1)) mma_half_dma(a0, b0, st == 0 ? 2 : st - 1, std::integral_constant<int, 1>{});
    else mma_half(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    const int nst = st == 2 ? 0 : st + 1;
    wait_lgkm<0>();  // second-half fragments are in (and every read of this stage is done: it may be refilled)
    __builtin_amdgcn_sched_barrier(0);
    if (u + 1 < total) {
      {  // step u+1's DMA: allow the newer step's loads (both parts are issued by now) + the stores of the counted
         // epilogue of the previous step (the only one issued after the last part of step u+1's DMA)
        const int code = (u + 2 < total ? 3 : 0) + eu1;
        switch (code) {
          case 0: wait_vm<0>(); break;
          case 1: wait_vm<NSU>(); break;
          case 2: wait_vm<2 * NSU>(); break;
          case 3: wait_vm<NL>(); break;
          case 4: wait_vm<NL + NSU>(); break;
          default: wait_vm<NL + 2 * NSU>(); break;
        }
      }
      if (!(G2_DBG & 4)) __builtin_amdgcn_s_barrier();
      if (!(G2_DBG & 2)) read_half(std::integral_constant<int, 0>{}, nst, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (u + 3 < total && !(G2_DBG & 1)) mma_half_dma(a1, b1, st, std::integral_constant<int, 0>{});
    else mma_half(a1, b1);
    int e = 0;
    if (++c_kt == C.nk) {
      if (!(G2_DBG & 8)) e = epilogue();
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (GCOLSUM) {
#pragma unroll
        for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      c_kt = 0;
      c_item += G;
      if (c_item < s.items) {
        C = decode(c_item);
        compute_setup();
      }
    }
    eu1 = e;
    st = nst;
  }
  }
  stamp_end(s.stamp);
#endif
}

// ---- host side -------------------------------------------------------------------------------------------------

#ifndef G2_PART  // (host-side planning lives in the main translation unit only: see the end of this file)
static int g2_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
    // MMSA_G2_CUS=<n>: cap the persistent grid, leaving CUs to kernels of other streams (a workgroup of this kernel
    // fills its CU's register file and LDS, so e.g. RCCL's blocks cannot co-reside with it: data-parallel runs that
    // overlap the gradient all-reduce with the backward can reserve the collective's CUs instead of having a second
    // round of tiles wait for them)
    if (const char* c = getenv("MMSA_G2_CUS")) {
      const int cap = atoi(c);
      if (cap >= 8 && cap < n) n = cap;
    }
  }
  return n;
}

// byte extent of the A / B operand views (what the buffer descriptors cover)
static void g2_extents(const GemmParams& p, long* ea, long* eb) {
  *ea = p.a_kmajor ? ((long)(p.K - 1) * p.lda + p.M) * 2 : ((long)(p.M - 1) * p.lda + p.K) * 2;
  *eb = p.b_kmajor ? ((long)(p.K - 1) * p.ldb + p.N) * 2 : ((long)(p.N - 1) * p.ldb + p.K) * 2;
  const long ghw = (long)p.g.GH * p.g.GW;
  if (p.gather == 1) *ea = (long)(p.M / (ghw > 0 ? ghw : 1)) * p.g.SH * p.g.SW * p.g.src_pix_stride * 2;  // source activation
  if (p.gather == 2) *eb = (long)(p.K / (ghw > 0 ? ghw : 1)) * p.g.SH * p.g.SW * p.g.src_pix_stride * 2;
  if (p.gather == 1 && p.b_kmajor)  // weight [cout][taps][cin] addressed through (k % cper) * ldb + tap * b_tap_stride
    *eb = ((long)(p.g.cper - 1) * p.ldb + b_tap_offset(p, p.g.KH - 1, p.g.KW - 1) + p.N) * 2;
}

bool gemm2_eligible(const GemmParams& p) {
  if (p.gather < 0 || p.gather > 2) return false;
  if (p.scale_a && (p.gather || p.a_kmajor || p.b_kmajor || !p.scale_b)) return false;  // fp8 operands: NT only
  if (p.K % G2_BK) return false;
  if (p.N % 4) return false;
  if (p.a_kmajor && (p.M % 8)) return false;
  if (p.b_kmajor && (p.N % 8)) return false;
  if ((p.lda % 8) || (p.ldb % 8)) return false;
  if (p.gather == 1) {
    if (p.a_kmajor || (p.g.cper % G2_BK) || (p.g.src_pix_stride % 8)) return false;
    if (p.M % ((long)p.g.GH * p.g.GW)) return false;
    if (p.g.div != 1 && p.g.div != 2) return false;
  }
  if (p.gather == 2) {
    if (!(p.a_kmajor && p.b_kmajor) || (p.g.cper % 8) || (p.g.src_pix_stride % 8)) return false;
    if (p.K % ((long)p.g.GH * p.g.GW)) return false;
    if (p.g.div != 1 && p.g.div != 2) return false;
  }
  long ea, eb;
  g2_extents(p, &ea, &eb);
  // tile origins may start up to one tile past the last row: keep every scalar + vector offset a positive int32
  const long slack = 2L * 256 * (p.lda > p.ldb ? p.lda : p.ldb) + 65536;
  if (ea + slack >= 0x7FFFFFF0L || eb + slack >= 0x7FFFFFF0L) return false;
  long ec = ((long)(p.M - 1) * p.ldc + p.N) * (p.out_f32 ? 4 : 2);
  if (p.c_gw > 0) {  // mapped output: plain bf16 store only
    if (p.out_f32 || p.bias || p.C2 || p.mul || p.add || p.act != MMSA_ACT_NONE || p.c_gh <= 0) return false;
    if (p.M % ((long)p.c_gh * p.c_gw)) return false;
    ec = ((long)(p.M / ((long)p.c_gh * p.c_gw)) * p.c_imgpitch) * 2;
  }
  if (ec >= 0x7FFFFFF0L) return false;
  return true;
}

thread_local int g2_last_plan[3] = {0, 0, 0};  // profiling only (gemm_mfma.hip): tile shape and K split of the latest launch

#endif  // !G2_PART
static inline int g2_epi_class(const GemmParams& p) {
  return (p.mul || p.add) ? 2 : (p.bias || p.C2 || p.act != MMSA_ACT_NONE || (p.out_f32 && p.accumulate)) ? 1 : 0;
}

// Which epilogue instantiation a non-plain launch takes (gemm2_kernel's EPI): 2-5 = the specialised ones, which exist for the
// plain (no gather) NT / NN problems; 1 = the general one. (0, the plain / slab store, is decided by the launcher.)
static inline int g2_epi_kind(const GemmParams& p) {
  static const bool off = mmsa_disabled("epi_spec");  // A/B hook
  if (off || p.a_kmajor || p.gather) return 1;
  if (p.out_f32 || p.act_after_add || (p.C2 && !p.c2_gelu_grad)) return 1;
  if (!p.mul && !p.add) {
    if (p.b_kmajor) return 1;
    if (!p.C2 && p.act == MMSA_ACT_NONE) return 2;
    if (p.C2 && p.c2_gelu_grad && p.act == MMSA_ACT_GELU) return 3;
    return 1;
  }
  if (p.C2 || p.act != MMSA_ACT_NONE) return 1;
  if (p.add && !p.mul) return 4;
  if (p.mul && p.mul_is_factor && !p.add && p.b_kmajor) return 5;
  return 1;
}

#ifndef G2_PART
struct G2Plan { int wm, nj, split; };  // wave rows (4: 256-row tile, 2: 128-row tile), column tiles per wave, K split
static inline int g2_bm(const G2Plan& pl) { return pl.wm * 64; }
static inline int g2_bn(const G2Plan& pl) { return pl.nj * 16 * (8 / pl.wm); }

// Tile shape and K split. Cost model in microseconds, calibrated on MI355X (profiles/): a K step of the 256-row tile
// costs about the same for every width (the 32 KiB A tile, the barrier and the LDS traffic dominate; MFMA-only time is
// 0.19 us per 32 columns): t_step = 1.0 + 0.06 nj; per item 0.3 + 0.2 nj (pipeline bubble + epilogue stores);
// the 128-row tile moves 2/3 of the bytes and half of the MFMAs per step: t_step = 0.72 + 0.05 per 32 columns (stand-
// alone it measures 0.42 + 0.07, but planning with that costs the whole step 0.22 ms: A/B on one box, 20.27 vs 20.50);
// a split adds the slab round trip at ~4 TB/s and the reducer launch. So: the biggest tile that keeps whole rounds of
// `cus` workgroups busy, and as many K slices as it takes to fill the chip when there are few tiles (weight
// gradients: 1..72 tiles with K = 8192..802816).
// cost-model constants (microseconds); MMSA_G2_MODEL="a4,b4,a2,b2,c_split,d_split" overrides them for calibration runs
struct G2Model { double a4, b4, a2, b2, c_split, d_split; };
static const G2Model& g2_model() {
  static const G2Model m = [] {
    G2Model d{1.0, 0.06, 0.72, 0.05, 3.0, 2.0};
    if (const char* v = MMSA_EXP_ENV("MMSA_G2_MODEL")) {
      G2Model t = d;
      if (sscanf(v, "%lf,%lf,%lf,%lf,%lf,%lf", &t.a4, &t.b4, &t.a2, &t.b2, &t.c_split, &t.d_split) == 6) d = t;
    }
    return d;
  }();
  return m;
}

static G2Plan g2_plan_search(const GemmParams& p, int cus, size_t ws_bytes_avail, int force_wm = 0, int force_nj = 0) {
  const int nsteps = p.K / G2_BK;
  G2Plan best{4, 4, 1};
  double best_cost = 1e300;
  static const int cfgs[6][2] = {{4, 4}, {4, 3}, {4, 2}, {2, 2}, {2, 1}, {4, 8}};
  // The 256 x 256 tile is opt-in (MMSA_G2_BIG=1, or forced per launch with MMSA_G2_NJ=4:8): measured on MI355X it ties the
  // 256 x 128 tile on large NT / NN problems (both ~45 % of the MFMA-only time of the same loop) and loses on everything with
  // fewer than ~256 tiles; see DESIGN.md section 3 for the ablations.
  static const bool no_big = [] { const char* v = MMSA_EXP_ENV("MMSA_G2_BIG"); return !(v && atoi(v) != 0); }();
  // epilogue class: 0 plain store, 1 bias / activation / side output (registers only), 2 reads gelu' / residual operands
  const int epi = g2_epi_class(p);
  const G2Model& mdl = g2_model();
  for (int ci = 0; ci < 6; ++ci) {
    const G2Plan shape{cfgs[ci][0], cfgs[ci][1], 1};
    if (force_wm && (shape.wm != force_wm || shape.nj != force_nj)) continue;
    // the 256 x 256 tile: plain problems only (no implicit-GEMM gather, no fp8 operands, no mapped output rows)
    if (shape.nj == 8 && (p.gather || p.scale_a || p.c_gw > 0 || p.colstat || p.mul || p.add || p.a_kmajor ||
                          (p.out_f32 && p.accumulate) || (no_big && !force_wm)))
      continue;
    const int bm = g2_bm(shape), bn = g2_bn(shape);
    const int ntm = cdiv(p.M, bm), ntn = cdiv(p.N, bn);
    const long tiles = (long)ntm * ntn;
    int smax = 1;
    if (ws_bytes_avail > 0 && p.ws) {
      smax = nsteps / 2;
      if (smax > 256) smax = 256;
      if (smax < 1) smax = 1;
      const long cap = (long)(ws_bytes_avail / ((size_t)p.M * p.N * sizeof(float)));
      if (cap < smax) smax = (int)(cap > 1 ? cap : 1);
    }
    int cand[16];
    int nc = 0;
    cand[nc++] = 1;
    for (int r = 1; r <= 6 && smax > 1; ++r) {
      const long sp = (long)r * cus / tiles;
      for (long d = sp; d <= sp + 1; ++d)
        if (d > 1 && d <= smax && nc < 16) cand[nc++] = (int)d;
    }
    const int w32 = bn / 32;  // tile width in 32-column units
    const double t_step = shape.wm == 4 ? mdl.a4 + mdl.b4 * w32 : mdl.a2 + mdl.b2 * w32;
    const double t_item = shape.wm == 4 ? 0.3 + 0.2 * w32 : 0.3 + 0.1 * w32;
    // the register-only epilogue costs ~1 us per item, the one with side operands one memory round trip more; on the
    // 256x128 tile both run out of registers (accumulators + side operands + the loop's prefetched fragments: the
    // compiler spills ~60 VGPRs there), measured +3 / +8 us per item (tools/microbench/bench_small.py, bench_epi2.py)
    // (r3, tools/microbench/bench_epi2.py on 8192x3072x768: bias + GELU + side output 94.8 us on the 256x128 tile against 82.0 on
    //  256x96, bias alone 65.7 / 57.9 — the 128-wide general epilogue is the one that spills; its price went up accordingly)
    // (r3: the specialised epilogues — g2_epi_kind 2, 4, 5 — do not spill on the 256x128 tile; in situ the x gelu' data gradient
    //  8192x3072x768 runs 58.7 us there against 64.3 on 256x96, the residual-add 1x1 data gradients 3-14 % faster)
    static const double epi_cost_gen[3][6] = {{0, 0, 0, 0, 0, 0}, {6.0, 1.0, 0.9, 0.65, 0.5, 4.0}, {12.0, 2.0, 1.2, 1.0, 0.8, 10.0}};
    static const double epi_cost_spec[2][6] = {{1.0, 1.0, 0.9, 0.65, 0.5, 4.0}, {2.0, 2.0, 1.2, 1.0, 0.8, 10.0}};
    const int kind = epi ? g2_epi_kind(p) : 0;
    const double* epi_cost_row = (kind == 2) ? epi_cost_spec[0] : (kind == 4 || kind == 5) ? epi_cost_spec[1] : epi_cost_gen[epi];
    static const bool no_epi = [] { const char* v = MMSA_EXP_ENV("MMSA_G2_NOEPI"); return v && atoi(v) != 0; }();  // A/B hook
    const double waste = (double)ntn * bn / p.N * ((double)ntm * bm / p.M);  // padding: only as a tie breaker
    for (int c = 0; c < nc; ++c) {
      const int per = cdiv(nsteps, cand[c]);
      const int split = cdiv(nsteps, per);  // no empty slices: e.g. 3136 steps / 256 -> 13 per slice -> 242 slices
      const long items = tiles * split;
      const long rounds = (items + cus - 1) / cus;
      // (with a K split the tiles store raw slabs and the reducer applies the epilogue)
      double cost = (double)rounds * (per * t_step + t_item + (split == 1 && !no_epi ? epi_cost_row[ci] : 0.0)) + 1e-3 * waste;
      if (split > 1) cost += mdl.c_split + mdl.d_split * split * (double)p.M * p.N * 4.0 / 4e6;
      if (cost < best_cost - 1e-9) { best_cost = cost; best = G2Plan{shape.wm, shape.nj, split}; }
    }
  }
  return best;
}

// plans are pure functions of the shape: memoize (one caller thread per process — include/mmsa.h)
static G2Plan g2_plan(const GemmParams& p, int cus, size_t ws_bytes_avail) {
  struct Key { int M, N, K, epi, big_ok; size_t ws; G2Plan plan; };  // (epi: class * 8 + kind)
  static std::vector<Key> cache;
  static std::mutex mu;  // the two encoders may be enqueued from two host threads (engine.py: EngineModule.use_host_worker)
  std::lock_guard<std::mutex> lock(mu);
  const int epi = g2_epi_class(p) * 8 + (g2_epi_class(p) ? g2_epi_kind(p) : 0);
  const int big_ok = !(p.gather || p.scale_a || p.c_gw > 0 || p.colstat || p.mul || p.add || p.a_kmajor);  // which tile shapes exist
  for (const Key& k : cache)
    if (k.M == p.M && k.N == p.N && k.K == p.K && k.epi == epi && k.big_ok == big_ok && k.ws == ws_bytes_avail) return k.plan;
  const G2Plan plan = g2_plan_search(p, cus, ws_bytes_avail);
  if (cache.size() < 4096) cache.push_back(Key{p.M, p.N, p.K, epi, big_ok, ws_bytes_avail, plan});
  return plan;
}

#endif  // !G2_PART
template <int WM, int NJ, bool A_KM, bool B_KM, int GATHER, bool FP8, int EPI>
static int g2_launch_e(const GemmParams& p, const G2Sched& s, int grid, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm2_kernel<WM, NJ, A_KM, B_KM, GATHER, FP8, EPI>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, NJ == 8 ? 163840 : G2_LDS);
    attr_set = true;
  }
  // (the 256 x 256 tile takes all 160 KiB of LDS: five 32 KiB slots)
  hipLaunchKernelGGL((gemm2_kernel<WM, NJ, A_KM, B_KM, GATHER, FP8, EPI>), dim3(grid), dim3(512), NJ == 8 ? 163840 : G2_LDS, st, p, s);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
template <int WM, int NJ, bool A_KM, bool B_KM, int GATHER, bool FP8 = false>
static int g2_launch_t(const GemmParams& p, const G2Sched& s, int grid, hipStream_t st) {
  if (p.c_gw > 0 && p.c_rmw) {  // mapped read-modify-write store: the strided 1x1 data gradient (NN) on top of the main path's
    if constexpr (!A_KM && B_KM && GATHER == 0 && !FP8 && NJ != 8) {
      if (s.split_k == 1) return g2_launch_e<WM, NJ, A_KM, B_KM, GATHER, FP8, 6>(p, s, grid, st);
    }
    return MMSA_ERR_UNSUPPORTED;
  }
  // plain-store and slab-store launches: the instantiation without the general epilogue (see gemm2_kernel)
  if (s.fast || s.split_k > 1) return g2_launch_e<WM, NJ, A_KM, B_KM, GATHER, FP8, 0>(p, s, grid, st);
  // the specialised epilogues exist for the plain (no gather, bf16 operands) NT and NN problems: the Linears of the text encoder
  // and their data gradients, the 1x1 convolutions' data gradients with the skip connection's gradient added
  if constexpr (!A_KM && GATHER == 0 && NJ != 8) {  // (fp8 operands: NT only, so kinds 2-4)
    const int kind = g2_epi_kind(p);
    if constexpr (!B_KM) {
      if (kind == 2) return g2_launch_e<WM, NJ, A_KM, B_KM, GATHER, FP8, 2>(p, s, grid, st);
      if (kind == 3) return g2_launch_e<WM, NJ, A_KM, B_KM, GATHER, FP8, 3>(p, s, grid, st);
    } else {
      if (kind == 5) return g2_launch_e<WM, NJ, A_KM, B_KM, GATHER, FP8, 5>(p, s, grid, st);
    }
    if (kind == 4) return g2_launch_e<WM, NJ, A_KM, B_KM, GATHER, FP8, 4>(p, s, grid, st);
  }
  return g2_launch_e<WM, NJ, A_KM, B_KM, GATHER, FP8, 1>(p, s, grid, st);
}

template <int WM, int NJ>
static int g2_launch_nj(const GemmParams& p, const G2Sched& s, int grid, hipStream_t st) {
  if (p.gather == 1) {
    if (!p.b_kmajor) return g2_launch_t<WM, NJ, false, false, 1>(p, s, grid, st);
    return g2_launch_t<WM, NJ, false, true, 1>(p, s, grid, st);
  }
  if (p.gather == 2) return g2_launch_t<WM, NJ, true, true, 2>(p, s, grid, st);
  if (!p.a_kmajor && !p.b_kmajor && p.scale_a) return g2_launch_t<WM, NJ, false, false, 0, true>(p, s, grid, st);
  if (!p.a_kmajor && !p.b_kmajor) return g2_launch_t<WM, NJ, false, false, 0>(p, s, grid, st);
  if (!p.a_kmajor && p.b_kmajor) return g2_launch_t<WM, NJ, false, true, 0>(p, s, grid, st);
  if (p.a_kmajor && p.b_kmajor) return g2_launch_t<WM, NJ, true, true, 0>(p, s, grid, st);
  return g2_launch_t<WM, NJ, true, false, 0>(p, s, grid, st);
}

// ---- one translation unit per tile shape -------------------------------------------------------------------------------------------
// This file is compiled six times (csrc/Makefile): once plain — the host side: planning, scheduling, the grouped launches, with the
// five g2_launch_nj<WM, NJ> families as EXTERNAL functions — and once per family with -DG2_PART=<WM><NJ>, which compiles nothing but
// that family's kernel instantiations and its launcher. As one translation unit the ~100 instantiations took 5-6 minutes on one
// core while the other seven idled: the whole library build was this file.
#ifdef G2_PART  // (G2_PART = 10 WM + NJ, G2_PART_NAME = g2_launch_nj_<WM><NJ>: both from the Makefile)
int G2_PART_NAME(const GemmParams& p, const G2Sched& s, int grid, hipStream_t st) {
  return g2_launch_nj<G2_PART / 10, G2_PART % 10>(p, s, grid, st);
}
#else
int g2_launch_nj_44(const GemmParams& p, const G2Sched& s, int grid, hipStream_t st);
int g2_launch_nj_43(const GemmParams& p, const G2Sched& s, int grid, hipStream_t st);
int g2_launch_nj_42(const GemmParams& p, const G2Sched& s, int grid, hipStream_t st);
int g2_launch_nj_22(const GemmParams& p, const G2Sched& s, int grid, hipStream_t st);
int g2_launch_nj_21(const GemmParams& p, const G2Sched& s, int grid, hipStream_t st);

// p.split_k on entry: 1 = no split wanted; > 1 = upper bound chosen by the caller (needs p.ws with room for it)
int gemm2_launch(const GemmParams& pin, size_t ws_bytes_avail, hipStream_t st) {
  GemmParams p = pin;
  const int cus = g2_num_cus();
  if (p.c_gw > 0) { ws_bytes_avail = 0; p.split_k = 1; }  // mapped output rows: no K split
  if (p.split_k < 1) p.split_k = 1;
  if (p.split_k > 1 && !p.ws) return MMSA_ERR_ARG;
  G2Plan plan;
  {
    int fwm = 0, fnj = 0;
    if (const char* f = getenv("MMSA_G2_NJ")) {  // test hook: "nj" (256-row tile) or "wm:nj"; the K split is still planned
      int a = 0, b = 0;
      const int n = sscanf(f, "%d:%d", &a, &b);
      if (n == 2 && (a == 4 || a == 2) && b >= 1 && (b <= (a == 4 ? 4 : 2) || (a == 4 && b == 8)) && !(a == 4 && b < 2)) { fwm = a; fnj = b; }
      else if (n == 1 && ((a >= 2 && a <= 4) || a == 8)) { fwm = 4; fnj = a; }
#ifndef MMSA_EXPERIMENTS
      if (fnj == 8) { fwm = 0; fnj = 0; }  // (the 256 x 256 tile exists in experiment builds only)
#endif
      if (fnj == 8 && (p.gather || p.scale_a || p.c_gw > 0 || p.colstat || p.mul || p.add || p.a_kmajor || (p.out_f32 && p.accumulate))) { fwm = 0; fnj = 0; }  // the forced shape does not exist for this problem
    }
    plan = fwm ? g2_plan_search(p, cus, ws_bytes_avail, fwm, fnj) : g2_plan(p, cus, ws_bytes_avail);
  }
  if (const char* f = MMSA_EXP_ENV("MMSA_G2_SPLIT")) {  // test hook: force the K split (needs a workspace that holds it)
    const int sp = atoi(f);
    if (sp >= 1 && sp <= p.K / G2_BK && (sp == 1 || (p.ws && (size_t)sp * p.M * p.N * 4 <= ws_bytes_avail))) plan.split = sp;
  }
  const long slab_bytes = (long)plan.split * p.M * p.N * 4;
  if (plan.split > 1 && slab_bytes >= 0x7FFFFFF0L) plan.split = 1;
  G2Sched s;
  const int bm = g2_bm(plan), bn = g2_bn(plan);
  s.ntm = cdiv(p.M, bm); s.ntn = cdiv(p.N, bn); s.ntiles = s.ntm * s.ntn;
  s.nsteps = p.K / G2_BK;
  s.split_k = plan.split;
  s.per = cdiv(s.nsteps, s.split_k);
  s.split_k = cdiv(s.nsteps, s.per);
  s.items = s.ntiles * s.split_k;
  s.fd_ntiles = make_fastdiv((uint32_t)s.ntiles);
  s.fd_ntn = make_fastdiv((uint32_t)s.ntn);
  s.rb = 0; s.cb = 0;
  s.use_order = 0;
  s.fd_cb = make_fastdiv(1); s.fd_sbc = make_fastdiv(1);
  {
    // On by default since round 3 (MMSA_DISABLE=g2_2d turns it off): in round 2 it changed neither the stand-alone GEMM times nor the
    // step (the CUs of an XCD walk K in lock-step, so whoever shares a panel waits for the same fill either way) and only lowered
    // the L2-miss traffic the Infinity Cache absorbs; with the leaner round-3 kernels the same A/B reads 17.79-17.87 against
    // 18.02-18.11 ms/step (GEMM time 10.15 against 10.36 ms).
    static const bool no2d = mmsa_disabled("g2_2d");
    // 2-D blocking when it lowers (2 rb + cb): A panel = 32 KiB, B panel = 16 KiB per K step and XCD
    int best_cb = 0;
    double best = 2.0 * 32.0 / s.ntn + s.ntn;  // row-major: a run of 32 tiles spans 32/ntn rows and all ntn columns
    if (s.ntn >= 32) best = 2.0 + 32.0;
    for (int cb = 2; cb <= 16; cb <<= 1) {
      const int rb = 32 / cb;
      if (s.ntn % cb || s.ntm % rb) continue;
      const double c = 2.0 * rb + cb;
      if (c < best - 1e-9) { best = c; best_cb = cb; }
    }
    if (best_cb && !no2d && cus == 256) {
      s.cb = best_cb; s.rb = 32 / best_cb;
      s.fd_cb = make_fastdiv((uint32_t)s.cb);
      s.fd_sbc = make_fastdiv((uint32_t)(s.ntn / s.cb));
    }
  }
  s.fast = (!p.bias && !p.C2 && p.act == MMSA_ACT_NONE && !p.mul && !p.add && !(p.out_f32 && p.accumulate)) ? 1 : 0;
  p.split_k = s.split_k;
  s.dbg = 0;
  s.ngroups = 0;
  s.stamp = p.stamp;
  s.colstat = nullptr;
  s.colstat_bytes = 0;
  if (p.colstat_rows) *p.colstat_rows = 0;
  if (p.colstat && p.colstat_rows && s.fast && s.split_k == 1 && !p.out_f32 && p.c_gw == 0 && !p.scale_a) {
    const long rows = cdiv(p.M, 64), need = rows * 2 * p.N;
    // (a 128- or 256-row tile may start a slice past cdiv(M, 64): those rows are zero and their stores are dropped by the range check)
    if (need <= p.colstat_cap && need * 4 < 0x7FFFFFF0L) {
      s.colstat = p.colstat;
      s.colstat_bytes = (unsigned)(need * 4);
      *p.colstat_rows = (int)rows;
    }
  }
  if (const char* d = MMSA_EXP_ENV("MMSA_G2_DBG")) s.dbg = atoi(d);
  s.c_bytes = s.split_k > 1 ? (unsigned)slab_bytes : (unsigned)(((long)(p.M - 1) * p.ldc + p.N) * (p.out_f32 ? 4 : 2));
  if (p.c_gw > 0) {
    if (s.split_k > 1) return MMSA_ERR_UNSUPPORTED;
    s.c_bytes = (unsigned)(((long)(p.M / ((long)p.c_gh * p.c_gw)) * p.c_imgpitch) * 2);
  }
  {
    long ea, eb;
    g2_extents(p, &ea, &eb);
    p.a_bytes = (unsigned)ea; p.b_bytes = (unsigned)eb;
    // timing-only diagnostic (results are wrong): zero-record descriptors make the range check drop every staging
    // load while the instruction stream, waits and barriers stay — prices the memory side of the loop.
    if (const char* nl = MMSA_EXP_ENV("MMSA_GEMM_DBG_NOLOAD"))
      if (atoi(nl)) { p.a_bytes = 0; p.b_bytes = 0; }
  }
  const int grid = (int)(s.items < cus ? s.items : cus);
  g2_last_plan[0] = plan.wm; g2_last_plan[1] = plan.nj; g2_last_plan[2] = s.split_k;
  int rc;
  if (plan.wm == 4) {
    if (plan.nj == 8) {
#ifdef MMSA_EXPERIMENTS  // the 256 x 256 tile (built in round 3, never faster: DESIGN.md section 3) is instantiated in experiment builds only
      if (!p.a_kmajor && !p.b_kmajor) rc = g2_launch_t<4, 8, false, false, 0>(p, s, grid, st);
      else if (!p.a_kmajor && p.b_kmajor) rc = g2_launch_t<4, 8, false, true, 0>(p, s, grid, st);
      else
#endif
      rc = MMSA_ERR_UNSUPPORTED;
    }
    else if (plan.nj == 4) rc = g2_launch_nj_44(p, s, grid, st);
    else if (plan.nj == 3) rc = g2_launch_nj_43(p, s, grid, st);
    else rc = g2_launch_nj_42(p, s, grid, st);
  } else {
    if (plan.nj == 2) rc = g2_launch_nj_22(p, s, grid, st);
    else rc = g2_launch_nj_21(p, s, grid, st);
  }
  if (rc) return rc;
  if (s.split_k > 1) {
    launch_splitk_reduce<bf16>(p, st);
    MMSA_CHECK_LAUNCH();
  }
  return MMSA_OK;
}

// Second pass of a grouped launch with a K split: one launch sums the slabs of every problem (slab order = fixed summation
// order -> bitwise reproducible) into its fp32 output (overwrite or +=).
struct G2GroupReduce {
  int n, split, accumulate;
  long total4;
  unsigned long long* stamp;
  struct { const float* slab; float* C; long mn, ldc, begin4; int N; } g[G2_MAX_GROUPS];
};
__global__ __launch_bounds__(256) void gemm2_group_reduce_kernel(G2GroupReduce a) {
  const bool stamped = (blockIdx.x & 15) == 0 || blockIdx.x == gridDim.x - 1;
  if (stamped) stamp_begin(a.stamp);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.total4; i += (long)gridDim.x * 256) {
    int g = 0;
#pragma unroll
    for (int q = 1; q < G2_MAX_GROUPS; ++q)
      if (q < a.n && i >= a.g[q].begin4) g = q;
    const long e = (i - a.g[g].begin4) * 4;
    const float* src = a.g[g].slab + e;
    f32x4 v = *(const f32x4*)src;
    for (int sp = 1; sp < a.split; ++sp) v += *(const f32x4*)(src + (long)sp * a.g[g].mn);
    const long m = e / a.g[g].N, nn = e - m * a.g[g].N;
    float* dst = a.g[g].C + m * a.g[g].ldc + nn;
    if (a.accumulate) v += *(const f32x4*)dst;
    *(f32x4*)dst = v;
  }
  if (stamped) stamp_end(a.stamp);
}

// Walk order of a grouped launch's tiles (G2Sched::order). An XCD works on a contiguous run of ~tiles/8 positions at a time and
// its L2 holds the panels of that run: a run of R tiles that spans r rows and c columns of one problem fetches r A panels (256 rows
// of dY: 32 KiB per K step) and c B panels (128 columns of X: 16 KiB), so the order keeps runs compact — each problem is walked in
// column blocks of ~sqrt(2 R) tiles, row by row inside a block (row-major for the narrow problems; the 24-column FFN-down weight
// gradient of a BERT layer in three blocks of 8: its runs then touch 3 + 10 panels instead of 2 + 24) — and a problem that reads
// the SAME A operand as an earlier one (a bias gradient: dY^T times a ones column) is dealt row by row into that problem's first
// column block, right behind the weight-gradient tiles of the same dY panel, instead of re-reading all of dY on CUs of its own.
// MMSA_DISABLE=group_order keeps the index order (A/B switch).
static void g2_group_order(G2Sched& s, const GemmParams* probs, int n, int bm, int bn, int tiles) {
  s.use_order = 0;
  static const bool off = mmsa_disabled("group_order");
  if (off || tiles > G2_ORDER_MAX || n < 2 || probs[0].gather != 0) return;
  int rider_of[G2_MAX_GROUPS];  // rider_of[g] = the later problem with one column tile that shares g's A operand, or -1
  bool is_rider[G2_MAX_GROUPS];
  for (int g = 0; g < n; ++g) { rider_of[g] = -1; is_rider[g] = false; }
  for (int g = 1; g < n; ++g)
    for (int h = 0; h < g; ++h)
      if (!is_rider[h] && rider_of[h] < 0 && probs[h].A == probs[g].A && probs[h].lda == probs[g].lda && probs[h].M == probs[g].M &&
          cdiv(probs[g].N, bn) == 1) {
        rider_of[h] = g; is_rider[g] = true;
        break;
      }
  const int run = tiles < 8 ? 1 : (tiles + 7) / 8;
  int target = 1;
  while ((target + 1) * (target + 1) <= 2 * run) ++target;  // ~sqrt(2 R) columns per block
  int pos = 0;
  for (int g = 0; g < n; ++g) {
    if (is_rider[g]) continue;
    const int ntm = cdiv(probs[g].M, bm), ntn = s.grp[g].ntn;
    const int nblk = ntn <= target + target / 2 ? 1 : cdiv(ntn, target);
    const int cb = cdiv(ntn, nblk);
    for (int c0 = 0; c0 < ntn; c0 += cb)
      for (int tm = 0; tm < ntm; ++tm) {
        for (int tn = c0; tn < ntn && tn < c0 + cb; ++tn) s.order[pos++] = (unsigned short)(s.grp[g].tile_begin + tm * ntn + tn);
        if (c0 == 0 && rider_of[g] >= 0) s.order[pos++] = (unsigned short)(s.grp[rider_of[g]].tile_begin + tm);
      }
  }
  if (pos != tiles) return;  // (cannot happen: every tile of every problem is dealt exactly once)
  s.use_order = 1;
}

// Grouped launch: n (2..G2_MAX_GROUPS) independent weight-gradient problems C_g[M_g, N_g] = A_g^T B_g with the same K, both
// operands k-major, fp32 output: one persistent launch walks the tiles of all problems. Either plain TN problems (any M_g, N_g)
// or implicit-GEMM weight gradients of convolutions with ONE geometry (gather 2: same M, N and ConvGeom; only the pointers
// differ). Without a workspace (probs[0].ws == null): no K split, overwrite only — the BERT layer group; colsum[g] (optional)
// receives sum_k A_g[k][m] = the bias gradient of that Linear. With probs[0].ws / ws_bytes: a K split common to the group is
// planned, the tiles store slabs and gemm2_group_reduce_kernel finishes (overwrite or accumulate). Returns
// MMSA_ERR_UNSUPPORTED when the problems do not fit (the caller then launches them one by one).
int gemm2_launch_group(const GemmParams* probs, float* const* colsum, int n, hipStream_t st) {
  if (n < 1 || n > G2_MAX_GROUPS) return MMSA_ERR_UNSUPPORTED;
  const int K = probs[0].K, gather = probs[0].gather;
  float* ws = probs[0].ws;
  const size_t ws_bytes = ws ? (size_t)probs[0].ws_bytes : 0;
  const int accumulate = probs[0].accumulate;
  if (gather != 0 && gather != 2) return MMSA_ERR_UNSUPPORTED;
  if (accumulate && !ws) return MMSA_ERR_UNSUPPORTED;
  long sum_mn = 0;
  for (int g = 0; g < n; ++g) {
    const GemmParams& q = probs[g];
    if (!(q.a_kmajor && q.b_kmajor) || q.gather != gather || q.K != K || !q.out_f32 || q.accumulate != accumulate || q.bias ||
        q.C2 || q.mul || q.add || q.act != MMSA_ACT_NONE || q.c_gw > 0 || q.colstat || !gemm2_eligible(q))
      return MMSA_ERR_UNSUPPORTED;
    if (gather == 2 && (q.M != probs[0].M || q.N != probs[0].N || q.lda != probs[0].lda || q.ldb != probs[0].ldb ||
                        memcmp(&q.g, &probs[0].g, sizeof(q.g)) != 0))
      return MMSA_ERR_UNSUPPORTED;
    sum_mn += (long)q.M * q.N;
  }
  const int cus = g2_num_cus();
  const int nsteps = K / G2_BK;
  // one tile shape and one K split for the whole group (cost model of g2_plan_search; without a workspace: the widest
  // 256-row tile whose total tile count fills the chip best, as before)
  G2Plan plan{4, 4, 1};
  {
    const G2Model& mdl = g2_model();
    double best = 1e300;
    static const int cfgs[5][2] = {{4, 4}, {4, 3}, {4, 2}, {2, 2}, {2, 1}};
    int smax = 1;
    if (ws) {
      smax = nsteps / 2;
      const long cap = (long)(ws_bytes / ((size_t)sum_mn * sizeof(float)));
      if (cap < smax) smax = (int)cap;
      if (smax > 128) smax = 128;
      if (smax < 1) smax = 1;
      if (accumulate && smax < 1) return MMSA_ERR_UNSUPPORTED;
    }
    for (int ci = 0; ci < (ws ? 5 : 3); ++ci) {
      const G2Plan shape{cfgs[ci][0], cfgs[ci][1], 1};
      const int bm = g2_bm(shape), bn = g2_bn(shape), w32 = bn / 32;
      long tiles = 0;
      for (int g = 0; g < n; ++g) tiles += (long)cdiv(probs[g].M, bm) * cdiv(probs[g].N, bn);
      const double t_step = shape.wm == 4 ? mdl.a4 + mdl.b4 * w32 : mdl.a2 + mdl.b2 * w32;
      const double t_item = shape.wm == 4 ? 0.3 + 0.2 * w32 : 0.3 + 0.1 * w32;
      for (int sp = (accumulate ? 2 : 1); sp <= (smax > 1 ? smax : (accumulate ? 2 : 1)); ++sp) {
        const int per = cdiv(nsteps, sp);
        const int split = cdiv(nsteps, per);
        if (split != sp) continue;  // (no empty slices)
        const long items = tiles * split;
        const long rounds = (items + cus - 1) / cus;
        double cost = (double)rounds * (per * t_step + t_item);
        if (split > 1) cost += mdl.c_split + mdl.d_split * split * (double)sum_mn * 4.0 / 4e6;
        if (cost < best - 1e-9) { best = cost; plan = G2Plan{shape.wm, shape.nj, split}; }
      }
    }
    if (best >= 1e300) return MMSA_ERR_UNSUPPORTED;
    if (plan.split > 1 && (size_t)plan.split * sum_mn * sizeof(float) > ws_bytes) return MMSA_ERR_UNSUPPORTED;
    if (plan.split > 1 && (long)plan.split * sum_mn * 4 >= 0x7FFFFFF0L) return MMSA_ERR_UNSUPPORTED;
  }
  G2Sched s;
  memset(&s, 0, sizeof(s));
  const int bm = g2_bm(plan), bn = g2_bn(plan);
  s.ngroups = n;
  int tiles = 0;
  long slab_off = 0;
  G2GroupReduce red;
  memset(&red, 0, sizeof(red));
  for (int g = 0; g < n; ++g) {
    const GemmParams& q = probs[g];
    G2Sched::Group& gr = s.grp[g];
    gr.A = q.A; gr.B = q.B; gr.C = q.C; gr.colsum = colsum ? colsum[g] : nullptr;
    gr.M = q.M; gr.N = q.N; gr.lda = q.lda; gr.ldb = q.ldb; gr.ldc = q.ldc;
    gr.tile_begin = tiles;
    gr.ntn = cdiv(q.N, bn);
    tiles += cdiv(q.M, bm) * gr.ntn;
    long ea, eb;
    g2_extents(q, &ea, &eb);
    gr.a_bytes = (unsigned)ea; gr.b_bytes = (unsigned)eb;
    gr.c_bytes = (unsigned)(((long)(q.M - 1) * q.ldc + q.N) * 4);
    if (plan.split > 1) {  // the tiles of this problem store into its own slab region
      const long mn = (long)q.M * q.N;
      gr.C = ws + slab_off;
      gr.c_bytes = (unsigned)((long)plan.split * mn * 4);
      red.g[g].slab = ws + slab_off; red.g[g].C = (float*)q.C; red.g[g].mn = mn; red.g[g].ldc = q.ldc;
      red.g[g].begin4 = red.total4; red.g[g].N = q.N;
      red.total4 += mn / 4;
      slab_off += (long)plan.split * mn;
    }
  }
  g2_group_order(s, probs, n, bm, bn, tiles);
  s.ntm = 1; s.ntn = 1; s.ntiles = tiles;
  s.nsteps = nsteps;
  s.split_k = plan.split;
  s.per = cdiv(nsteps, plan.split);
  s.items = tiles * plan.split;
  s.fd_ntiles = make_fastdiv((uint32_t)tiles);
  s.fd_ntn = make_fastdiv(1);
  s.fd_cb = make_fastdiv(1); s.fd_sbc = make_fastdiv(1);
  s.fast = 1;
  s.stamp = probs[0].stamp;
  s.c_bytes = s.grp[0].c_bytes;
  if (const char* d = MMSA_EXP_ENV("MMSA_G2_DBG")) s.dbg = atoi(d);
  GemmParams p = probs[0];
  p.split_k = plan.split;
  p.accumulate = 0;
  p.ws = ws;
  p.a_bytes = s.grp[0].a_bytes; p.b_bytes = s.grp[0].b_bytes;
  const int grid = s.items < cus ? s.items : cus;
  g2_last_plan[0] = plan.wm; g2_last_plan[1] = plan.nj; g2_last_plan[2] = plan.split;
  int rc;
  if (plan.wm == 4) {
    if (plan.nj == 4) rc = g2_launch_nj_44(p, s, grid, st);
    else if (plan.nj == 3) rc = g2_launch_nj_43(p, s, grid, st);
    else rc = g2_launch_nj_42(p, s, grid, st);
  } else {
    if (plan.nj == 2) rc = g2_launch_nj_22(p, s, grid, st);
    else rc = g2_launch_nj_21(p, s, grid, st);
  }
  if (rc) return rc;
  if (plan.split > 1) {
    red.n = n; red.split = plan.split; red.accumulate = accumulate; red.stamp = probs[0].stamp;
    long blocks = (red.total4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm2_group_reduce_kernel, dim3((int)blocks), dim3(256), 0, st, red);
    MMSA_CHECK_LAUNCH();
  }
  return MMSA_OK;
}
#endif  // !G2_PART
