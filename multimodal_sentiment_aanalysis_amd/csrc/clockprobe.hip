// Sustained matrix-core clock of THIS chip under a dense bf16 MFMA load (bench.py prints the peak derived from it next to the
// datasheet peak: SURVEY.md section 8(d) "restate the exact figure from the CU count x sustained clock x MFMA FLOP/CU/clk").
// Every wave runs back-to-back v_mfma_f32_16x16x32_bf16 on non-trivial operands (16 independent accumulators, one wave per SIMD)
// and stamps the shader clock (s_memtime, one tick per shader cycle) and the constant 100 MHz clock (s_memrealtime) around the
// loop: clock = d(memtime) / d(memrealtime) * 100 MHz (CDNA4 guide, 'DVFS give-back' item 6). The stamps go to a buffer of their
// own; the accumulators are written once so that the loop is not dead code.
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float cp_f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 cp_bf16x8;

__global__ __launch_bounds__(256) void mfma_clock_probe_kernel(int iters, unsigned long long* __restrict__ stamps, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  cp_bf16x8 a[4], b[4];  // 4 x 4 distinct operand pairs: identical accumulators would be merged by the compiler
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int e = 0; e < 8; ++e) {  // full-range pseudo-random operands (zero or constant operands let the chip clock higher)
      const unsigned h = (unsigned)(lane * 8 + e + 1 + f * 4099u + blockIdx.x * 977u) * 2654435761u;
      a[f][e] = (__bf16)(((int)(h >> 16 & 1023) - 512) * (1.0f / 256.0f));
      b[f][e] = (__bf16)(((int)(h >> 6 & 1023) - 512) * (1.0f / 256.0f));
    }
  cp_f32x4 acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = cp_f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  // inline asm with the accumulator as an in-place operand: written with the builtin, hipcc rotated the loop-carried accumulators
  // through v_accvgpr moves (a read-after-MFMA stall in every iteration: the "bare" loop then measured the copies, not the pipe)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
#ifdef PROBE_VGPR_ACC  // experiment: accumulators in ArchVGPRs (what hipcc picks for the GEMM kernels) instead of AccVGPRs
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a[k >> 2]), "v"(b[k & 3]));
#else
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(a[k >> 2]), "v"(b[k & 3]));
#endif
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // MFMA result -> VALU read: the wait states hipcc would have inserted itself
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  cp_f32x4 s = acc[0];
#pragma unroll
  for (int k = 1; k < 16; ++k) s += acc[k];
  sink[(size_t)blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x + 0] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

// ws: >= blocks * (16 + 1024) bytes of device scratch. Launches `launches` back-to-back probes of `iters` x 16 MFMAs per wave on
// `blocks` workgroups (one per CU: 4 waves, one per SIMD); the caller reads the stamps of the LAST launch after synchronizing:
// stamps[2b] = shader cycles, stamps[2b + 1] = 100 MHz ticks of workgroup b.
int mfma_clock_probe(void* ws, int blocks, int iters, int launches, hipStream_t st) {
  if (!ws || blocks <= 0 || iters <= 0 || launches <= 0) return MMSA_ERR_ARG;
  unsigned long long* stamps = (unsigned long long*)ws;
  float* sink = (float*)((char*)ws + (size_t)blocks * 16);
  for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(mfma_clock_probe_kernel, dim3(blocks), dim3(256), 0, st, iters, stamps, sink);
  MMSA_CHECK_LAUNCH();
  return MMSA_OK;
}
