// Internal launcher prototypes (C++ linkage). Every launcher enqueues on `st`, allocates nothing, never
// synchronizes, and returns an MMSA_* status. `dtype` is the activation storage type (MMSA_F32 / MMSA_BF16).
#pragma once
#include "common.h"
#include "gemm.h"

int gemm_bf16_simt_launch(const GemmParams& p, hipStream_t st);

// rowops.hip
int partial_finalize(const float* part, int nblk, long stride, int n, float* out, int accumulate, float scale,
                     hipStream_t st);
int layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                  int M, int H, float eps, hipStream_t st);
size_t layernorm_bwd_ws_bytes(int H);
int layernorm_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                  void* dx, float* dgamma, float* dbeta, int accumulate, float* ws, int M, int H, hipStream_t st);
size_t colsum_ws_bytes(int N);
int colsum(int dtype, const void* x, long ldx, float* out, int accumulate, float* ws, int M, int N, hipStream_t st);
int embed_gather(int dtype, const long long* ids, const void* word, const void* pos, const void* type, void* e, int M,
                 int S, int H, int vocab, hipStream_t st);
int embed_backward(int dtype, const long long* ids, const void* de, float* dword, float* dpos, float* dtype0,
                   int accumulate, float* ws, int B, int S, int H, int vocab, int maxpos, hipStream_t st);

// attention.hip
size_t attention_bwd_ws_bytes(int B, int S, int heads);
int attention_fwd(int impl, const void* qkv, const float* mask, void* ctx, int B, int S, int heads, int head_dim,
                  hipStream_t st);
int attention_bwd(int impl, const void* qkv, const float* mask, const void* dctx, void* dqkv, float* ws, int B, int S,
                  int heads, int head_dim, hipStream_t st);
