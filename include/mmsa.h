/*
 * mmsa.h — C ABI of libmmsa_hip.so: the MI355X (gfx950) kernels behind the text+image sentiment hot path.
 *
 * The reference (zhouyuchenzyccccc/Multimodal-Sentiment-Aanalysis, MML_ZYC/) has no FFI/operator API: its
 * boundary is duck-typed Python (`model(x1, x2, x3[, labels])`). Every entry point below
 * therefore names the reference call site whose arithmetic it replaces; the Python side
 * (multimodal_sentiment_aanalysis_amd/) binds them with ctypes and wraps them in torch.autograd.Functions so
 * the reference's Trainer.py / Tester.py / train.py loops run unchanged on top.
 *
 * Conventions
 *  - plain pointers to DEVICE memory + sizes; no torch types; `stream` is a hipStream_t passed as void*.
 *  - every call only enqueues work on `stream`: no allocation, no synchronization, no host copies
 *    (hipGraph-capturable). Scratch memory is caller-provided (`ws`), sized by the matching *_ws_bytes().
 *  - return value: 0 = MMSA_OK, 1 = bad argument, 2 = launch failure, 3 = unsupported shape. No exceptions.
 *  - `dtype`: storage type of activations / working weights: 0 = fp32, 1 = bf16. Accumulation, statistics,
 *    parameters' gradients and optimizer state are always fp32.
 *  - thread-compatible: no global mutable state besides lazily-set function attributes; one caller thread
 *    per GPU process.
 */
#ifndef MMSA_H
#define MMSA_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMSA_OK 0
#define MMSA_ERR_ARG 1
#define MMSA_ERR_LAUNCH 2
#define MMSA_ERR_UNSUPPORTED 3

#define MMSA_F32 0
#define MMSA_BF16 1

/* GEMM implementation selector */
#define MMSA_GEMM_F32_SIMT 0  /* fp32 storage, VALU fma (exact-fp32 mode, fusion head) */
#define MMSA_GEMM_BF16_MFMA 1 /* bf16 storage, v_mfma_f32_16x16x32_bf16 */
#define MMSA_GEMM_BF16_SIMT 2 /* bf16 storage, VALU fma (on-device cross-check of the MFMA kernel) */

#define MMSA_ACT_NONE 0
#define MMSA_ACT_GELU 1 /* exact erf GELU: nn.GELU() at MultimodalModel.py:173,182,187,195 */
#define MMSA_ACT_RELU 2 /* nn.ReLU() at MultimodalModel.py:379,418,422,441 */
#define MMSA_ACT_TANH 3 /* BERT pooler */
#define MMSA_ACT_SIGMOID 4 /* gate, MultimodalModel.py:120 */

int mmsa_abi_version(void);

/* ---- implicit-GEMM convolution geometry (NHWC activations, weights [Cout][KH][KW][Cin]) ------------------- */
typedef struct mmsa_conv_geom {
  int32_t SH, SW;               /* source spatial size */
  int32_t GH, GW;               /* row-space grid: GEMM rows are (image, y, x) over GH x GW */
  int32_t KH, KW;
  int32_t mul, kmul, off, div;  /* source y = (y*mul + ky*kmul + off) / div, tap valid iff divisible & in range */
  int32_t cper;                 /* channels per tap along the gathered dimension */
  int64_t src_pix_stride;       /* elements between consecutive source pixels */
} mmsa_conv_geom;

/* ---- GEMM with fused epilogue -------------------------------------------------------------------------------
 * C[M,N] = epi( sum_k opA[m][k] * opB[k][n] ), fp32 accumulate.
 * Replaces every torch.nn.Linear / addmm / conv the hot path dispatches (BERT-base and ResNet-50
 * encoders that fill the encoder slot MultimodalModel.py:264-266) and their data / weight gradients.
 *   a_kmajor = 0: A is [M][K] (lda = row stride); 1: A is [K][M]
 *   b_kmajor = 0: B is [N][K] (nn.Linear weight [out,in]); 1: B is [K][N]
 *   gather   = 0 none; 1 = A rows are convolution pixels (forward conv / data gradient); 2 = B rows are pixels
 *              (weight gradient). See csrc/gemm.h for the exact addressing.
 * epilogue: v = acc (+ bias[n]); C2 = v (optional); v = act(v); v *= gelu'(mul[m][n]) (optional);
 *           v += add[m][n] (optional); C = v (storage dtype, or fp32 if out_f32; += if accumulate).
 * split_k > 1 needs ws of mmsa_gemm_ws_bytes(M, N, split_k) bytes.
 */
typedef struct mmsa_gemm_desc {
  const void* A;
  const void* B;
  void* C;
  int32_t M, N, K;
  int64_t lda, ldb, ldc;
  int32_t a_kmajor, b_kmajor, gather;
  int64_t b_tap_stride;
  mmsa_conv_geom geom;
  const float* bias;
  void* C2;
  int64_t ldc2;
  int32_t act;
  const void* mul;
  int64_t ldmul;
  const void* add;
  int64_t ldadd;
  int32_t out_f32, accumulate, split_k;
  float* ws;
} mmsa_gemm_desc;

size_t mmsa_gemm_ws_bytes(int32_t M, int32_t N, int32_t split_k);
int mmsa_gemm(const mmsa_gemm_desc* d, int32_t impl, void* stream);

/* ---- LayerNorm (nn.LayerNorm: MultimodalModel.py:122,149 eps 1e-5; BERT eps 1e-12) ------------------------- */
int mmsa_layernorm_fwd(int32_t dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                       float* rstd, int32_t M, int32_t H, float eps, void* stream);
size_t mmsa_layernorm_bwd_ws_bytes(int32_t H);
int mmsa_layernorm_bwd(int32_t dtype, const void* dy, const void* x, const float* mean, const float* rstd,
                       const float* gamma, void* dx, float* dgamma, float* dbeta, int32_t accumulate, float* ws,
                       int32_t M, int32_t H, void* stream);

/* ---- column sum: bias gradients of every Linear ------------------------------------------------------------- */
size_t mmsa_colsum_ws_bytes(int32_t N);
int mmsa_colsum(int32_t dtype, const void* x, int64_t ldx, float* out, int32_t accumulate, float* ws, int32_t M,
                int32_t N, void* stream);

/* ---- BERT self-attention core, head_dim 64: softmax(QK^T/8 + mask)V on the packed QKV projection ------------
 * impl as MMSA_GEMM_* (0 fp32 SIMT, 1 bf16 MFMA, 2 bf16 SIMT). qkv [B*S][3*heads*64], ctx [B*S][heads*64],
 * mask [B][S] fp32 (1 keep / 0 masked) or NULL. Backward writes d_qkv packed like qkv. */
size_t mmsa_attention_bwd_ws_bytes(int32_t B, int32_t S, int32_t heads);
int mmsa_attention_fwd(int32_t impl, const void* qkv, const float* mask, void* ctx, int32_t B, int32_t S, int32_t heads,
                       int32_t head_dim, void* stream);
int mmsa_attention_bwd(int32_t impl, const void* qkv, const float* mask, const void* dctx, void* dqkv, float* ws,
                       int32_t B, int32_t S, int32_t heads, int32_t head_dim, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMSA_H */
