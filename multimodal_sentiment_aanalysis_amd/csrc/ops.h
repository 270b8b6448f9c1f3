// Internal launcher prototypes (C++ linkage). Every launcher enqueues on `st`, allocates nothing, never
// synchronizes, and returns an MMSA_* status. `dtype` is the activation storage type (MMSA_F32 / MMSA_BF16).
#pragma once
#include "common.h"
#include "gemm.h"

int gemm_bf16_simt_launch(const GemmParams& p, hipStream_t st);

// gemm_fp8.hip
size_t fp8_quantize_ws_bytes();
int fp8_quantize(const void* x_bf16, long n, void* out_e4m3, float* scale, unsigned* amax_ws, hipStream_t st);
int fp8_quantize_rows(const void* x_bf16, long ldx, int M, int K, void* out_e4m3, float* scales, hipStream_t st);
#define FP8_BATCH_MAX 128
size_t fp8_quantize_batch_ws_bytes(int n);
int fp8_quantize_batch(const void* base_bf16, const long* off, const long* numel, int n, void* out_e4m3, float* scales, float* ws,
                       hipStream_t st);
bool gemm_fp8_eligible(const GemmParams& p);
int gemm_fp8_launch(const GemmParams& p, const float* scale_a, const float* scale_b, hipStream_t st);

// rowops.hip
int partial_finalize(const float* part, int nblk, long stride, int n, float* out, int accumulate, float scale,
                     hipStream_t st);
int layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                  int M, int H, float eps, hipStream_t st, void* q_out = nullptr, float* q_scales = nullptr);
size_t layernorm_bwd_ws_bytes(int H);
// deferred finalize of a LayerNorm backward (rowops.hip): what layernorm_bwd records instead of launching it when `defer` is given
struct LnFinJob { const float* part; int nblk, nq, accumulate; float *out0, *out1, *out2; };
#define LN_FIN_MAX 32
struct LnFinBatch { int H; LnFinJob job[LN_FIN_MAX]; };
int layernorm_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                  void* dx, float* dgamma, float* dbeta, int accumulate, float* ws, int M, int H, hipStream_t st,
                  float* dxsum = nullptr, LnFinJob* defer = nullptr);
int layernorm_bwd_finalize_batch(const LnFinJob* jobs, int n, int H, hipStream_t st);
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
int tanh_bwd(int dtype, const void* dy, const void* y, void* dx, long n, hipStream_t st);

// dst[i] = src[8 i] for one or two (src, dst, n) pairs: picks the bias gradients out of the grouped launch's [N][8] results
int bias_pick(const float* src, float* dst, int n, const float* src2, float* dst2, int n2, hipStream_t st);
struct BiasPickJob { const float* src; float* dst; int n; };
#define BIAS_PICK_MAX 32
struct BiasPickBatch { BiasPickJob job[BIAS_PICK_MAX]; };
int bias_pick_batch(const BiasPickJob* jobs, int n, hipStream_t st);
int fill_ones_bf16(void* dst, long n, hipStream_t st);

// contrastive.hip (N1)
size_t contrastive_ws_bytes(int B, int D);
int infonce_fwd_bwd(const float* feat1, const float* feat2, const long long* labels, const float* temperature, float* loss,
                    float* dfeat1, float* dfeat2, float* dtemp, int B, int D, float grad_scale, float* ws, hipStream_t st);
int supcon_fwd_bwd(const float* z1, const float* z2, const long long* labels, float temperature, float* loss, float* dz1,
                   float* dz2, int B, int D, float grad_scale, float* ws, hipStream_t st);

int bn_fold(int dtype, const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
            int C, float* scale, float* shift, const float* w, void* wout, int K, int ld_out, hipStream_t st);
// clockprobe.hip
int mfma_clock_probe(void* ws, int blocks, int iters, int launches, hipStream_t st);

// bnops.hip
size_t bn_ws_bytes(int C);
int bn_forward(int dtype, const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
               float* mean, float* invstd, const void* res, void* y, float* ws, int M, int C, float eps, float momentum,
               int act, int training, hipStream_t st, unsigned char* relu_mask = nullptr, const float* pre_part = nullptr,
               int pre_rows = 0);
// pre_part / pre_rows (training): per-slice column sums [pre_rows][2][C] of x already taken by its producer (the convolution
// GEMM's epilogue, GemmParams::colstat): the statistics pass over x is skipped, the finalize reads these partials.
int bn_backward(int dtype, const void* dy, const void* x, const void* y, const float* mean, const float* invstd,
                const float* gamma, const float* beta, void* dx, void* dres, float* dgamma, float* dbeta, int accumulate,
                float* ws, int M, int C, int act, int training, hipStream_t st, const unsigned char* relu_mask = nullptr);
int bn_small_backward_unit(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, float* dx, float* dgamma, float* dbeta, int accumulate, int M, int C, int act,
                           int training, const unsigned char* drop_mask, float drop_p, int relu_in, hipStream_t st);
// relu_mask (optional, act == RELU): [M][C/8] bytes, bit e of byte (m, c/8) = output (m, c+e) > 0. Written by the forward,
// read by the backward instead of the saved output y (1/16 of its bytes in two streamed passes).

// poolops.hip
int stem_im2col(int dtype, const float* img, void* col, int B, int Cin, int H, int W, int OH, int OW, int KH, int KW,
                int stride, int pad, int Kpad, hipStream_t st);
int maxpool_fwd(int dtype, const void* x, void* y, unsigned char* idx, int B, int H, int W, int C, hipStream_t st);
int maxpool_bwd(int dtype, const void* dy, const unsigned char* idx, void* dx, int B, int H, int W, int C, hipStream_t st);
int avgpool_fwd(int dtype, const void* x, void* y, int B, int HW, int C, hipStream_t st);
int avgpool_bwd(int dtype, const void* dy, void* dx, int B, int HW, int C, hipStream_t st);
int cast_f32(int dtype, const float* src, void* dst, long n, hipStream_t st);
int widen_bf16(const void* src_bf16, float* dst, long n, hipStream_t st);
int pad_rows(int dtype, const float* src, void* dst, int rows, int cols_src, int cols_dst, hipStream_t st);
int unpad_rows(const float* src, float* dst, int rows, int cols_src, int cols_dst, int accumulate, hipStream_t st);

// headops.hip (fp32)
int l2norm_fwd(const float* x, float* y, float* nrm, int M, int E, float eps, hipStream_t st);
int l2norm_fwd_ld(const float* x, float* y, long ldy, float* nrm, int M, int E, float eps, hipStream_t st);
int l2norm_bwd_ld(const float* dy, const float* y, long ldin, const float* nrm, float* dx, int M, int E, float eps,
                  int accumulate, hipStream_t st);
int cat3_fwd(const float* a, const float* b, const float* c, float* out, int rows, int E, hipStream_t st);
int cat3_bwd(const float* dcat, const float* add0, float* da, float* db, float* dc, int rows, int E, hipStream_t st);
int l2norm_bwd(const float* dy, const float* y, const float* nrm, float* dx, int M, int E, float eps, int accumulate,
               hipStream_t st);
int mha_core_fwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* ctx, long ldc,
                 float* probs, int B, int Lq, int Lk, int E, int heads, hipStream_t st);
int mha_core_bwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, const float* probs,
                 const float* dctx, long ldc, float* dq, long lddq, float* dk, long lddk, float* dv, long lddv, int B,
                 int Lq, int Lk, int E, int heads, hipStream_t st);
int seq_pool_fwd(const float* x, float* y, unsigned char* idx, int B, int L, int E, int mode, hipStream_t st);
int seq_pool_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int L, int E, int mode, hipStream_t st);
// head_fused.hip (MMSA_DISABLE=head_units: the separate launches)
bool head_units_on();
// Linear -> BatchNorm1d (training statistics over <= 256 rows) -> activation [-> Dropout] unit as one launch; K in {128, 256, 768},
// N % 16 == 0, else MMSA_ERR_UNSUPPORTED (the caller launches the separate kernels)
struct UnitFwd {
  const float* x; long ldx;                 // [M][K]
  const float *W, *bias;                    // [N][K], [N]
  const float *gamma, *beta;
  float *rmean, *rvar;                      // running statistics (or null)
  float *z;                                 // the Linear's output after lin_act (what the BatchNorm backward reads)
  float *y;                                 // bn_act(BatchNorm(z)) (or null when only the Dropout output is kept)
  float *yd; unsigned char* mask;           // Dropout output + keep bytes (both null: no Dropout)
  float *out2;                              // a second copy of the unit's final output (or null)
  float *mean, *invstd;
  int M, N, lin_act, bn_act;
  float eps, momentum, drop_p;
  unsigned long long seed;
};
int unit_fwd_fused(const UnitFwd& p, int K, hipStream_t st);
int gate_mix_fwd(const float* g, const float* q, long ldq, const float* a, long lda, float* mix, int B, int E, hipStream_t st);
int gate_mix_bwd(const float* dmix, const float* g, const float* q, long ldq, const float* a, long lda, float* dq, float* da,
                 float* dgpre, int B, int E, hipStream_t st);
int weighted_concat_fwd(const float* wl, const float* f1, const float* f2, const float* f3, float* w, float* out, int B, int E,
                        hipStream_t st);
int weighted_concat_bwd(const float* dout, const float* w, const float* f1, const float* f2, const float* f3, float* df1,
                        float* df2, float* df3, float* dwl, int B, int E, hipStream_t st);
int ce_fwd_bwd(const float* logits, const long long* labels, float* loss, float* dlogits, float* probs, int B, int C,
               float grad_scale, hipStream_t st);
int dropout_fwd(const float* x, float* y, unsigned char* mask, long n, float p, unsigned long long seed, hipStream_t st);
int dropout_bwd(const float* dy, const unsigned char* mask, float* dx, long n, float p, hipStream_t st);
#define EW_COPY 0
#define EW_ADD 1
#define EW_RELU_BWD 2
#define EW_GELU_BWD 3
int ew2d(int op, const float* a, long lda, const float* b, long ldb, float* out, long ldo, int rows, int cols, hipStream_t st);

// optim.hip
size_t grad_norm_ws_bytes();
int grad_norm(const float* g, long n, float grad_scale, float max_norm, float* norm_out, void* ws, hipStream_t st,
              const float* loss = nullptr, int* step_count = nullptr, float b1 = 0.9f, float b2 = 0.999f);
int grad_norm_ranges(const float* g, const long* offs, const long* lens, int nr, float grad_scale, float max_norm,
                     float* norm_out, void* ws, hipStream_t st, const float* loss, int* step_count, float b1 = 0.9f,
                     float b2 = 0.999f);
int grad_sumsq_ranges(const float* g, const long* offs, const long* lens, int nr, double* sumsq, void* ws, hipStream_t st);
int grad_norm_from_sumsq(const double* sumsq, int n, float grad_scale, float max_norm, float* norm_out, hipStream_t st,
                         const float* loss, int* step_count, float b1, float b2);
int grad_scale_clip(float* g, long n, const float* norm_clip, hipStream_t st);
int adamw_step(float* w, const float* g, float* m, float* v, void* w16, long n, float lr, float b1, float b2, float eps,
               float wd, int step, const float* norm_clip, float grad_scale, hipStream_t st, const int* step_count = nullptr);
