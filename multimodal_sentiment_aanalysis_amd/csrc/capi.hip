// extern "C" surface of libmmsa_hip.so (declared in include/mmsa.h). Thin: validates, converts the plain C
// descriptors to the internal ones, and calls the launchers.
#include "../../include/mmsa.h"
#include <string.h>
#include "ops.h"

int gemm_prof_begin(int max_records);
int gemm_prof_sample(int stride, int phase);
int gemm_prof_mode(int mode);
int gemm_prof_end(double* total_ms, double* total_flop, long* launches);
double gemm_prof_last_bytes();

static GemmParams to_params(const mmsa_gemm_desc* d) {
  GemmParams p;
  memset(&p, 0, sizeof(p));  // every field the descriptor does not carry (epilogue extras, column statistics, stamps) is off
  p.A = d->A; p.B = d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.a_kmajor = d->a_kmajor; p.b_kmajor = d->b_kmajor; p.gather = d->gather;
  p.b_tap_stride = d->b_tap_stride; p.b_tap_stride_y = 0;
  p.c2_gelu_grad = 0; p.mul_is_factor = 0;
  p.scale_a = p.scale_b = nullptr;
  p.b_ones = 0;
  const mmsa_conv_geom& s = d->geom;
  ConvGeom& g = p.g;
  g.SH = s.SH; g.SW = s.SW; g.GH = s.GH; g.GW = s.GW; g.KH = s.KH; g.KW = s.KW;
  g.mul = s.mul; g.kmul = s.kmul; g.off = s.off; g.offx = s.off; g.div = s.div; g.cper = s.cper;
  g.src_pix_stride = s.src_pix_stride;
  g.fd_gw = make_fastdiv(s.GW > 0 ? s.GW : 1);
  g.fd_ghw = make_fastdiv(s.GH * s.GW > 0 ? s.GH * s.GW : 1);
  g.fd_kw = make_fastdiv(s.KW > 0 ? s.KW : 1);
  g.fd_cper = make_fastdiv(s.cper > 0 ? s.cper : 1);
  p.bias = d->bias; p.C2 = d->C2; p.ldc2 = d->ldc2; p.act = d->act;
  p.mul = d->mul; p.ldmul = d->ldmul; p.add = d->add; p.ldadd = d->ldadd;
  p.out_f32 = d->out_f32; p.accumulate = d->accumulate; p.split_k = d->split_k; p.ws = d->ws; p.ws_bytes = 0;
  p.zero_page = nullptr;
  p.a_bytes = p.b_bytes = 0; p.use_srd = 0; p.stamp = nullptr; p.c_gw = 0; p.c_gh = 0;
  return p;
}

extern "C" {

int mmsa_abi_version(void) { return 1; }

size_t mmsa_gemm_ws_bytes(int32_t M, int32_t N, int32_t split_k) { return gemm_splitk_ws_bytes(M, N, split_k); }

int mmsa_gemm(const mmsa_gemm_desc* d, int32_t impl, void* stream) {
  if (!d || !d->A || !d->B || !d->C) return MMSA_ERR_ARG;
  if (d->gather && (d->geom.GH <= 0 || d->geom.GW <= 0 || d->geom.KW <= 0 || d->geom.cper <= 0 || d->geom.div <= 0))
    return MMSA_ERR_ARG;
  const GemmParams p = to_params(d);
  hipStream_t st = (hipStream_t)stream;
  switch (impl) {
    case MMSA_GEMM_F32_SIMT: return gemm_f32_launch(p, st);
    case MMSA_GEMM_BF16_MFMA: return gemm_bf16_launch(p, st);
    case MMSA_GEMM_BF16_SIMT: return gemm_bf16_simt_launch(p, st);
    case MMSA_GEMM_F32_MFMA: return p.c_gw > 0 ? MMSA_ERR_UNSUPPORTED : gemm_f32_mfma_launch(p, st);
    case MMSA_GEMM_F32_VALU: return gemm_f32_valu_launch(p, st);
    default: return MMSA_ERR_ARG;
  }
}

static int gemm_group_impl(const mmsa_gemm_desc* d, int32_t n, float* ws, size_t ws_bytes, void* stream) {
  if (!d || n < 1 || n > GEMM_MAX_GROUPS) return MMSA_ERR_ARG;
  GemmParams ps[GEMM_MAX_GROUPS];
  for (int g = 0; g < n; ++g) {
    if (!d[g].A || !d[g].B || !d[g].C) return MMSA_ERR_ARG;
    if (d[g].gather && (d[g].geom.GH <= 0 || d[g].geom.GW <= 0 || d[g].geom.KW <= 0 || d[g].geom.cper <= 0 || d[g].geom.div <= 0))
      return MMSA_ERR_ARG;
    ps[g] = to_params(&d[g]);
    ps[g].split_k = 1;
    ps[g].ws = nullptr;
  }
  ps[0].ws = ws;
  ps[0].ws_bytes = (long)ws_bytes;
  return gemm_bf16_launch_group(ps, nullptr, n, (hipStream_t)stream);
}
int mmsa_gemm_group(const mmsa_gemm_desc* d, int32_t n, void* stream) { return gemm_group_impl(d, n, nullptr, 0, stream); }
int mmsa_gemm_group_split(const mmsa_gemm_desc* d, int32_t n, float* ws, size_t ws_bytes, void* stream) {
  if (!ws || ws_bytes == 0) return MMSA_ERR_ARG;
  return gemm_group_impl(d, n, ws, ws_bytes, stream);
}

size_t mmsa_fp8_quantize_ws_bytes(void) { return fp8_quantize_ws_bytes(); }
int mmsa_fp8_quantize(const void* x_bf16, int64_t n, void* out_e4m3, float* scale, void* amax_ws, void* stream) {
  return fp8_quantize(x_bf16, n, out_e4m3, scale, (unsigned*)amax_ws, (hipStream_t)stream);
}
int mmsa_gemm_fp8(const mmsa_gemm_desc* d, const float* scale_a, const float* scale_b, void* stream) {
  if (!d || !d->A || !d->B || !d->C || !scale_a || !scale_b) return MMSA_ERR_ARG;
  return gemm_fp8_launch(to_params(d), scale_a, scale_b, (hipStream_t)stream);
}
int mmsa_fp8_quantize_rows(const void* x_bf16, int64_t ldx, int32_t M, int32_t K, void* out_e4m3, float* row_scales, void* stream) {
  return fp8_quantize_rows(x_bf16, (long)ldx, M, K, out_e4m3, row_scales, (hipStream_t)stream);
}
int mmsa_gemm_fp8_rows(const mmsa_gemm_desc* d, const float* row_scales_a, const float* scale_b, void* stream) {
  if (!d || !d->A || !d->B || !d->C || !row_scales_a || !scale_b) return MMSA_ERR_ARG;
  GemmParams p = to_params(d);
  p.scale_a_rows = 1;
  return gemm_fp8_launch(p, row_scales_a, scale_b, (hipStream_t)stream);
}
size_t mmsa_fp8_quantize_batch_ws_bytes(int32_t n) { return fp8_quantize_batch_ws_bytes(n); }
int mmsa_fp8_quantize_batch(const void* base_bf16, const int64_t* offsets, const int64_t* numel, int32_t n, void* out_e4m3,
                            float* scales, float* ws, void* stream) {
  if (!offsets || !numel || n <= 0 || n > FP8_BATCH_MAX) return MMSA_ERR_ARG;
  long off[FP8_BATCH_MAX], num[FP8_BATCH_MAX];
  for (int t = 0; t < n; ++t) { off[t] = (long)offsets[t]; num[t] = (long)numel[t]; }
  return fp8_quantize_batch(base_bf16, off, num, n, out_e4m3, scales, ws, (hipStream_t)stream);
}

int mmsa_layernorm_fwd(int32_t dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                       float* rstd, int32_t M, int32_t H, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || M <= 0) return MMSA_ERR_ARG;
  return layernorm_fwd(dtype, x, gamma, beta, y, mean, rstd, M, H, eps, (hipStream_t)stream);
}
size_t mmsa_layernorm_bwd_ws_bytes(int32_t H) { return layernorm_bwd_ws_bytes(H); }
int mmsa_layernorm_bwd(int32_t dtype, const void* dy, const void* x, const float* mean, const float* rstd,
                       const float* gamma, void* dx, float* dgamma, float* dbeta, int32_t accumulate, float* ws,
                       int32_t M, int32_t H, void* stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !ws || M <= 0) return MMSA_ERR_ARG;
  return layernorm_bwd(dtype, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, accumulate, ws, M, H, (hipStream_t)stream);
}

size_t mmsa_colsum_ws_bytes(int32_t N) { return colsum_ws_bytes(N); }
int mmsa_colsum(int32_t dtype, const void* x, int64_t ldx, float* out, int32_t accumulate, float* ws, int32_t M,
                int32_t N, void* stream) {
  if (!x || !out || !ws || M <= 0 || N <= 0) return MMSA_ERR_ARG;
  return colsum(dtype, x, ldx, out, accumulate, ws, M, N, (hipStream_t)stream);
}

size_t mmsa_attention_bwd_ws_bytes(int32_t B, int32_t S, int32_t heads) { return attention_bwd_ws_bytes(B, S, heads); }
int mmsa_attention_fwd(int32_t impl, const void* qkv, const float* mask, void* ctx, int32_t B, int32_t S, int32_t heads,
                       int32_t head_dim, void* stream) {
  if (!qkv || !ctx || B <= 0 || S <= 0 || heads <= 0) return MMSA_ERR_ARG;
  return attention_fwd(impl, qkv, mask, ctx, B, S, heads, head_dim, (hipStream_t)stream);
}
int mmsa_attention_bwd(int32_t impl, const void* qkv, const float* mask, const void* dctx, void* dqkv, float* ws,
                       int32_t B, int32_t S, int32_t heads, int32_t head_dim, void* stream) {
  if (!qkv || !dctx || !dqkv || !ws || B <= 0 || S <= 0 || heads <= 0) return MMSA_ERR_ARG;
  return attention_bwd(impl, qkv, mask, dctx, dqkv, ws, B, S, heads, head_dim, (hipStream_t)stream);
}

size_t mmsa_grad_norm_ws_bytes(void) { return grad_norm_ws_bytes(); }
int mmsa_grad_norm(const float* g, int64_t n, float grad_scale, float max_norm, float* norm_out, void* ws, void* stream) {
  if (!g || !norm_out || !ws || n <= 0) return MMSA_ERR_ARG;
  return grad_norm(g, n, grad_scale, max_norm, norm_out, ws, (hipStream_t)stream);
}
int mmsa_adamw_step(float* w, const float* g, float* m, float* v, void* w16, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int32_t step, const float* norm_clip, float grad_scale, void* stream) {
  if (!w || !g || !m || !v) return MMSA_ERR_ARG;
  return adamw_step(w, g, m, v, w16, n, lr, beta1, beta2, eps, weight_decay, step, norm_clip, grad_scale, (hipStream_t)stream);
}
int mmsa_grad_norm_guard(const float* g, int64_t n, float grad_scale, float max_norm, const float* loss, int32_t* step_count,
                         float* norm_out, void* ws, float beta1, float beta2, void* stream) {
  if (!g || !norm_out || !ws || n <= 0) return MMSA_ERR_ARG;
  return grad_norm(g, n, grad_scale, max_norm, norm_out, ws, (hipStream_t)stream, loss, step_count, beta1, beta2);
}
int mmsa_grad_norm_ranges(const float* g, const int64_t* offsets, const int64_t* lengths, int32_t nranges, float grad_scale,
                          float max_norm, const float* loss, int32_t* step_count, float* norm_out, void* ws, float beta1,
                          float beta2, void* stream) {
  if (!g || !offsets || !lengths || !norm_out || !ws) return MMSA_ERR_ARG;
  static_assert(sizeof(long) == sizeof(int64_t), "LP64");
  return grad_norm_ranges(g, (const long*)offsets, (const long*)lengths, nranges, grad_scale, max_norm, norm_out, ws,
                          (hipStream_t)stream, loss, step_count, beta1, beta2);
}
int mmsa_grad_sumsq_ranges(const float* g, const int64_t* offsets, const int64_t* lengths, int32_t nranges, double* sumsq, void* ws,
                           void* stream) {
  if (!g || !sumsq || !ws || nranges < 0 || (nranges > 0 && (!offsets || !lengths))) return MMSA_ERR_ARG;
  return grad_sumsq_ranges(g, (const long*)offsets, (const long*)lengths, nranges, sumsq, ws, (hipStream_t)stream);
}
int mmsa_grad_norm_from_sumsq(const double* sumsq, int32_t n, float grad_scale, float max_norm, const float* loss,
                              int32_t* step_count, float* norm_out, float beta1, float beta2, void* stream) {
  if (!sumsq || !norm_out) return MMSA_ERR_ARG;
  return grad_norm_from_sumsq(sumsq, n, grad_scale, max_norm, norm_out, (hipStream_t)stream, loss, step_count, beta1, beta2);
}
int mmsa_grad_scale_clip(float* g, int64_t n, const float* norm_clip, void* stream) {
  if (!g || !norm_clip) return MMSA_ERR_ARG;
  return grad_scale_clip(g, n, norm_clip, (hipStream_t)stream);
}
int mmsa_adamw_step_dev(float* w, const float* g, float* m, float* v, void* w16, int64_t n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, const int32_t* step_count, const float* norm_clip, float grad_scale,
                        void* stream) {
  // with a device step count the kernel reads the bias corrections from norm_clip[2..3] (written by the norm's finalize)
  if (!w || !g || !m || !v || !step_count || !norm_clip) return MMSA_ERR_ARG;
  return adamw_step(w, g, m, v, w16, n, lr, beta1, beta2, eps, weight_decay, 0, norm_clip, grad_scale, (hipStream_t)stream,
                    step_count);
}
int mmsa_cast_f32(int32_t dtype, const float* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst) return MMSA_ERR_ARG;
  return cast_f32(dtype, src, dst, n, (hipStream_t)stream);
}

int mmsa_widen_bf16(const void* src, float* dst, int64_t n, void* stream) {
  if (!src || !dst) return MMSA_ERR_ARG;
  return widen_bf16(src, dst, n, (hipStream_t)stream);
}

int mmsa_mfma_clock_probe(void* ws, int32_t blocks, int32_t iters, int32_t launches, void* stream) {
  return mfma_clock_probe(ws, blocks, iters, launches, (hipStream_t)stream);
}
int mmsa_prof_begin(int32_t max_records) { return gemm_prof_begin(max_records); }
int mmsa_prof_sample(int32_t stride, int32_t phase) { return gemm_prof_sample(stride, phase); }
int mmsa_prof_mode(int32_t mode) { return gemm_prof_mode(mode); }
int mmsa_prof_end(double* total_ms, double* total_flop, int64_t* launches) {
  if (!total_ms || !total_flop || !launches) return MMSA_ERR_ARG;
  long n = 0;
  const int rc = gemm_prof_end(total_ms, total_flop, &n);
  *launches = n;
  return rc;
}
double mmsa_prof_last_bytes(void) { return gemm_prof_last_bytes(); }

}  // extern "C"
