// Helpers shared by the encoder / head engines: Linear forward, data gradient, weight gradient and bias gradient
// expressed on the GEMM descriptor, a bump allocator for caller-provided workspaces, and parameter tables.
#pragma once
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "ops.h"

#define RET_IF(x)            \
  do {                       \
    const int _rc = (x);     \
    if (_rc) return _rc;     \
  } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Bump {  // carves a caller-provided buffer; with base == nullptr it only measures
  char* base;
  size_t off;
  explicit Bump(void* b) : base((char*)b), off(0) {}
  void* take(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  }
};

struct ParamEntry {
  std::string name;
  long offset;  // elements into the flat fp32 parameter (and gradient, and working-copy) buffer
  int ndim;
  long shape[4];
  long numel() const {
    long n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    return n;
  }
};

struct ParamTable {
  std::vector<ParamEntry> entries;
  long total = 0;
  long add(const std::string& name, std::initializer_list<long> shape) {
    ParamEntry e;
    e.name = name;
    e.offset = total;
    e.ndim = (int)shape.size();
    int i = 0;
    for (long s : shape) e.shape[i++] = s;
    for (; i < 4; ++i) e.shape[i] = 1;
    entries.push_back(e);
    total += (long)align_up((size_t)e.numel(), 64);  // 64-element alignment: 16-byte aligned in bf16 and fp32
    return e.offset;
  }
};

struct Eng {
  int dtype;  // MMSA_F32 / MMSA_BF16 storage
  hipStream_t st;
  float* splitk_ws;
  size_t splitk_bytes;
  float* col_ws;  // colsum partials
  size_t col_ws_bytes = 0;      // capacity of col_ws (0: unknown, bias problems are not grouped)
  const void* ones8 = nullptr;  // optional [rows][8] bf16 ones (rows >= the M of wgrad_group): bias gradients ride in the group
  // optional: the picks of the grouped bias gradients are recorded here instead of launched (the caller then owns col_ws until
  // it flushes them with bias_pick_batch: one launch for a whole backward instead of one per layer)
  std::vector<BiasPickJob>* defer_picks = nullptr;

  size_t esz() const { return dtype == MMSA_BF16 ? 2 : 4; }
  // MMSA_BF16_SIMT=1 (diagnostic): run the bf16 path on the SIMT kernels — same storage rounding, independent code —
  // so the MFMA engines can be cross-checked end to end on the device.
  static bool force_simt() {
    const char* v = getenv("MMSA_BF16_SIMT");  // read per call so a test can toggle it inside one process
    return v && atoi(v) != 0;
  }
  static bool v1_only() {
    const char* v = getenv("MMSA_GEMM_V1");
    return v && atoi(v) != 0;
  }
  int gemm(const GemmParams& pin) const {
    if (dtype != MMSA_BF16) {
      GemmParams p = pin;
      if (splitk_ws && !(p.N % 4) && gemm_f32_mfma_eligible(p)) {  // the fp32-MFMA kernel plans its own K split
        if (p.split_k < 1) p.split_k = 1;
        p.ws = splitk_ws;
        p.ws_bytes = (long)splitk_bytes;
      }
      return gemm_f32_launch(p, st);
    }
    if (force_simt()) return gemm_bf16_simt_launch(pin, st);
    // Few-tile, deep-K launches (stage-3/4 convolutions: 100-200 tiles of 128x128 for 256 CUs x 2 slots) get a split
    // over K so that the machine is filled; the slab reducer applies the epilogue.
    GemmParams p = pin;
    if (splitk_ws && !(p.N % 4) && gemm2_eligible(p) && !v1_only()) {  // the persistent kernel plans its own K split
      if (p.split_k < 1) p.split_k = 1;
      p.ws = splitk_ws;
      p.ws_bytes = (long)splitk_bytes;
      return gemm_bf16_launch(p, st);
    }
    if (p.split_k <= 1 && splitk_ws && !(p.N % 4)) {
      const long tiles = (long)cdiv(p.M, 128) * cdiv(p.N, 128);
      if (tiles <= 200 && p.K >= 1024) {
        int split = (int)(512 / tiles);
        if (split > p.K / 512) split = p.K / 512;
        if (split > 8) split = 8;
        while (split > 1 && (size_t)split * p.M * p.N * sizeof(float) > splitk_bytes) --split;
        if (split > 1) { p.split_k = split; p.ws = splitk_ws; }
      }
    }
    return gemm_bf16_launch(p, st);
  }
  int attn_impl() const { return dtype != MMSA_BF16 ? 0 : (force_simt() ? 2 : 1); }

  static GemmParams blank() {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.split_k = 1;
    return p;
  }

  // y[M,N] = act(x[M,K] W[N,K]^T + bias) (+ add)
  int linear_fwd(const void* x, long ldx, const void* W, const float* bias, void* y, long ldy, int M, int N, int K,
                 int act = MMSA_ACT_NONE, void* pre = nullptr, const void* add = nullptr, long ldadd = 0,
                 int out_f32 = 0, int pre_is_gelu_grad = 0) const {
    GemmParams p = blank();
    p.A = x; p.lda = ldx; p.B = W; p.ldb = K; p.C = y; p.ldc = ldy;
    p.M = M; p.N = N; p.K = K;
    p.bias = bias; p.act = act; p.C2 = pre; p.ldc2 = N; p.add = add; p.ldadd = ldadd; p.out_f32 = out_f32;
    p.c2_gelu_grad = (pre && act == MMSA_ACT_GELU) ? pre_is_gelu_grad : 0;
    small_split(p);
    return gemm(p);
  }
  // The same Linear with fp8 (e4m3) operands: the input x [M][K] (bf16 rows, stride ldx) is quantized per ROW (one scale per
  // token: fp8_quantize_rows, one pass) into q_act / q_row_scales; the weight comes already quantized (qW: e4m3 [N][K], per-tensor
  // scale *w_scale — fp8_quantize_batch, once per forward for all Linears); then one fp8 MFMA GEMM with the usual epilogue.
  // MMSA_ERR_UNSUPPORTED: caller falls back to bf16.
  int linear_fwd_fp8(const void* x, long ldx, const void* qW, const float* w_scale, const float* bias, void* y, long ldy, int M,
                     int N, int K, int act, void* pre, const void* add, long ldadd, int pre_is_gelu_grad, void* q_act,
                     float* q_row_scales, bool prequantized = false) const {
    if (dtype != MMSA_BF16 || force_simt() || !qW || !w_scale || !q_act || !q_row_scales || K > 4096 || (ldx % 8)) return MMSA_ERR_UNSUPPORTED;
    GemmParams p = blank();
    p.A = q_act; p.lda = K; p.B = qW; p.ldb = K; p.C = y; p.ldc = ldy;
    p.M = M; p.N = N; p.K = K;
    p.bias = bias; p.act = act; p.C2 = pre; p.ldc2 = N; p.add = add; p.ldadd = ldadd;
    p.c2_gelu_grad = (pre && act == MMSA_ACT_GELU) ? pre_is_gelu_grad : 0;
    p.scale_a_rows = 1;
    if (!gemm_fp8_eligible(p)) return MMSA_ERR_UNSUPPORTED;
    // (prequantized: the LayerNorm that produced x already wrote q_act / q_row_scales — layernorm_fwd's q_out)
    if (!prequantized) RET_IF(fp8_quantize_rows(x, ldx, M, K, q_act, q_row_scales, st));
    return gemm_fp8_launch(p, q_row_scales, w_scale, st);
  }
  // fp32 SIMT GEMMs of the fusion head have M = batch rows: a handful of workgroups each walking the whole K serially.
  // Split K over workgroups (the reducer applies the epilogue) so the launch is a few microseconds instead of ~25.
  void small_split(GemmParams& p) const {
    if (dtype != MMSA_F32 || (p.N % 4) || p.K < 256) return;
    const long tiles = (long)cdiv(p.M, 64) * cdiv(p.N, 64);
    if (tiles >= 64) return;
    int split = p.K / 64;
    if (split > 16) split = 16;
    while (split > 1 && (size_t)split * p.M * p.N * sizeof(float) > splitk_bytes) --split;
    if (split > 1) { p.split_k = split; p.ws = splitk_ws; }
  }
  // dx[M,K] = dy[M,N] W[N,K]  (* gelu'(mul)) (+ add)
  int linear_dgrad(const void* dy, long lddy, const void* W, void* dx, long lddx, int M, int N, int K,
                   const void* mul = nullptr, long ldmul = 0, const void* add = nullptr, long ldadd = 0,
                   int mul_is_factor = 0) const {
    GemmParams p = blank();
    p.A = dy; p.lda = lddy; p.B = W; p.ldb = K; p.b_kmajor = 1; p.C = dx; p.ldc = lddx;
    p.M = M; p.N = K; p.K = N;
    p.mul = mul; p.ldmul = ldmul; p.add = add; p.ldadd = ldadd; p.mul_is_factor = mul ? mul_is_factor : 0;
    small_split(p);
    return gemm(p);
  }
  int pick_split(int Mo, int No, int Kred) const {
    const int tile = dtype == MMSA_BF16 ? 128 : 64;
    const long tiles = (long)cdiv(Mo, tile) * cdiv(No, tile);
    int split = (int)(512 / (tiles > 0 ? tiles : 1));
    const int maxs = Kred / 256;
    if (split > maxs) split = maxs;
    if (split > 256) split = 256;
    if (split < 1) split = 1;
    while (split > 1 && (size_t)split * Mo * No * sizeof(float) > splitk_bytes) --split;
    return split;
  }
  // dW[N,K] (+)= dy[M,N]^T x[M,K]   (fp32 gradient)
  int linear_wgrad(const void* dy, long lddy, const void* x, long ldx, float* dW, int M, int N, int K, int accumulate) const {
    GemmParams p = blank();
    p.A = dy; p.lda = lddy; p.a_kmajor = 1; p.B = x; p.ldb = ldx; p.b_kmajor = 1; p.C = dW; p.ldc = K;
    p.M = N; p.N = K; p.K = M;
    p.out_f32 = 1; p.accumulate = accumulate;
    p.split_k = pick_split(N, K, M);
    p.ws = splitk_ws;
    return gemm(p);
  }
  // Up to 4 weight gradients dW_g[N_g, K_g] = dy_g[M, N_g]^T x_g[M, K_g] (+ bias gradients db_g = column sums of dy_g) of
  // one layer as ONE launch when the MFMA path can group them; otherwise one by one (+ separate column sums).
  struct WgradJob { const void* dy; long lddy; const void* x; long ldx; float* dW; float* db; int N, K; };
  int wgrad_group(const WgradJob* jobs, int n, int M, int accumulate) const {
    if (dtype == MMSA_BF16 && !force_simt() && !accumulate && n >= 2 && n <= 4) {
      GemmParams ps[6];
      float* cs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
      // Bias gradients db_g[n] = sum_m dy_g[m][n] as extra problems of the same launch: dy_g^T x ones[M][8] -> [N_g][8]
      // fp32 in col_ws (every column holds db_g), picked up by one small kernel. The group's weight-gradient tiles do
      // not fill the chip (BERT-base layer: 216 tiles of 256x128 on 256 CUs), so the 9-12 extra tiles of a bias problem
      // run on CUs that were idle anyway — instead of a column-sum kernel + its finalize per bias (4 launches a layer).
      static const bool no_bias_group = mmsa_disabled("bias_group");
      int nb = 0, bias_of[2] = {-1, -1};
      long scratch_off[2] = {0, 0};
      if (ones8 && !no_bias_group) {
        long off = 0;
        for (int g = 0; g < n && nb < 2; ++g)
          if (jobs[g].db && (size_t)(off + (long)jobs[g].N * 8) * sizeof(float) <= col_ws_bytes) {
            bias_of[nb] = g; scratch_off[nb] = off; off += (long)jobs[g].N * 8; ++nb;
          }
      }
      for (int g = 0; g < n; ++g) {
        GemmParams p = blank();
        p.A = jobs[g].dy; p.lda = jobs[g].lddy; p.a_kmajor = 1; p.B = jobs[g].x; p.ldb = jobs[g].ldx; p.b_kmajor = 1;
        p.C = jobs[g].dW; p.ldc = jobs[g].K; p.M = jobs[g].N; p.N = jobs[g].K; p.K = M; p.out_f32 = 1;
        ps[g] = p;
        cs[g] = nullptr;  // (the kernel can fuse the column sums — G2_GROUP_COLSUM — but the extra accumulators make the
                          //  256x128 variant spill inside its main loop; the bias gradients stay separate launches)
      }
      for (int b = 0; b < nb; ++b) {
        const WgradJob& j = jobs[bias_of[b]];
        GemmParams p = blank();
        p.A = j.dy; p.lda = j.lddy; p.a_kmajor = 1; p.B = ones8; p.ldb = 8; p.b_kmajor = 1;
        p.C = col_ws + scratch_off[b]; p.ldc = 8; p.M = j.N; p.N = 8; p.K = M; p.out_f32 = 1;
        ps[n + b] = p;
      }
      int rc = gemm_bf16_launch_group(ps, cs, n + nb, st);
      bool biases_in_group = rc != MMSA_ERR_UNSUPPORTED && nb > 0;
      if (rc == MMSA_ERR_UNSUPPORTED && nb > 0) rc = gemm_bf16_launch_group(ps, cs, n, st);  // without the bias problems
      if (rc != MMSA_ERR_UNSUPPORTED) {
        if (rc) return rc;
        if (biases_in_group && defer_picks) {
          for (int b = 0; b < nb; ++b) defer_picks->push_back(BiasPickJob{col_ws + scratch_off[b], jobs[bias_of[b]].db, jobs[bias_of[b]].N});
        } else if (biases_in_group)
          RET_IF(bias_pick(col_ws + scratch_off[0], jobs[bias_of[0]].db, jobs[bias_of[0]].N,
                           nb > 1 ? col_ws + scratch_off[1] : nullptr, nb > 1 ? jobs[bias_of[1]].db : nullptr,
                           nb > 1 ? jobs[bias_of[1]].N : 0, st));
        for (int g = 0; g < n; ++g) {
          const bool done = biases_in_group && (g == bias_of[0] || g == bias_of[1]);
          if (jobs[g].db && !done) RET_IF(bias_grad(jobs[g].dy, jobs[g].lddy, jobs[g].db, M, jobs[g].N, accumulate));
        }
        return MMSA_OK;
      }
    }
    for (int g = 0; g < n; ++g) {
      if (jobs[g].db) RET_IF(bias_grad(jobs[g].dy, jobs[g].lddy, jobs[g].db, M, jobs[g].N, accumulate));
      RET_IF(linear_wgrad(jobs[g].dy, jobs[g].lddy, jobs[g].x, jobs[g].ldx, jobs[g].dW, M, jobs[g].N, jobs[g].K, accumulate));
    }
    return MMSA_OK;
  }
  // Deferred weight gradients of several layers (each built as for gemm(): both operands k-major, fp32 output): plain problems
  // with the same K, and convolution problems with one geometry, go out as grouped launches with ONE common K split
  // (gemm2_launch_group: a ResNet stage's 5-11 same-size weight gradients fill the chip with 2-8 K slices instead of 16-60
  // each); whatever does not group is launched on its own.
  int wgrad_batch(const GemmParams* jobs, int n) const {
    std::vector<char> done((size_t)n, 0);
    static const bool off = mmsa_disabled("wgrad_group");
    const bool can_group = dtype == MMSA_BF16 && !force_simt() && !v1_only() && splitk_ws && !off;
    for (int i = 0; i < n; ++i) {
      if (done[i]) continue;
      int idx[GEMM_MAX_GROUPS], m = 0;
      if (can_group && jobs[i].a_kmajor && jobs[i].b_kmajor && jobs[i].out_f32) {
        const GemmParams& a = jobs[i];
        for (int j = i; j < n && m < GEMM_MAX_GROUPS; ++j) {
          if (done[j]) continue;
          const GemmParams& b = jobs[j];
          bool same = b.a_kmajor && b.b_kmajor && b.out_f32 && a.K == b.K && a.gather == b.gather && a.accumulate == b.accumulate;
          if (same && a.gather)
            same = a.M == b.M && a.N == b.N && a.lda == b.lda && a.ldb == b.ldb && memcmp(&a.g, &b.g, sizeof(a.g)) == 0;
          if (same) idx[m++] = j;
        }
      }
      if (m >= 2) {
        GemmParams ps[GEMM_MAX_GROUPS];
        for (int k = 0; k < m; ++k) { ps[k] = jobs[idx[k]]; ps[k].split_k = 1; }
        ps[0].ws = splitk_ws;
        ps[0].ws_bytes = (long)splitk_bytes;
        const int rc = gemm_bf16_launch_group(ps, nullptr, m, st);
        if (rc != MMSA_ERR_UNSUPPORTED) {
          if (rc) return rc;
          for (int k = 0; k < m; ++k) done[idx[k]] = 1;
          continue;
        }
      }
      RET_IF(gemm(jobs[i]));
      done[i] = 1;
    }
    return MMSA_OK;
  }
  int bias_grad(const void* dy, long lddy, float* db, int M, int N, int accumulate) const {
    return colsum(dtype, dy, lddy, db, accumulate, col_ws, M, N, st);
  }
  // Backward of one Linear y = x W^T + b:  dW[N,K] (+)= dy^T x,  db[N] (+)= column sums of dy,  dx[M,K] = dy W (* gelu'(mul)).
  // Any of dW / db / dx may be null. The fusion head's batch-row problems (fp32) go out as ONE launch of the batch-row
  // kernel (the bias gradient as dy^T times a ones column, gemm.h b_ones); everything else as the separate calls.
  int linear_bwd(const void* dy, long lddy, const void* x, long ldx, const void* W, float* dW, float* db, void* dx, long lddx,
                 int M, int N, int K, int accumulate, const void* mul = nullptr, long ldmul = 0) const {
    if (dtype == MMSA_F32) {
      GemmParams ps[3];
      int n = 0;
      if (dW) {
        GemmParams& p = ps[n++];
        p = blank();
        p.A = dy; p.lda = lddy; p.a_kmajor = 1; p.B = x; p.ldb = ldx; p.b_kmajor = 1; p.C = dW; p.ldc = K;
        p.M = N; p.N = K; p.K = M; p.out_f32 = 1; p.accumulate = accumulate;
      }
      if (db) {
        GemmParams& p = ps[n++];
        p = blank();
        p.A = dy; p.lda = lddy; p.a_kmajor = 1; p.B = dy; p.ldb = 1; p.b_kmajor = 1; p.b_ones = 1; p.C = db; p.ldc = 1;
        p.M = N; p.N = 1; p.K = M; p.out_f32 = 1; p.accumulate = accumulate;
      }
      if (dx) {
        GemmParams& p = ps[n++];
        p = blank();
        p.A = dy; p.lda = lddy; p.B = W; p.ldb = K; p.b_kmajor = 1; p.C = dx; p.ldc = lddx;
        p.M = M; p.N = K; p.K = N; p.mul = mul; p.ldmul = ldmul;
      }
      if (n >= 2) {
        const int rc = gemm_f32_tiny_launch_multi(ps, n, st);
        if (rc != MMSA_ERR_UNSUPPORTED) return rc;
      }
    }
    if (db) RET_IF(bias_grad(dy, lddy, db, M, N, accumulate));
    if (dW) RET_IF(linear_wgrad(dy, lddy, x, ldx, dW, M, N, K, accumulate));
    if (dx) RET_IF(linear_dgrad(dy, lddy, W, dx, lddx, M, N, K, mul, ldmul));
    return MMSA_OK;
  }
};
