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
 *  - every call only enqueues work on `stream`: no allocation, no synchronization, no host copies. A call that takes ONE stream
 *    is hipGraph-capturable (tools/microbench/graph_probe.py replays the whole single-stream training step from a graph);
 *    mmsa_resnet_bwd_cb2 with a second stream forks and joins it with events inside the call, and the Python step that drives
 *    the encoders on three streams is NOT capturable (FusedTrainStep raises MmsaError when a capture is attempted with its side
 *    streams on). Scratch memory is caller-provided (`ws`), sized by the matching *_ws_bytes().
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
#define MMSA_FP8 2 /* encoder cfg.dtype only: bf16 storage, fp8 (OCP e4m3) operands in the text encoder's forward Linears */

/* GEMM implementation selector */
#define MMSA_GEMM_F32_SIMT 0  /* fp32 storage: the exact-fp32 mode as the engines run it — the fp32-MFMA kernel when the problem is
                                 large enough for its 128x128 tile, else the VALU-fma kernel (fusion head, 3-class heads) */
#define MMSA_GEMM_BF16_MFMA 1 /* bf16 storage, v_mfma_f32_16x16x32_bf16 */
#define MMSA_GEMM_BF16_SIMT 2 /* bf16 storage, VALU fma (on-device cross-check of the MFMA kernel) */
#define MMSA_GEMM_F32_MFMA 3  /* fp32 storage, v_mfma_f32_32x32x2_f32 forced (any size): bitwise equal to impl 4 */
#define MMSA_GEMM_F32_VALU 4  /* fp32 storage, VALU fma forced (the checker of impl 3) */

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
/* n (2..12) independent weight-gradient GEMMs with the same K as ONE launch of the bf16 MFMA kernel: every descriptor must
 * have a_kmajor = b_kmajor = 1, out_f32 = 1, no epilogue operands, no split. This is how the backward of a BERT layer
 * issues its four dW = dY^T X products (the `.backward()` of the reference's train step, Trainer.py:79, reaches them
 * through autograd). Returns MMSA_ERR_UNSUPPORTED (3) when the problems cannot be grouped; nothing is launched then. */
int mmsa_gemm_group(const mmsa_gemm_desc* d, int32_t n, void* stream);
/* The same with a K split common to the group, planned by the library inside `ws` (device scratch for the fp32 slabs; the split
 * is the largest useful one that fits ws_bytes): the tiles of all problems store slabs and ONE reducer launch sums them into the
 * outputs (accumulate = 1 in every descriptor: C += instead of C =). Besides plain problems the group may be the weight
 * gradients of convolutions with ONE geometry (gather = 2 and identical M, N, lda, ldb, geom in every descriptor; only the
 * pointers differ). This is how the ResNet backward issues the 5-11 same-size 1x1 and the 2-5 same-geometry 3x3 weight gradients
 * of a stage (autograd reaches them from Trainer.py:79's `.backward()`): launched one by one each needed 16-60 K slices of 4-13
 * K steps to fill 256 CUs. Returns MMSA_ERR_UNSUPPORTED (3) when the problems cannot be grouped; nothing is launched then. */
int mmsa_gemm_group_split(const mmsa_gemm_desc* d, int32_t n, float* ws, size_t ws_bytes, void* stream);
/* fp8 (OCP e4m3, the gfx950 format) path of BASELINE.json configs[4]: mmsa_fp8_quantize turns a contiguous bf16 tensor (n % 8 == 0)
 * into e4m3 bytes with a per-tensor scale = amax / 448 (device float; amax_ws: mmsa_fp8_quantize_ws_bytes() of device scratch); mmsa_gemm_fp8 is the
 * NT GEMM of mmsa_gemm on such operands (desc->A / B: e4m3 bytes, k-contiguous rows, lda / ldb in bytes; K % 128 == 0, no
 * gather / split): C = epilogue(scale_a * scale_b * A B^T), fp32 accumulation in v_mfma_f32_16x16x32_fp8_fp8, bf16 (or fp32)
 * output. Returns 3 (unsupported) for shapes it does not take. */
size_t mmsa_fp8_quantize_ws_bytes(void);
int mmsa_fp8_quantize(const void* x_bf16, int64_t n, void* out_e4m3, float* scale, void* amax_ws, void* stream);
int mmsa_gemm_fp8(const mmsa_gemm_desc* d, const float* scale_a, const float* scale_b, void* stream);
/* The forms the text encoder's forward uses (mmsa_bert_fwd with dtype MMSA_FP8):
 * - mmsa_fp8_quantize_rows: per-TOKEN scales in ONE pass — row m of x [M][K] (bf16, row stride ldx elements, K % 8 == 0,
 *   K <= 4096) becomes e4m3 bytes out[m][0..K) with row_scales[m] = max|x[m]| / 448 (2 B read + 1 B written per element, against
 *   5 B and two launches for a per-tensor scale, which must see the whole tensor before it can convert);
 * - mmsa_gemm_fp8_rows: mmsa_gemm_fp8 whose A operand carries those per-row scales (C[m][n] = epilogue(row_scales_a[m] *
 *   scale_b[0] * sum_k ...));
 * - mmsa_fp8_quantize_batch: per-tensor quantization of n <= 128 tensors of one bf16 buffer in TWO launches (the weights of every
 *   quantized Linear, once per forward): tensor t = numel[t] elements at element offset offsets[t] (both % 8 == 0); its e4m3
 *   bytes go to the same offset of out_e4m3, its scale to scales[t]; ws: mmsa_fp8_quantize_batch_ws_bytes(n) of device scratch. */
int mmsa_fp8_quantize_rows(const void* x_bf16, int64_t ldx, int32_t M, int32_t K, void* out_e4m3, float* row_scales, void* stream);
int mmsa_gemm_fp8_rows(const mmsa_gemm_desc* d, const float* row_scales_a, const float* scale_b, void* stream);
size_t mmsa_fp8_quantize_batch_ws_bytes(int32_t n);
int mmsa_fp8_quantize_batch(const void* base_bf16, const int64_t* offsets, const int64_t* numel, int32_t n, void* out_e4m3,
                            float* scales, float* ws, void* stream);

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
 * impl as MMSA_GEMM_* (0 fp32 SIMT, 1 bf16 MFMA, 2 bf16 SIMT; attention has no 3 / 4). qkv [B*S][3*heads*64], ctx [B*S][heads*64],
 * mask [B][S] fp32 (1 keep / 0 masked) or NULL. Backward writes d_qkv packed like qkv. */
size_t mmsa_attention_bwd_ws_bytes(int32_t B, int32_t S, int32_t heads);
int mmsa_attention_fwd(int32_t impl, const void* qkv, const float* mask, void* ctx, int32_t B, int32_t S, int32_t heads,
                       int32_t head_dim, void* stream);
int mmsa_attention_bwd(int32_t impl, const void* qkv, const float* mask, const void* dctx, void* dqkv, float* ws,
                       int32_t B, int32_t S, int32_t heads, int32_t head_dim, void* stream);

/* ---- BERT text encoder engine (fills the encoder slot MultimodalModel.py:264-266; NOT in the reference) ------
 * Parameters live in ONE flat fp32 buffer (`w32`) whose layout is reported by mmsa_bert_param_info (HF BertModel
 * names under "bert.", plus "proj.*" = Linear(hidden, out_dim) into the fusion width). `wt` is the working copy in
 * the storage dtype with the same element offsets (== w32 for fp32). Gradients go to a flat fp32 buffer with
 * the same offsets (+= when accumulate, = otherwise). `ws` (mmsa_bert_ws_bytes) holds the activations saved by
 * the forward for the backward, which must see the same ws, ids and mask. */
typedef struct mmsa_bert_cfg {
  int32_t batch, seq, hidden, layers, heads, intermediate, vocab, max_pos, type_vocab, out_dim, dtype;
  float ln_eps;
} mmsa_bert_cfg;
int mmsa_bert_param_count(const mmsa_bert_cfg* c);
int64_t mmsa_bert_param_total(const mmsa_bert_cfg* c);
int mmsa_bert_param_info(const mmsa_bert_cfg* c, int idx, char* name, int name_cap, int64_t* offset, int32_t* ndim,
                         int64_t* shape /*[4]*/);
size_t mmsa_bert_ws_bytes(const mmsa_bert_cfg* c);
int mmsa_bert_fwd(const mmsa_bert_cfg* c, const float* w32, const void* wt, const int64_t* ids, const float* mask, void* ws,
                  float* feat /*[batch][out_dim] fp32*/, void* stream);
int mmsa_bert_bwd(const mmsa_bert_cfg* c, const float* w32, const void* wt, const int64_t* ids, const float* mask, void* ws,
                  const float* dfeat, float* grad, int32_t accumulate, void* stream);

/* ---- ResNet (bottleneck v1.5) image encoder engine (encoder slot; NOT in the reference) -------------------------
 * image is NCHW fp32 [batch][3][height][width] as the reference-style loaders deliver it; everything inside is NHWC
 * in the storage dtype. Convolution weights are physically [Cout][KH][KW][Cin] (channels_last of the logical
 * torchvision shape). `bnbuf` = flat fp32 running_mean / running_var buffers (param_info with buffers = 1),
 * updated in training mode. `training`: 1 = batch statistics (running buffers updated); 0 = eval mode, every tensor a backward
 * needs is kept; 2 = inference (eval mode, forward only): each BatchNorm is folded into the convolution that feeds it — the
 * scale gamma / sqrt(running_var + eps) goes into a scratch copy of the weights, the shift into the GEMM's bias, the ReLU and the
 * residual add into its epilogue: no BatchNorm kernel, the pre-normalisation tensor is never stored (mmsa_resnet_bwd must not
 * follow). */
typedef struct mmsa_resnet_cfg {
  int32_t batch, height, width;
  int32_t blocks[4], widths[4];
  int32_t out_dim, dtype, training;
  float bn_eps, bn_momentum;
} mmsa_resnet_cfg;
int mmsa_resnet_param_count(const mmsa_resnet_cfg* c, int32_t buffers);
int64_t mmsa_resnet_param_total(const mmsa_resnet_cfg* c, int32_t buffers);
int mmsa_resnet_param_info(const mmsa_resnet_cfg* c, int32_t buffers, int idx, char* name, int name_cap, int64_t* offset,
                           int32_t* ndim, int64_t* shape /*[4]*/);
size_t mmsa_resnet_ws_bytes(const mmsa_resnet_cfg* c);
/* Test / diagnostic aid: byte offset inside the ResNet workspace of a tensor the forward saved for the backward (NHWC,
 * storage dtype). block = -1: stem (which 0 = conv output z, 1 = BN+ReLU output y, 2 = max-pooled); block >= 0 (bottleneck
 * index in network order): which 0..7 = c1.z, c1.y, c2.z, c2.y, c3.z, c3.y (= block output), ds.z, ds.y; block = -2: which
 * 0..3 = the gradient ping-pong buffers. -1 when absent. Lets a test evaluate the oracle's backward AT the forward values
 * the device actually stored (tests/test_engines_gpu.py::test_resnet_backward_teacher_forced). */
int64_t mmsa_resnet_ws_offset(const mmsa_resnet_cfg* cfg, int32_t block, int32_t which);
int mmsa_resnet_fwd(const mmsa_resnet_cfg* c, const float* w32, const void* wt, float* bnbuf, const float* image, void* ws,
                    float* feat /*[batch][out_dim] fp32*/, void* stream);
int mmsa_resnet_bwd(const mmsa_resnet_cfg* c, const float* w32, const void* wt, void* ws, const float* dfeat, float* grad,
                    int32_t accumulate, void* stream);
/* Backward with a "gradient range ready" callback, for the data-parallel trainer (all-reduce overlapped
 * with the backward): cb(user, offset, length) is called on the host, from inside the call, each time the kernels producing
 * grad[offset, offset + length) have been enqueued on `stream` (ranges arrive from the end of the flat buffer to its start
 * and tile it exactly). BERT: pooler + projection, then layers_per_chunk encoder layers at a time, then the embeddings.
 * ResNet: projection, stage 4, 3, 2, stage 1 + stem. cb == NULL: identical to the plain entry points. */
typedef void (*mmsa_range_cb)(void* user, int64_t offset, int64_t length);
int mmsa_bert_bwd_cb(const mmsa_bert_cfg* c, const float* w32, const void* wt, const int64_t* ids, const float* mask, void* ws,
                     const float* dfeat, float* grad, int32_t accumulate, void* stream, mmsa_range_cb cb, void* user,
                     int32_t layers_per_chunk, const uint8_t* frozen);
int mmsa_resnet_bwd_cb(const mmsa_resnet_cfg* c, const float* w32, const void* wt, void* ws, const float* dfeat, float* grad,
                       int32_t accumulate, void* stream, mmsa_range_cb cb, void* user, const uint8_t* frozen);
/* The same with an optional second stream of the same device for the weight gradients: the stage-wise grouped weight-gradient
 * launches (mmsa_gemm_group_split's kernel) are enqueued on wgrad_stream, ordered after their stage's backward by an event and
 * joined into `stream` by an event before the call returns, so these throughput-bound launches overlap the latency-bound
 * BatchNorm / data-gradient chain of the following stages; the join happens on every exit path (errors included). With cb AND
 * wgrad_stream set, every announced range is complete on wgrad_stream (it waits for an event recorded on `stream` right before each
 * announcement): record the range's event there. wgrad_stream null: everything on `stream`. A backward after an inference-mode
 * forward (cfg.training == 2: nothing was stored) returns MMSA_ERR_ARG. Results are bit-identical to mmsa_resnet_bwd_cb. */
int mmsa_resnet_bwd_cb2(const mmsa_resnet_cfg* c, const float* w32, const void* wt, void* ws, const float* dfeat, float* grad,
                        int32_t accumulate, void* stream, void* wgrad_stream, mmsa_range_cb cb, void* user,
                        const uint8_t* frozen);
/* `frozen` (host array, one byte per entry of the engine's parameter table, NULL = everything trainable): SURVEY.md §8f N2 — the
 * reference's curriculum phases (dataLoader/MultiTaskTrainer.py:50-177) and fine-tuning (train.py:90-92) freeze sub-graphs. A
 * wholly frozen group (BERT: embeddings / one encoder layer / pooler + projection; ResNet: stem / one bottleneck / projection)
 * gets no weight-gradient kernels, the backward stops below the lowest trainable group, and only ranges holding trainable
 * parameters are announced through cb. Gradients of frozen entries are left undefined (not read by anything). */

/* ---- fusion-head engines (fp32) ---------------------------------------------------------------------------------
 * kind 0 CrossModalTransformer (MultimodalModel.py:108-149): inputs {query[B,E], key[B,Lk,E], value[B,Lk,E]} -> {out[B,E]}
 * kind 1 ME-MHACL fusion       (MultimodalModel.py:374-404): inputs {feat_0..feat_{tokens-1} [B,E]} -> {fused[B,E]}
 * kind 2 weighted head         (MultimodalModel.py:171-225,298-313): inputs {anchor, raw2, raw3, enh2, enh3 [B,E]}
 *                              -> {logits[B,C], fused[B,128], valence logits[B,C] if cfg.valence}
 * kind 3 Classifier            (MultimodalModel.py:432-451): {x[B,E]} -> {out_a[B,C], out_v[B,C]}
 * kind 4 ProjectionHead        (MultimodalModel.py:409-429): {x[B,E]} -> {z[B,out_dim]}
 * Parameter names reported by mmsa_head_param_info are the reference's state_dict keys. Backward takes the same
 * inputs and ws as the forward, douts in output order (a NULL dout means zero), writes dinputs in input order (a NULL
 * dinput is skipped where the module allows) and parameter gradients into `grad` (NULL = parameters frozen). */
#define MMSA_HEAD_CROSS_MODAL 0
#define MMSA_HEAD_MM_FUSION 1
#define MMSA_HEAD_WEIGHTED 2
#define MMSA_HEAD_CLASSIFIER 3
#define MMSA_HEAD_PROJECTION 4
typedef struct mmsa_head_cfg {
  int32_t batch, embed, tokens, heads, pool_mode /*0 max, 1 mean*/, num_classes, valence, hidden, out_dim, training;
  float bn_eps, bn_momentum, ln_eps, dropout_p;
  uint64_t seed; /* dropout stream for this call (the backward must see the same value) */
} mmsa_head_cfg;
int mmsa_head_param_count(int32_t kind, const mmsa_head_cfg* c, int32_t buffers);
int64_t mmsa_head_param_total(int32_t kind, const mmsa_head_cfg* c, int32_t buffers);
int mmsa_head_param_info(int32_t kind, const mmsa_head_cfg* c, int32_t buffers, int idx, char* name, int name_cap,
                         int64_t* offset, int32_t* ndim, int64_t* shape /*[4]*/);
size_t mmsa_head_ws_bytes(int32_t kind, const mmsa_head_cfg* c);
int mmsa_head_fwd(int32_t kind, const mmsa_head_cfg* c, const float* w, float* bnbuf, const float* const* inputs,
                  float* const* outputs, void* ws, void* stream);
int mmsa_head_bwd(int32_t kind, const mmsa_head_cfg* c, const float* w, const float* const* inputs, const float* const* douts,
                  float* const* dinputs, float* grad, int32_t accumulate, void* ws, void* stream);

/* ---- fused cross-entropy forward + backward entry: nn.CrossEntropyLoss() at Trainer.py:17,68; Tester.py:20,57 ----
 * loss = mean_b -log softmax(logits)[b, labels[b]]; dlogits = grad_scale * (softmax - onehot) / B (NULL to skip);
 * probs = softmax (NULL to skip; Tester.py:54). */
int mmsa_ce_fwd_bwd(const float* logits, const int64_t* labels, float* loss, float* dlogits, float* probs, int32_t B,
                    int32_t C, float grad_scale, void* stream);

/* ---- N1: the reference's contrastive losses, fused forward + backward ---------------------------------
 * mmsa_infonce_fwd_bwd: MultimodalTransformerModel.compute_contrastive_loss (MultimodalModel.py:232-260):
 *   f = x / max(||x||, 1e-12); S = f1 f2^T / T; positives = same label, diagonal excluded; S -= rowmax(S);
 *   loss = mean_i -log((sum_j e^S pos + 1e-12) / (sum_j e^S + 1e-12)).  temperature: device scalar (nn.Parameter :229).
 *   Outputs: loss (device scalar), dfeat1 / dfeat2 [B][D] = grad_scale * dloss/dfeat (distinct buffers; the reference's
 *   forward passes the same tensor twice (:272-284): the caller adds the two), dtemp = grad_scale * dloss/dT (may be NULL).
 * mmsa_supcon_fwd_bwd: contrastive_loss(z1, z2, labels, temperature=0.1) of train.py:16-40 (two views, [2B, 2B] similarity,
 *   diagonal removed from the denominator, +1e-8 terms as there).
 * ws: mmsa_contrastive_ws_bytes(B, D) bytes of device memory. fp32 features, int64 labels [B]. B <= 1024, D <= 4096. */
size_t mmsa_contrastive_ws_bytes(int32_t B, int32_t D);
int mmsa_infonce_fwd_bwd(const float* feat1, const float* feat2, const int64_t* labels, const float* temperature, float* loss,
                         float* dfeat1, float* dfeat2, float* dtemp, int32_t B, int32_t D, float grad_scale, void* ws,
                         void* stream);
int mmsa_supcon_fwd_bwd(const float* z1, const float* z2, const int64_t* labels, float temperature, float* loss, float* dz1,
                        float* dz2, int32_t B, int32_t D, float grad_scale, void* ws, void* stream);

/* ---- step tail over flat buffers: clip_grad_norm_(1.0) + AdamW (Trainer.py:19-21,80-81) ---------------------------
 * mmsa_grad_norm: norm_out[0] = grad_scale * ||g||_2, norm_out[1] = min(1, max_norm / (norm + 1e-6)) (device memory).
 * mmsa_adamw_step: decoupled-decay AdamW on w using g * norm_clip[1] * grad_scale; also refreshes the bf16 working
 * copy w16 (may be NULL). grad_scale carries the 1/world_size of the data-parallel gradient average. */
size_t mmsa_grad_norm_ws_bytes(void);
int mmsa_grad_norm(const float* g, int64_t n, float grad_scale, float max_norm, float* norm_out, void* ws, void* stream);
int mmsa_adamw_step(float* w, const float* g, float* m, float* v, void* w16, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int32_t step, const float* norm_clip, float grad_scale, void* stream);
/* NaN rule of the reference's step (Trainer.py:74-76 "NaN loss detected, skipping batch": no update): the guarded forms decide
 * on the device, without a host sync. mmsa_grad_norm (both forms) writes norm_out[1] = -1 when the gradient norm -- or, in
 * the guarded form, the device scalar *loss (may be NULL) -- is not finite; mmsa_adamw_step (both forms) returns without
 * touching w, m, v, w16 when norm_clip[1] < 0. *step_count (device int32, may be NULL in the guard) counts the APPLIED steps:
 * the guard increments it when the step is not skipped and — norm_out then has FOUR floats — writes the AdamW bias corrections
 * of that count, 1 - beta1^t and sqrt(1 - beta2^t), to norm_out[2], norm_out[3]; mmsa_adamw_step_dev reads them from norm_clip. */
int mmsa_grad_norm_guard(const float* g, int64_t n, float grad_scale, float max_norm, const float* loss, int32_t* step_count,
                         float* norm_out /*[4]*/, void* ws, float beta1, float beta2, void* stream);
/* the norm over `nranges` disjoint ranges (host arrays of element offsets / lengths, nranges <= 256) of one gradient buffer:
 * the trainable sub-ranges of a curriculum phase (dataLoader/MultiTaskTrainer.py:50-177 freezes everything else) */
int mmsa_grad_norm_ranges(const float* g, const int64_t* offsets, const int64_t* lengths, int32_t nranges, float grad_scale,
                          float max_norm, const float* loss, int32_t* step_count, float* norm_out /*[4]*/, void* ws,
                          float beta1, float beta2, void* stream);
/* The two halves of mmsa_grad_norm_ranges as separate calls (same reference lines: clip_grad_norm_, Trainer.py:80): the sum of
 * squares over `nranges` ranges (0 allowed) left as ONE device double in sumsq[0], and the finalize over n such doubles (their sum ->
 * norm, clip coefficient, skip flag, step count, bias corrections as mmsa_grad_norm_guard). Used by the reduce-scatter form of the
 * data-parallel step (every rank sums the squares of the reduced gradient shards it owns; a one-element all-reduce adds them) */
int mmsa_grad_sumsq_ranges(const float* g, const int64_t* offsets, const int64_t* lengths, int32_t nranges, double* sumsq /*[1]*/,
                           void* ws, void* stream);
int mmsa_grad_norm_from_sumsq(const double* sumsq, int32_t n, float grad_scale, float max_norm, const float* loss,
                              int32_t* step_count, float* norm_out /*[4]*/, float beta1, float beta2, void* stream);
/* g[0, n) *= norm_clip[1] (no-op on a skipped step): the in-place scaling clip_grad_norm_ applies to gradients that no
 * optimizer owns (phase 3 of dataLoader/MultiTaskTrainer.py:147-177 clips four modules and steps one) */
int mmsa_grad_scale_clip(float* g, int64_t n, const float* norm_clip, void* stream);
int mmsa_adamw_step_dev(float* w, const float* g, float* m, float* v, void* w16, int64_t n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, const int32_t* step_count, const float* norm_clip, float grad_scale,
                        void* stream);
/* storage cast fp32 -> dtype (refresh of the working weights after a foreign optimizer touched the fp32 master) */
int mmsa_cast_f32(int32_t dtype, const float* src, void* dst, int64_t n, void* stream);
/* bf16 -> fp32 (the data-parallel step's bf16 gradient payload, widened back into the fp32 gradient buffer after the all-reduce) */
int mmsa_widen_bf16(const void* src_bf16, float* dst, int64_t n, void* stream);

/* ---- BatchNorm over the rows of [M][C] (nn.BatchNorm1d: MultimodalModel.py:181,186,194,380; BatchNorm2d of the
 * ResNet encoder on NHWC), fused affine (+ residual) + activation; two-stage deterministic statistics ----------- */
size_t mmsa_bn_ws_bytes(int32_t C);
int mmsa_bn_fwd(int32_t dtype, const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                float* mean, float* invstd, const void* res, void* y, float* ws, int32_t M, int32_t C, float eps,
                float momentum, int32_t act, int32_t training, void* stream);
int mmsa_bn_bwd(int32_t dtype, const void* dy, const void* x, const void* y, const float* mean, const float* invstd,
                const float* gamma, const float* beta, void* dx, void* dres, float* dgamma, float* dbeta, int32_t accumulate,
                float* ws, int32_t M, int32_t C, int32_t act, int32_t training, void* stream);
/* ---- pooling / stem layout kernels of the ResNet encoder (NHWC) ---------------------------------------------------- */
int mmsa_maxpool_fwd(int32_t dtype, const void* x, void* y, uint8_t* idx, int32_t B, int32_t H, int32_t W, int32_t C,
                     void* stream);
int mmsa_maxpool_bwd(int32_t dtype, const void* dy, const uint8_t* idx, void* dx, int32_t B, int32_t H, int32_t W, int32_t C,
                     void* stream);
int mmsa_avgpool_fwd(int32_t dtype, const void* x, void* y, int32_t B, int32_t HW, int32_t C, void* stream);
int mmsa_avgpool_bwd(int32_t dtype, const void* dy, void* dx, int32_t B, int32_t HW, int32_t C, void* stream);
int mmsa_stem_im2col(int32_t dtype, const float* img, void* col, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t OH,
                     int32_t OW, int32_t KH, int32_t KW, int32_t stride, int32_t pad, int32_t Kpad, void* stream);

/* ---- live timing of the MFMA GEMM launches (bench.py `roofline`): when enabled, every bf16 MFMA GEMM launch (incl.
 * its split-K reduction) is bracketed by HIP events on its launch stream. mmsa_prof_end must be called after the
 * stream has been synchronized; it returns the summed device time, the algorithmic flop (2*M*N*K) and the count. */
int mmsa_prof_begin(int32_t max_records);
/* Sampling: only launches whose running index (reset by this call) satisfies index % stride == phase are bracketed.
 * An event pair drains the queue around its kernel (+5..15 us per launch when every launch is bracketed), so bench.py
 * brackets every 5th launch in each of 5 steps, rotating the phase: each launch is measured once, undisturbed. */
int mmsa_prof_sample(int32_t stride, int32_t phase);
/* mode 0 (default): HIP events around each sampled launch. mode 1: no events; the MFMA GEMM kernels (and the split-K
 * reducer) stamp {first workgroup start, last workgroup end} with s_memrealtime (100 MHz) into a device record —
 * the kernel's own duration, as rocprofv3 --kernel-trace reports it. Set before mmsa_prof_begin. */
int mmsa_prof_mode(int32_t mode);
int mmsa_prof_end(double* total_ms, double* total_flop, int64_t* launches);
/* Algorithmic HBM bytes of the launches the latest mmsa_prof_end summed: per problem every distinct operand read once (an
 * implicit-GEMM gather counts each source pixel once), the output written once, epilogue side operands / side outputs included. */
double mmsa_prof_last_bytes(void);

/* Sustained matrix-core clock of this chip (bench.py: the peak restated from CU count x sustained clock x MFMA FLOP/CU/clk,
 * SURVEY.md section 8(d); the reference has no counterpart). Runs `launches` back-to-back launches of a dense bf16 MFMA loop
 * (`iters` x 16 v_mfma_f32_16x16x32_bf16 per wave, one wave per SIMD, `blocks` workgroups) and leaves, for the LAST launch,
 * ws[2b] = shader cycles and ws[2b+1] = 100 MHz ticks of workgroup b as uint64 (ws: blocks * 1040 bytes of device memory).
 * clock [GHz] = cycles / ticks * 0.1 after a stream synchronize. */
int mmsa_mfma_clock_probe(void* ws, int32_t blocks, int32_t iters, int32_t launches, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMSA_H */
