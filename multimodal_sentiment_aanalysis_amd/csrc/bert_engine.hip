// BERT text encoder engine: one call runs the whole forward (embeddings -> L post-LN layers -> tanh pooler ->
// Linear(H, D) projection into the fusion width) and one call the whole backward, as a fixed sequence of kernel
// launches on the caller's stream over a caller-provided workspace (no allocation, no sync: hipGraph-capturable).
//
// NOT IN THE REFERENCE: BERT fills the reference's encoder slot (MultimodalModel.py:264-266); the arithmetic is the
// public BERT architecture (HF BertModel names are kept for the parameters). Storage dtype bf16 -> MFMA GEMMs and
// MFMA attention; fp32 -> SIMT kernels (exact mode).
//
// Flat parameter layout (elements; same offsets in the fp32 master, the storage-type working copy and the fp32
// gradient buffer): query/key/value weights (and biases) of a layer are adjacent so the QKV projection is ONE GEMM.
#include "../../include/mmsa.h"
#include "engine_common.h"

struct BertLayerOff {
  long wqkv, bqkv, wo, bo, ln1w, ln1b, w1, b1, w2, b2, ln2w, ln2b;
};
struct BertLayout {
  ParamTable t;
  long word, pos, type, lnw, lnb, wp, bp, wproj, bproj;
  std::vector<BertLayerOff> L;
};

static BertLayout bert_layout(const mmsa_bert_cfg& c) {
  BertLayout o;
  ParamTable& t = o.t;
  const long H = c.hidden, I = c.intermediate;
  o.word = t.add("bert.embeddings.word_embeddings.weight", {c.vocab, H});
  o.pos = t.add("bert.embeddings.position_embeddings.weight", {c.max_pos, H});
  o.type = t.add("bert.embeddings.token_type_embeddings.weight", {c.type_vocab, H});
  o.lnw = t.add("bert.embeddings.LayerNorm.weight", {H});
  o.lnb = t.add("bert.embeddings.LayerNorm.bias", {H});
  for (int l = 0; l < c.layers; ++l) {
    const std::string p = "bert.encoder.layer." + std::to_string(l) + ".";
    BertLayerOff f;
    // H*H and H are multiples of 64 for every supported config, so q|k|v stay contiguous
    f.wqkv = t.add(p + "attention.self.query.weight", {H, H});
    t.add(p + "attention.self.key.weight", {H, H});
    t.add(p + "attention.self.value.weight", {H, H});
    f.bqkv = t.add(p + "attention.self.query.bias", {H});
    t.add(p + "attention.self.key.bias", {H});
    t.add(p + "attention.self.value.bias", {H});
    f.wo = t.add(p + "attention.output.dense.weight", {H, H});
    f.bo = t.add(p + "attention.output.dense.bias", {H});
    f.ln1w = t.add(p + "attention.output.LayerNorm.weight", {H});
    f.ln1b = t.add(p + "attention.output.LayerNorm.bias", {H});
    f.w1 = t.add(p + "intermediate.dense.weight", {I, H});
    f.b1 = t.add(p + "intermediate.dense.bias", {I});
    f.w2 = t.add(p + "output.dense.weight", {H, I});
    f.b2 = t.add(p + "output.dense.bias", {H});
    f.ln2w = t.add(p + "output.LayerNorm.weight", {H});
    f.ln2b = t.add(p + "output.LayerNorm.bias", {H});
    o.L.push_back(f);
  }
  o.wp = t.add("bert.pooler.dense.weight", {H, H});
  o.bp = t.add("bert.pooler.dense.bias", {H});
  o.wproj = t.add("proj.weight", {c.out_dim, H});
  o.bproj = t.add("proj.bias", {c.out_dim});
  return o;
}

// dtype 2 (MMSA_FP8, BASELINE.json configs[4]): bf16 storage everywhere, the forward's four Linears per layer take fp8 (e4m3)
// operands (gemm_fp8.hip). sdt = the storage type every other kernel sees.
static inline int sdt(const mmsa_bert_cfg& c) { return c.dtype == MMSA_FP8 ? MMSA_BF16 : c.dtype; }

static bool bert_cfg_ok(const mmsa_bert_cfg& c) {
  return c.batch > 0 && c.seq > 0 && c.seq <= c.max_pos && c.hidden % 64 == 0 && c.heads > 0 &&
         c.hidden == c.heads * 64 && c.intermediate % 64 == 0 && c.out_dim % 64 == 0 && c.layers > 0 &&
         c.vocab > 0 && c.type_vocab > 0 && (c.dtype == MMSA_F32 || c.dtype == MMSA_BF16 || c.dtype == MMSA_FP8);
}

struct BertLayerWs {
  void *qkv, *ctx, *s1, *h1, *pre, *act, *s2, *out;
  float *mean1, *rstd1, *mean2, *rstd2;
};
struct BertWs {
  void *e, *x0;
  float *mean0, *rstd0;
  std::vector<BertLayerWs> L;
  void *pooled, *dfeat_t, *dpool, *dprepool;
  void *bufA, *bufB, *bufC, *bufD, *bufI, *bufQ;
  void* bufS;  // ds1 of the current layer (read last, by the layer's grouped weight gradients)
  void* ones8;  // [B*S][8] bf16 ones (grouped bias gradients, engine_common.h)
  void *q_act, *q_w;     // fp8 mode: e4m3 copy of a Linear's input ([B*S][max(H, I)]); e4m3 image of the whole weight table
                         // (byte i = element i of the bf16 working copy; only the quantized Linears' ranges are written)
  float *q_wscales, *q_rowscales, *q_batchws;  // per-tensor weight scales [4 * layers], per-token scales [B*S], quantizer scratch
  size_t colws_bytes;
  float *splitk, *colws, *lnws, *attnws;
  // deferred finalizes (round 4): every LayerNorm backward of the step keeps its own partial buffer and every layer its own bias
  // scratch until ONE batched finalize / pick launch per announced range (bwd: lnws_l[2 l], [2 l + 1], [2 layers] = embeddings)
  std::vector<float*> lnws_l, colws_l;
  size_t splitk_bytes;
  size_t total;
};

static BertWs bert_ws(const mmsa_bert_cfg& c, void* base) {
  BertWs w;
  Bump b(base);
  const size_t es = sdt(c) == MMSA_BF16 ? 2 : 4;
  const size_t M = (size_t)c.batch * c.seq, H = c.hidden, I = c.intermediate;
  w.e = b.take(M * H * es);
  w.x0 = b.take(M * H * es);
  w.mean0 = (float*)b.take(M * 4);
  w.rstd0 = (float*)b.take(M * 4);
  for (int l = 0; l < c.layers; ++l) {
    BertLayerWs x;
    x.qkv = b.take(M * 3 * H * es);
    x.ctx = b.take(M * H * es);
    x.s1 = b.take(M * H * es);
    x.h1 = b.take(M * H * es);
    x.pre = b.take(M * I * es);
    x.act = b.take(M * I * es);
    x.s2 = b.take(M * H * es);
    x.out = b.take(M * H * es);
    x.mean1 = (float*)b.take(M * 4);
    x.rstd1 = (float*)b.take(M * 4);
    x.mean2 = (float*)b.take(M * 4);
    x.rstd2 = (float*)b.take(M * 4);
    w.L.push_back(x);
  }
  w.pooled = b.take((size_t)c.batch * H * es);
  w.dfeat_t = b.take((size_t)c.batch * c.out_dim * es);
  w.dpool = b.take((size_t)c.batch * H * es);
  w.dprepool = b.take((size_t)c.batch * H * es);
  w.bufA = b.take(M * H * es);
  w.bufB = b.take(M * H * es);
  w.bufC = b.take(M * H * es);
  w.bufD = b.take(M * H * es);
  w.bufI = b.take(M * I * es);
  w.bufQ = b.take(M * 3 * H * es);
  w.bufS = b.take(M * H * es);
  // split-K slabs: the largest weight gradient is [I][H]; allow up to 8 slabs of it (pick_split respects the size)
  w.splitk_bytes = (size_t)8 * I * H * sizeof(float);
  w.splitk = (float*)b.take(w.splitk_bytes);
  size_t colb = colsum_ws_bytes((int)(3 * H > I ? 3 * H : I));
  w.colws = (float*)b.take(colb);
  w.colws_bytes = colb;
  w.ones8 = b.take(M * 8 * 2);
  w.q_act = w.q_w = nullptr; w.q_wscales = w.q_rowscales = w.q_batchws = nullptr;
  if (c.dtype == MMSA_FP8) {
    w.q_act = b.take(M * (H > I ? H : I));
    w.q_w = b.take((size_t)bert_layout(c).t.total);
    w.q_wscales = (float*)b.take((size_t)4 * c.layers * sizeof(float));
    w.q_rowscales = (float*)b.take(M * sizeof(float));
    w.q_batchws = (float*)b.take(fp8_quantize_batch_ws_bytes(FP8_BATCH_MAX));
  }
  w.lnws = (float*)b.take(layernorm_bwd_ws_bytes((int)H));
  for (int l = 0; l < 2 * c.layers + 1; ++l) w.lnws_l.push_back((float*)b.take(layernorm_bwd_ws_bytes((int)H)));
  for (int l = 0; l < c.layers; ++l) w.colws_l.push_back((float*)b.take(colb));
  w.attnws = (float*)b.take(attention_bwd_ws_bytes(c.batch, c.seq, c.heads));
  w.total = b.off;
  return w;
}

// MMSA_DISABLE=gelu_factor: keep the pre-activation in a.pre and evaluate gelu' in the backward epilogue (A/B hook)
static int gelu_factor() {
  static const bool off = mmsa_disabled("gelu_factor");
  return off ? 0 : 1;
}

static inline const char* at(const void* base, long off, size_t es) { return (const char*)base + (size_t)off * es; }

extern "C" {

int mmsa_bert_param_count(const mmsa_bert_cfg* c) {
  if (!c || !bert_cfg_ok(*c)) return -1;
  return (int)bert_layout(*c).t.entries.size();
}
int64_t mmsa_bert_param_total(const mmsa_bert_cfg* c) {
  if (!c || !bert_cfg_ok(*c)) return -1;
  return bert_layout(*c).t.total;
}
int mmsa_bert_param_info(const mmsa_bert_cfg* c, int idx, char* name, int name_cap, int64_t* offset, int32_t* ndim,
                         int64_t* shape) {
  if (!c || !bert_cfg_ok(*c)) return MMSA_ERR_ARG;
  const BertLayout L = bert_layout(*c);
  if (idx < 0 || idx >= (int)L.t.entries.size()) return MMSA_ERR_ARG;
  const ParamEntry& e = L.t.entries[idx];
  if ((int)e.name.size() + 1 > name_cap) return MMSA_ERR_ARG;
  strcpy(name, e.name.c_str());
  *offset = e.offset;
  *ndim = e.ndim;
  for (int i = 0; i < 4; ++i) shape[i] = e.shape[i];
  return MMSA_OK;
}
size_t mmsa_bert_ws_bytes(const mmsa_bert_cfg* c) {
  if (!c || !bert_cfg_ok(*c)) return 0;
  return bert_ws(*c, nullptr).total;
}

int mmsa_bert_fwd(const mmsa_bert_cfg* cp, const float* w32, const void* wt, const int64_t* ids, const float* mask,
                  void* ws_base, float* feat, void* stream) {
  if (!cp || !bert_cfg_ok(*cp) || !w32 || !wt || !ids || !ws_base || !feat) return MMSA_ERR_ARG;
  const mmsa_bert_cfg& c = *cp;
  const BertLayout lay = bert_layout(c);
  BertWs ws = bert_ws(c, ws_base);
  hipStream_t st = (hipStream_t)stream;
  Eng e{sdt(c), st, ws.splitk, ws.splitk_bytes, ws.colws};
  const size_t es = e.esz();
  const int M = c.batch * c.seq, H = c.hidden, I = c.intermediate, S = c.seq;
  const int aimpl = e.attn_impl();
  auto W = [&](long off) { return (const void*)at(wt, off, es); };
  auto P = [&](long off) { return w32 + off; };

  // fp8 mode: the weights of the 4 Linears of every layer are quantized up front (per-tensor scales from their own amax; two
  // launches for all of them), a Linear's input per token right before its GEMM (one pass)
  bool fp8 = c.dtype == MMSA_FP8 && !Eng::force_simt();
  if (fp8) {
    std::vector<long> off, num;
    for (int l = 0; l < c.layers; ++l) {
      const BertLayerOff& f = lay.L[l];
      const long o[4] = {f.wqkv, f.wo, f.w1, f.w2};
      const long n[4] = {3L * H * H, (long)H * H, (long)I * H, (long)H * I};
      for (int k = 0; k < 4; ++k) { off.push_back(o[k]); num.push_back(n[k]); }
    }
    for (size_t t0 = 0; t0 < off.size() && fp8; t0 += FP8_BATCH_MAX) {
      const int cnt = (int)(off.size() - t0 < FP8_BATCH_MAX ? off.size() - t0 : FP8_BATCH_MAX);
      const int rc = fp8_quantize_batch(wt, off.data() + t0, num.data() + t0, cnt, ws.q_w, ws.q_wscales + t0, ws.q_batchws, st);
      if (rc == MMSA_ERR_ARG) fp8 = false;  // (a width that is not a multiple of 8: the bf16 path takes the whole forward)
      else RET_IF(rc);
    }
  }
  // (k = 0, 2: the input is a LayerNorm output whose per-token e4m3 image that LayerNorm wrote itself — no quantizer pass)
  const bool ln_q = fp8 && !(H % 128) && !mmsa_disabled("fp8_ln_fuse");
  void* lnq = ln_q ? ws.q_act : nullptr;
  float* lnqs = ln_q ? ws.q_rowscales : nullptr;
  auto lin = [&](int l, int k, const void* x, long ldx, long woff, const float* bias, void* y, long ldy, int Mr, int N, int K, int act,
                 void* pre, const void* add, long ldadd, int pre_is_gelu_grad) -> int {
    if (fp8) {
      const int rc = e.linear_fwd_fp8(x, ldx, (const char*)ws.q_w + woff, ws.q_wscales + 4 * l + k, bias, y, ldy, Mr, N, K, act, pre,
                                      add, ldadd, pre_is_gelu_grad, ws.q_act, ws.q_rowscales, ln_q && (k == 0 || k == 2));
      if (rc != MMSA_ERR_UNSUPPORTED) return rc;
    }
    return e.linear_fwd(x, ldx, W(woff), bias, y, ldy, Mr, N, K, act, pre, add, ldadd, 0, pre_is_gelu_grad);
  };
  RET_IF(embed_gather(sdt(c), (const long long*)ids, W(lay.word), W(lay.pos), W(lay.type), ws.e, M, S, H, c.vocab, st));
  RET_IF(layernorm_fwd(sdt(c), ws.e, P(lay.lnw), P(lay.lnb), ws.x0, ws.mean0, ws.rstd0, M, H, c.ln_eps, st, lnq, lnqs));
  const void* x = ws.x0;
  for (int l = 0; l < c.layers; ++l) {
    const BertLayerOff& f = lay.L[l];
    BertLayerWs& a = ws.L[l];
    RET_IF(lin(l, 0, x, H, f.wqkv, P(f.bqkv), a.qkv, 3 * H, M, 3 * H, H, MMSA_ACT_NONE, nullptr, nullptr, 0, 0));
    RET_IF(attention_fwd(aimpl, a.qkv, mask, a.ctx, c.batch, S, c.heads, 64, st));
    RET_IF(lin(l, 1, a.ctx, H, f.wo, P(f.bo), a.s1, H, M, H, H, MMSA_ACT_NONE, nullptr, x, H, 0));
    RET_IF(layernorm_fwd(sdt(c), a.s1, P(f.ln1w), P(f.ln1b), a.h1, a.mean1, a.rstd1, M, H, c.ln_eps, st, lnq, lnqs));
    // a.pre receives gelu'(pre-activation): the factor the backward multiplies by (one exp / erf for both outputs)
    RET_IF(lin(l, 2, a.h1, H, f.w1, P(f.b1), a.act, I, M, I, H, MMSA_ACT_GELU, a.pre, nullptr, 0, gelu_factor()));
    RET_IF(lin(l, 3, a.act, I, f.w2, P(f.b2), a.s2, H, M, H, I, MMSA_ACT_NONE, nullptr, a.h1, H, 0));
    RET_IF(layernorm_fwd(sdt(c), a.s2, P(f.ln2w), P(f.ln2b), a.out, a.mean2, a.rstd2, M, H, c.ln_eps, st,
                         l + 1 < c.layers ? lnq : nullptr, lnqs));
    x = a.out;
  }
  // pooler on the first token of every sequence (row stride S*H), then the projection into the fusion width (fp32 out)
  RET_IF(e.linear_fwd(x, (long)S * H, W(lay.wp), P(lay.bp), ws.pooled, H, c.batch, H, H, MMSA_ACT_TANH));
  RET_IF(e.linear_fwd(ws.pooled, H, W(lay.wproj), P(lay.bproj), feat, c.out_dim, c.batch, c.out_dim, H, MMSA_ACT_NONE,
                      nullptr, nullptr, 0, 1));
  return MMSA_OK;
}

int mmsa_bert_bwd(const mmsa_bert_cfg* cp, const float* w32, const void* wt, const int64_t* ids, const float* mask,
                  void* ws_base, const float* dfeat, float* grad, int32_t accumulate, void* stream) {
  return mmsa_bert_bwd_cb(cp, w32, wt, ids, mask, ws_base, dfeat, grad, accumulate, stream, nullptr, nullptr, 0, nullptr);
}

// The backward with a "gradient range ready" callback: cb(user, offset, length) is called — on the host, from inside this
// call — each time the kernels that produce the parameter gradients grad[offset, offset + length) have been ENQUEUED on
// `stream` (pooler + projection first, then `layers_per_chunk` encoder layers at a time from the last layer down, the
// embeddings last). The data-parallel trainer records an event there and all-reduces that range on a side stream while
// the remaining backward runs (fused.py GradReducer; Trainer.py:79-81 needs the REDUCED gradients only at the clip).
// (Round 3 also carried a variant with the per-layer weight-gradient groups on a second stream: two persistent GEMMs that share
//  the chip only stretch each other — it measured 0.13 ms per step slower and was removed in round 4)
int mmsa_bert_bwd_cb(const mmsa_bert_cfg* cp, const float* w32, const void* wt, const int64_t* ids, const float* mask,
                     void* ws_base, const float* dfeat, float* grad, int32_t accumulate, void* stream, mmsa_range_cb cb,
                     void* user, int32_t layers_per_chunk, const uint8_t* frozen) {
  if (!cp || !bert_cfg_ok(*cp) || !w32 || !wt || !ids || !ws_base || !dfeat || !grad) return MMSA_ERR_ARG;
  if (layers_per_chunk < 1) layers_per_chunk = 1;
  const mmsa_bert_cfg& c = *cp;
  const BertLayout lay = bert_layout(c);
  BertWs ws = bert_ws(c, ws_base);
  hipStream_t st = (hipStream_t)stream;
  Eng e{sdt(c), st, ws.splitk, ws.splitk_bytes, ws.colws};
  const size_t es = e.esz();
  const int M = c.batch * c.seq, H = c.hidden, I = c.intermediate, S = c.seq, B = c.batch, D = c.out_dim;
  // Frozen parameter groups (N2: the curriculum phases of dataLoader/MultiTaskTrainer.py:50-177 and train.py:90-92 freeze
  // sub-graphs; `frozen[i]` = 1 for entry i of the parameter table): a wholly frozen encoder layer skips its four weight-
  // gradient GEMMs (half of its backward FLOPs), and the backward stops below the lowest trainable group (no data gradient is
  // computed for anything that nothing trainable is reached through). Entries: 5 embedding tensors, 16 per layer, 4 tail.
  const int nl = c.layers;
  auto all_frozen = [&](int first, int count) {
    if (!frozen) return false;
    for (int i = first; i < first + count; ++i)
      if (!frozen[i]) return false;
    return true;
  };
  const bool emb_frozen = all_frozen(0, 5), tail_frozen = all_frozen(5 + 16 * nl, 4);
  std::vector<char> layer_frozen(nl);
  int lowest = emb_frozen ? nl : -1;  // lowest trainable group: -1 = embeddings, l = encoder layer l, nl = none below the tail
  for (int l = nl - 1; l >= 0; --l) {
    layer_frozen[l] = all_frozen(5 + 16 * l, 16);
    if (!layer_frozen[l] && emb_frozen) lowest = l;
  }
  if (sdt(c) == MMSA_BF16) {  // bias gradients ride in the grouped weight-gradient launch (Eng::wgrad_group)
    RET_IF(fill_ones_bf16(ws.ones8, (long)M * 8, st));
    e.ones8 = ws.ones8;
    e.col_ws_bytes = ws.colws_bytes;
  }
  const int aimpl = e.attn_impl();
  const int acc = accumulate ? 1 : 0;
  auto W = [&](long off) { return (const void*)at(wt, off, es); };
  auto P = [&](long off) { return w32 + off; };
  auto G = [&](long off) { return grad + off; };

  const void* xL = ws.L[c.layers - 1].out;
  // projection + pooler
  RET_IF(cast_f32(sdt(c), dfeat, ws.dfeat_t, (long)B * D, st));
  if (!tail_frozen) {
    RET_IF(e.bias_grad(ws.dfeat_t, D, G(lay.bproj), B, D, acc));
    RET_IF(e.linear_wgrad(ws.dfeat_t, D, ws.pooled, H, G(lay.wproj), B, D, H, acc));
  }
  if (tail_frozen && lowest == nl) return MMSA_OK;  // nothing trainable in this encoder
  RET_IF(e.linear_dgrad(ws.dfeat_t, D, W(lay.wproj), ws.dpool, H, B, D, H));
  RET_IF(tanh_bwd(sdt(c), ws.dpool, ws.pooled, ws.dprepool, (long)B * H, st));
  if (!tail_frozen) {
    RET_IF(e.bias_grad(ws.dprepool, H, G(lay.bp), B, H, acc));
    RET_IF(e.linear_wgrad(ws.dprepool, H, xL, (long)S * H, G(lay.wp), B, H, H, acc));
    if (cb) cb(user, lay.wp, lay.t.total - lay.wp);
  }
  if (lowest == nl) return MMSA_OK;  // every encoder layer and the embeddings are frozen: the backward ends here
  void *dOut = ws.bufA, *bC = ws.bufC;
  // The finalizes of the LayerNorm parameter gradients and the picks of the grouped bias gradients feed nothing but the optimizer:
  // they are recorded and flushed as ONE launch each per announced range (every `layers_per_chunk` layers under data parallelism,
  // once per backward otherwise) instead of 37 launches of ~6 us between the GEMMs of this serial chain. MMSA_DISABLE=defer_finalize
  // launches them in place (A/B and parity switch: same kernels bodies, bit-identical gradients).
  const bool deferf = !mmsa_disabled("defer_finalize");
  std::vector<LnFinJob> ln_jobs;
  std::vector<BiasPickJob> pick_jobs;
  if (deferf) e.defer_picks = &pick_jobs;
  auto ln_defer = [&](LnFinJob& j) { if (j.part) ln_jobs.push_back(j); };
  auto flush_deferred = [&]() -> int {
    if (!ln_jobs.empty()) RET_IF(layernorm_bwd_finalize_batch(ln_jobs.data(), (int)ln_jobs.size(), H, st));
    if (!pick_jobs.empty()) RET_IF(bias_pick_batch(pick_jobs.data(), (int)pick_jobs.size(), st));
    ln_jobs.clear();
    pick_jobs.clear();
    return MMSA_OK;
  };
  if (hipMemsetAsync(dOut, 0, (size_t)M * H * es, st) != hipSuccess) return MMSA_ERR_LAUNCH;
  RET_IF(e.linear_dgrad(ws.dprepool, H, W(lay.wp), dOut, (long)S * H, B, H, H));  // only the [CLS] rows receive gradient
  long chunk_end = lay.wp;  // encoder layers [l, ...) up to chunk_end are complete but not yet announced
  bool chunk_live = false;  // does the pending chunk hold a trainable layer?

  for (int l = c.layers - 1; l >= 0 && l >= lowest; --l) {
    const BertLayerOff& f = lay.L[l];
    BertLayerWs& a = ws.L[l];
    const void* xin = l == 0 ? ws.x0 : ws.L[l - 1].out;
    const bool last_needed = (l == lowest);  // nothing trainable below: this layer's input needs no gradient
    void* ds2 = ws.bufB;
    // ds2 is also dY of the FFN output Linear: its bias gradient (column sums of ds2) comes out of the same pass
    // (a wholly frozen layer produces NO parameter gradient: its LayerNorm / bias gradient outputs are null, so that stale
    //  gradients of an earlier phase stay what torch would keep and a data-parallel replica never steps an un-reduced range)
    const bool lf = layer_frozen[l];
    LnFinJob lj;
    RET_IF(layernorm_bwd(sdt(c), dOut, a.s2, a.mean2, a.rstd2, P(f.ln2w), ds2, lf ? nullptr : G(f.ln2w),
                         lf ? nullptr : G(f.ln2b), acc, deferf ? ws.lnws_l[2 * l] : ws.lnws, M, H, st, lf ? nullptr : G(f.b2),
                         deferf ? &lj : nullptr));
    if (deferf) ln_defer(lj);
    void* dpre = ws.bufI;
    RET_IF(e.linear_dgrad(ds2, H, W(f.w2), dpre, I, M, H, I, a.pre, I, nullptr, 0, gelu_factor()));  // * gelu'(pre), stored by the forward
    void* dh1 = bC;
    RET_IF(e.linear_dgrad(dpre, I, W(f.w1), dh1, H, M, I, H, nullptr, 0, ds2, H));  // + residual branch
    void* ds1 = ws.bufS;  // (its own buffer: dOut / bC rotate under it and the grouped weight gradients read it last)
    RET_IF(layernorm_bwd(sdt(c), dh1, a.s1, a.mean1, a.rstd1, P(f.ln1w), ds1, lf ? nullptr : G(f.ln1w),
                         lf ? nullptr : G(f.ln1b), acc, deferf ? ws.lnws_l[2 * l + 1] : ws.lnws, M, H, st,
                         lf ? nullptr : G(f.bo), deferf ? &lj : nullptr));  // + bias gradient of the attention output Linear
    if (deferf) ln_defer(lj);
    void* dctx = ws.bufD;
    RET_IF(e.linear_dgrad(ds1, H, W(f.wo), dctx, H, M, H, H));
    void* dqkv = ws.bufQ;
    RET_IF(attention_bwd(aimpl, a.qkv, mask, dctx, dqkv, ws.attnws, B, S, c.heads, 64, st));
    void* dx = bC;
    if (!last_needed)
      RET_IF(e.linear_dgrad(dqkv, 3 * H, W(f.wqkv), dx, H, M, 3 * H, H, nullptr, 0, ds1, H));  // + residual branch
    // the layer's four weight gradients (K = B*S rows each, 18-72 tiles of 256x128) as one launch: together they fill
    // the chip without a K split; the two bias gradients the LayerNorm backward does not produce ride along as column sums
    if (!layer_frozen[l]) {
      chunk_live = true;
      const Eng::WgradJob jobs[4] = {
          {ds2, H, a.act, I, G(f.w2), nullptr, H, I},
          {dpre, I, a.h1, H, G(f.w1), G(f.b1), I, H},
          {ds1, H, a.ctx, H, G(f.wo), nullptr, H, H},
          {dqkv, 3L * H, xin, H, G(f.wqkv), G(f.bqkv), 3 * H, H},
      };
      if (deferf) e.col_ws = ws.colws_l[l];  // (this layer's bias results stay put until the flush)
      RET_IF(e.wgrad_group(jobs, 4, M, acc));
      e.col_ws = ws.colws;
    }
    // rotate: dx becomes the next layer's dOut
    void* t = dOut; dOut = bC; bC = t;
    if ((c.layers - l) % layers_per_chunk == 0 || l == 0 || last_needed) {
      if (cb && chunk_live) {
        RET_IF(flush_deferred());  // an announced range is complete
        cb(user, f.wqkv, chunk_end - f.wqkv);
      }
      chunk_end = f.wqkv;
      chunk_live = false;
    }
  }
  if (lowest >= 0) return flush_deferred();  // the embeddings are frozen
  // embeddings
  void* de = ws.bufB;
  {
    LnFinJob lj;
    RET_IF(layernorm_bwd(sdt(c), dOut, ws.e, ws.mean0, ws.rstd0, P(lay.lnw), de, G(lay.lnw), G(lay.lnb), acc,
                         deferf ? ws.lnws_l[2 * c.layers] : ws.lnws, M, H, st, nullptr, deferf ? &lj : nullptr));
    if (deferf) ln_defer(lj);
  }
  if (!acc && c.type_vocab > 1 &&
      hipMemsetAsync(G(lay.type) + H, 0, (size_t)(c.type_vocab - 1) * H * sizeof(float), st) != hipSuccess)
    return MMSA_ERR_LAUNCH;
  RET_IF(embed_backward(sdt(c), (const long long*)ids, de, G(lay.word), G(lay.pos), G(lay.type), acc, ws.colws, B, S, H,
                        c.vocab, c.max_pos, st));
  RET_IF(flush_deferred());
  if (cb) cb(user, 0, lay.L[0].wqkv);
  return MMSA_OK;
}

}  // extern "C"
