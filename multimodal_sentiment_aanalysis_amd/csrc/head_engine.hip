// Fusion-head engines (fp32): each reference fusion module as one forward and one backward call.
//   kind 0  CrossModalTransformer   MultimodalModel.py:108-149   (A1)
//   kind 1  ME-MHACL fusion         MultimodalModel.py:374-404   (A2: L2-norm, 8-head MHA over modality tokens, pool, MLP+BN)
//   kind 2  weighted head           MultimodalModel.py:171-225, 298-313 (A4: dynamic weights, fusion MLP, arousal [+valence] head)
//   kind 3  Classifier              MultimodalModel.py:432-451   (A5)
//   kind 4  ProjectionHead          MultimodalModel.py:409-429   (A5)
// Parameters of a module live in one flat fp32 buffer (layout via mmsa_head_param_info, names = the reference's
// state_dict keys), BatchNorm running statistics in a second one; gradients mirror the parameter layout.
// Inputs / outputs are passed as arrays of device pointers (meaning per kind documented in include/mmsa.h).
#include "../../include/mmsa.h"
#include "engine_common.h"

struct LinP { long w, b; int in, out; };
struct BnP { long g, b, rm, rv; int c; };

static LinP add_lin(ParamTable& t, const std::string& n, int in, int out) {
  LinP l; l.in = in; l.out = out;
  l.w = t.add(n + ".weight", {out, in});
  l.b = t.add(n + ".bias", {out});
  return l;
}
static BnP add_bn(ParamTable& t, ParamTable& bt, const std::string& n, int c) {
  BnP b; b.c = c;
  b.g = t.add(n + ".weight", {c});
  b.b = t.add(n + ".bias", {c});
  b.rm = bt.add(n + ".running_mean", {c});
  b.rv = bt.add(n + ".running_var", {c});
  return b;
}
struct MhaP { long inw, inb, outw, outb; };
static MhaP add_mha(ParamTable& t, const std::string& n, int E) {
  MhaP m;
  m.inw = t.add(n + ".in_proj_weight", {3 * E, E});
  m.inb = t.add(n + ".in_proj_bias", {3 * E});
  m.outw = t.add(n + ".out_proj.weight", {E, E});
  m.outb = t.add(n + ".out_proj.bias", {E});
  return m;
}

struct HeadLayout {
  ParamTable t, bt;
  // kind 0 / 1
  MhaP mha; LinP gate; long lnw, lnb; LinP fm0; BnP fm2;
  // kind 2
  LinP aw0, aw2, fu0, fu4, ar0, ar4; BnP fu1, fu5, ar1;
  LinP va[5]; BnP vb[4];
  // kind 3 / 4
  LinP sh, fa, fv, p0, p4, p8; BnP p2, p6;
};

static bool head_cfg_ok(int kind, const mmsa_head_cfg& c) {
  if (kind < 0 || kind > 4 || c.batch <= 0 || c.embed % 64 || c.embed <= 0) return false;
  if (kind == 0 && (c.tokens < 1 || c.tokens > 4 || c.heads <= 0)) return false;
  if (kind == 1 && (c.tokens < 1 || c.tokens > 4 || c.heads <= 0)) return false;
  if (kind == 2 && (c.num_classes < 1 || c.num_classes > 64)) return false;
  return true;
}

static HeadLayout head_layout(int kind, const mmsa_head_cfg& c) {
  HeadLayout L;
  const int E = c.embed;
  if (kind == 0) {
    L.mha = add_mha(L.t, "multihead_attn", E);
    L.gate = add_lin(L.t, "gate.0", 2 * E, E);
    L.lnw = L.t.add("norm.weight", {E});
    L.lnb = L.t.add("norm.bias", {E});
  } else if (kind == 1) {
    L.mha = add_mha(L.t, "multihead_attn", E);
    L.fm0 = add_lin(L.t, "fusion_mlp.0", E, E);
    L.fm2 = add_bn(L.t, L.bt, "fusion_mlp.2", E);
  } else if (kind == 2) {
    L.aw0 = add_lin(L.t, "attention_weights.0", 3 * E, 64);
    L.aw2 = add_lin(L.t, "attention_weights.2", 64, 3);
    L.fu0 = add_lin(L.t, "fusion.0", 3 * E, 256);
    L.fu1 = add_bn(L.t, L.bt, "fusion.1", 256);
    L.fu4 = add_lin(L.t, "fusion.4", 256, 128);
    L.fu5 = add_bn(L.t, L.bt, "fusion.5", 128);
    L.ar0 = add_lin(L.t, "arousal_head.0", 128, 128);
    L.ar1 = add_bn(L.t, L.bt, "arousal_head.1", 128);
    L.ar4 = add_lin(L.t, "arousal_head.4", 128, c.num_classes);
    if (c.valence) {
      const int dims[6] = {128, 256, 256, 128, 64, c.num_classes};
      for (int i = 0; i < 5; ++i) {
        L.va[i] = add_lin(L.t, "valence_head." + std::to_string(4 * i), dims[i], dims[i + 1]);
        if (i < 4) L.vb[i] = add_bn(L.t, L.bt, "valence_head." + std::to_string(4 * i + 1), dims[i + 1]);
      }
    }
  } else if (kind == 3) {
    L.sh = add_lin(L.t, "shared.0", E, c.hidden);
    L.fa = add_lin(L.t, "fc_arousal", c.hidden, c.num_classes);
    L.fv = add_lin(L.t, "fc_valence", c.hidden, c.num_classes);
  } else {
    L.p0 = add_lin(L.t, "net.0", E, c.hidden);
    L.p2 = add_bn(L.t, L.bt, "net.2", c.hidden);
    L.p4 = add_lin(L.t, "net.4", c.hidden, c.out_dim);
    L.p6 = add_bn(L.t, L.bt, "net.6", c.out_dim);
    L.p8 = add_lin(L.t, "net.8", c.out_dim, c.out_dim);
  }
  return L;
}

// ---- context shared by the module bodies
struct HCtx {
  const mmsa_head_cfg& c;
  Eng e;
  const float* w;
  float* bn;
  float* g;
  int acc;
  float* bnws;
  float* lnws;
  unsigned long long seed;
  const float* P(long o) const { return w + o; }
  float* G(long o) const { return g + o; }
  int lin_fwd(const LinP& l, const float* x, long ldx, float* y, long ldy, int M, int act = MMSA_ACT_NONE, float* pre = nullptr) const {
    return e.linear_fwd(x, ldx, P(l.w), P(l.b), y, ldy, M, l.out, l.in, act, pre);
  }
  // dx may be null (input needs no gradient); mul = pre-activation for a fused gelu'
  int lin_bwd(const LinP& l, const float* dy, long lddy, const float* x, long ldx, float* dx, long lddx, int M,
              const float* mul = nullptr, long ldmul = 0) const {
    return e.linear_bwd(dy, lddy, x, ldx, P(l.w), g ? G(l.w) : nullptr, g ? G(l.b) : nullptr, dx, lddx, M, l.out, l.in, acc, mul,
                        ldmul);
  }
  int bn_fwd(const BnP& b, const float* x, float* y, float* mean, float* invstd, int M, int act) const {
    return bn_forward(MMSA_F32, x, P(b.g), P(b.b), bn + b.rm, bn + b.rv, mean, invstd, nullptr, y, bnws, M, b.c, c.bn_eps,
                      c.bn_momentum, act, c.training, e.st);
  }
  int bn_bwd(const BnP& b, const float* dy, const float* x, const float* y, const float* mean, const float* invstd, float* dx,
             int M, int act) const {
    return bn_backward(MMSA_F32, dy, x, y, mean, invstd, P(b.g), P(b.b), dx, nullptr, g ? G(b.g) : nullptr,
                       g ? G(b.b) : nullptr, acc, bnws, M, b.c, act, c.training, e.st);
  }
  bool drop() const { return c.training && c.dropout_p > 0.f; }
};

// Linear -> BatchNorm -> GELU -> Dropout unit (fusion / arousal / valence heads, MultimodalModel.py:179-225)
struct LbgWs { float *z, *y, *yd, *mean, *invstd; unsigned char* mask; float *dyb, *dz; };
static LbgWs lbg_ws(Bump& b, int M, int n) {
  LbgWs w;
  w.z = (float*)b.take((size_t)M * n * 4); w.y = (float*)b.take((size_t)M * n * 4); w.yd = (float*)b.take((size_t)M * n * 4);
  w.mean = (float*)b.take(n * 4); w.invstd = (float*)b.take(n * 4); w.mask = (unsigned char*)b.take((size_t)M * n);
  w.dyb = (float*)b.take((size_t)M * n * 4); w.dz = (float*)b.take((size_t)M * n * 4);
  return w;
}
// out2 (optional): a second destination of the unit's output; *copied says whether the launch wrote it (else the caller copies)
static int lbg_fwd(const HCtx& h, const LinP& l, const BnP& bn, const float* x, long ldx, LbgWs& w, int M, int salt, const float** out,
                   float* out2 = nullptr, bool* copied = nullptr) {
  if (h.c.training && head_units_on()) {  // the whole unit as one launch (head_fused.hip)
    UnitFwd f;
    f.x = x; f.ldx = ldx; f.W = h.P(l.w); f.bias = h.P(l.b); f.gamma = h.P(bn.g); f.beta = h.P(bn.b);
    f.rmean = h.bn + bn.rm; f.rvar = h.bn + bn.rv;
    f.z = w.z; f.y = w.y; f.yd = h.drop() ? w.yd : nullptr; f.mask = h.drop() ? w.mask : nullptr; f.out2 = out2;
    f.mean = w.mean; f.invstd = w.invstd;
    f.M = M; f.N = l.out; f.lin_act = MMSA_ACT_NONE; f.bn_act = MMSA_ACT_GELU;
    f.eps = h.c.bn_eps; f.momentum = h.c.bn_momentum; f.drop_p = h.c.dropout_p; f.seed = h.seed + 0x1000ull * salt;
    const int rc = unit_fwd_fused(f, l.in, h.e.st);
    if (rc == MMSA_OK) { *out = h.drop() ? w.yd : w.y; if (copied) *copied = out2 != nullptr; return MMSA_OK; }
    if (rc != MMSA_ERR_UNSUPPORTED) return rc;
  }
  if (copied) *copied = false;
  RET_IF(h.lin_fwd(l, x, ldx, w.z, l.out, M));
  RET_IF(h.bn_fwd(bn, w.z, w.y, w.mean, w.invstd, M, MMSA_ACT_GELU));
  *out = w.y;
  if (h.drop()) {
    RET_IF(dropout_fwd(w.y, w.yd, w.mask, (long)M * l.out, h.c.dropout_p, h.seed + 0x1000ull * salt, h.e.st));
    *out = w.yd;
  }
  return MMSA_OK;
}
static int lbg_bwd(const HCtx& h, const LinP& l, const BnP& bn, const float* dy, const float* x, long ldx, LbgWs& w, float* dx,
                   long lddx, int M) {
  if (head_units_on()) {  // Dropout backward + BatchNorm backward as one launch
    const int rc = bn_small_backward_unit(dy, w.z, w.mean, w.invstd, h.P(bn.g), h.P(bn.b), w.dz, h.g ? h.G(bn.g) : nullptr,
                                          h.g ? h.G(bn.b) : nullptr, h.acc, M, bn.c, MMSA_ACT_GELU, h.c.training,
                                          h.drop() ? w.mask : nullptr, h.c.dropout_p, 0, h.e.st);
    if (rc == MMSA_OK) return h.lin_bwd(l, w.dz, l.out, x, ldx, dx, lddx, M);
    if (rc != MMSA_ERR_UNSUPPORTED) return rc;
  }
  const float* d = dy;
  if (h.drop()) { RET_IF(dropout_bwd(dy, w.mask, w.dyb, (long)M * l.out, h.c.dropout_p, h.e.st)); d = w.dyb; }
  RET_IF(h.bn_bwd(bn, d, w.z, nullptr, w.mean, w.invstd, w.dz, M, MMSA_ACT_GELU));
  return h.lin_bwd(l, w.dz, l.out, x, ldx, dx, lddx, M);
}

// Linear -> ReLU -> BatchNorm -> Dropout unit (fusion_mlp :377-381, ProjectionHead :416-424)
struct LrbWs { float *r, *y, *yd, *mean, *invstd; unsigned char* mask; float *dyb, *dr, *dpre; };
static LrbWs lrb_ws(Bump& b, int M, int n) {
  LrbWs w;
  w.r = (float*)b.take((size_t)M * n * 4); w.y = (float*)b.take((size_t)M * n * 4); w.yd = (float*)b.take((size_t)M * n * 4);
  w.mean = (float*)b.take(n * 4); w.invstd = (float*)b.take(n * 4); w.mask = (unsigned char*)b.take((size_t)M * n);
  w.dyb = (float*)b.take((size_t)M * n * 4); w.dr = (float*)b.take((size_t)M * n * 4); w.dpre = (float*)b.take((size_t)M * n * 4);
  return w;
}
static int lrb_fwd(const HCtx& h, const LinP& l, const BnP& bn, const float* x, long ldx, LrbWs& w, int M, float drop_p, int salt,
                   const float** out, float* direct_out = nullptr) {
  if (h.c.training && head_units_on()) {  // the whole unit as one launch (head_fused.hip)
    const bool dr = drop_p > 0.f;
    UnitFwd f;
    f.x = x; f.ldx = ldx; f.W = h.P(l.w); f.bias = h.P(l.b); f.gamma = h.P(bn.g); f.beta = h.P(bn.b);
    f.rmean = h.bn + bn.rm; f.rvar = h.bn + bn.rv;
    f.z = w.r; f.y = (direct_out && !dr) ? direct_out : w.y; f.yd = dr ? (direct_out ? direct_out : w.yd) : nullptr;
    f.mask = dr ? w.mask : nullptr; f.out2 = nullptr;
    f.mean = w.mean; f.invstd = w.invstd;
    f.M = M; f.N = l.out; f.lin_act = MMSA_ACT_RELU; f.bn_act = MMSA_ACT_NONE;
    f.eps = h.c.bn_eps; f.momentum = h.c.bn_momentum; f.drop_p = drop_p; f.seed = h.seed + 0x1000ull * salt;
    const int rc = unit_fwd_fused(f, l.in, h.e.st);
    if (rc == MMSA_OK) { *out = dr ? f.yd : f.y; return MMSA_OK; }
    if (rc != MMSA_ERR_UNSUPPORTED) return rc;
  }
  RET_IF(h.lin_fwd(l, x, ldx, w.r, l.out, M, MMSA_ACT_RELU));
  float* y = (direct_out && !(h.c.training && drop_p > 0.f)) ? direct_out : w.y;
  RET_IF(h.bn_fwd(bn, w.r, y, w.mean, w.invstd, M, MMSA_ACT_NONE));
  *out = y;
  if (h.c.training && drop_p > 0.f) {
    float* yd = direct_out ? direct_out : w.yd;
    RET_IF(dropout_fwd(w.y, yd, w.mask, (long)M * l.out, drop_p, h.seed + 0x1000ull * salt, h.e.st));
    *out = yd;
  }
  return MMSA_OK;
}
static int lrb_bwd(const HCtx& h, const LinP& l, const BnP& bn, const float* dy, const float* x, long ldx, LrbWs& w, float drop_p,
                   float* dx, long lddx, int M) {
  if (head_units_on()) {  // Dropout backward + BatchNorm backward + ReLU backward as one launch
    const bool dr = h.c.training && drop_p > 0.f;
    const int rc = bn_small_backward_unit(dy, w.r, w.mean, w.invstd, h.P(bn.g), h.P(bn.b), w.dpre, h.g ? h.G(bn.g) : nullptr,
                                          h.g ? h.G(bn.b) : nullptr, h.acc, M, bn.c, MMSA_ACT_NONE, h.c.training,
                                          dr ? w.mask : nullptr, drop_p, 1, h.e.st);
    if (rc == MMSA_OK) return h.lin_bwd(l, w.dpre, l.out, x, ldx, dx, lddx, M);
    if (rc != MMSA_ERR_UNSUPPORTED) return rc;
  }
  const float* d = dy;
  if (h.c.training && drop_p > 0.f) { RET_IF(dropout_bwd(dy, w.mask, w.dyb, (long)M * l.out, drop_p, h.e.st)); d = w.dyb; }
  RET_IF(h.bn_bwd(bn, d, w.r, nullptr, w.mean, w.invstd, w.dr, M, MMSA_ACT_NONE));
  RET_IF(ew2d(EW_RELU_BWD, w.dr, l.out, w.r, l.out, w.dpre, l.out, M, l.out, h.e.st));
  return h.lin_bwd(l, w.dpre, l.out, x, ldx, dx, lddx, M);
}

// ------------------------------------------------------------------------------------------------ workspaces
struct HeadWs {
  // kind 0
  float *qp, *kp, *vp, *ctx, *probs, *cat, *g, *mix, *mean, *rstd, *dmix, *dcat, *dgpre, *dctx, *dqp, *dkp, *dvp, *dq2;
  // kind 1
  float *nrm, *seqn, *qkv, *attn, *pooled, *dpooled, *dattn, *dctx1, *dqkv, *dseq, *dfused;
  unsigned char* pidx;
  LrbWs fm;
  // kind 2
  float *cat3, *awpre, *awact, *wl, *wsm, *wcat, *dwcat, *dwl, *dawact, *dcat3, *dlast, *dlast2, *df[3];
  LbgWs fu0, fu4, ar0, va[4];
  float* dfused2;
  // kind 3
  float *sh, *shd, *dsh, *dsh2, *dshpre; unsigned char* shmask;
  // kind 4
  LrbWs p0, p4; float* dmid;
  float *splitk, *colws, *bnws, *lnws;
  size_t splitk_bytes, total;
};

static HeadWs head_ws(int kind, const mmsa_head_cfg& c, void* base) {
  HeadWs w;
  memset(&w, 0, sizeof(w));
  Bump b(base);
  const int B = c.batch, E = c.embed, L = c.tokens > 0 ? c.tokens : 1;
  auto F = [&](size_t n) { return (float*)b.take(n * 4); };
  if (kind == 0) {
    w.qp = F((size_t)B * E); w.kp = F((size_t)B * L * E); w.vp = F((size_t)B * L * E); w.ctx = F((size_t)B * E);
    w.probs = F((size_t)B * c.heads * L); w.cat = F((size_t)B * 2 * E); w.g = F((size_t)B * E); w.mix = F((size_t)B * E);
    w.mean = F(B); w.rstd = F(B); w.dmix = F((size_t)B * E); w.dcat = F((size_t)B * 2 * E); w.dgpre = F((size_t)B * E);
    w.dctx = F((size_t)B * E); w.dqp = F((size_t)B * E); w.dkp = F((size_t)B * L * E); w.dvp = F((size_t)B * L * E);
    w.dq2 = F((size_t)B * 2 * E);
  } else if (kind == 1) {
    w.nrm = F((size_t)L * B); w.seqn = F((size_t)B * L * E); w.qkv = F((size_t)B * L * 3 * E); w.ctx = F((size_t)B * L * E);
    w.probs = F((size_t)B * c.heads * L * L); w.attn = F((size_t)B * L * E); w.pooled = F((size_t)B * E);
    w.pidx = (unsigned char*)b.take((size_t)B * E);
    w.fm = lrb_ws(b, B, E);
    w.dpooled = F((size_t)B * E); w.dattn = F((size_t)B * L * E); w.dctx1 = F((size_t)B * L * E); w.dqkv = F((size_t)B * L * 3 * E);
    w.dseq = F((size_t)B * L * E);
  } else if (kind == 2) {
    w.cat3 = F((size_t)B * 3 * E); w.awpre = F((size_t)B * 64); w.awact = F((size_t)B * 64); w.wl = F((size_t)B * 4);
    w.wsm = F((size_t)B * 4); w.wcat = F((size_t)B * 3 * E);
    w.fu0 = lbg_ws(b, B, 256); w.fu4 = lbg_ws(b, B, 128); w.ar0 = lbg_ws(b, B, 128);
    if (c.valence) { const int d[4] = {256, 256, 128, 64}; for (int i = 0; i < 4; ++i) w.va[i] = lbg_ws(b, B, d[i]); }
    w.dwcat = F((size_t)B * 3 * E); w.dwl = F((size_t)B * 4); w.dawact = F((size_t)B * 64); w.dcat3 = F((size_t)B * 3 * E);
    w.dlast = F((size_t)B * 256); w.dlast2 = F((size_t)B * 256); w.dfused2 = F((size_t)B * 128);
    for (int i = 0; i < 3; ++i) w.df[i] = F((size_t)B * E);
  } else if (kind == 3) {
    const size_t n = (size_t)B * c.hidden;
    w.sh = F(n); w.shd = F(n); w.dsh = F(n); w.dsh2 = F(n); w.dshpre = F(n); w.shmask = (unsigned char*)b.take(n);
  } else {
    w.p0 = lrb_ws(b, B, c.hidden); w.p4 = lrb_ws(b, B, c.out_dim); w.dmid = F((size_t)B * (c.hidden > c.out_dim ? c.hidden : c.out_dim));
  }
  w.splitk_bytes = (size_t)4 * 3 * E * 3 * E * 4;
  w.splitk = (float*)b.take(w.splitk_bytes);
  w.colws = (float*)b.take(colsum_ws_bytes(3 * E > 1024 ? 3 * E : 1024));
  w.bnws = (float*)b.take(bn_ws_bytes(E > 256 ? E : 256));
  w.lnws = (float*)b.take(layernorm_bwd_ws_bytes(E));
  w.total = b.off;
  return w;
}

// ------------------------------------------------------------------------------------------------ kind 0: A1
// inputs: {query [B,E], key [B,Lk,E], value [B,Lk,E]}  outputs: {out [B,E]}
static int cross_fwd(const HCtx& h, const HeadLayout& L, HeadWs& w, const float* const* in, float* const* out) {
  const int B = h.c.batch, E = h.c.embed, Lk = h.c.tokens;
  const float *q = in[0], *k = in[1], *v = in[2];
  const Eng& e = h.e;
  // packed in-projection: rows [0,E) = Wq, [E,2E) = Wk, [2E,3E) = Wv (torch MultiheadAttention)
  // One key (Lk == 1: how the reference's forward calls it, MultimodalModel.py:132-137 unsqueezes [B,256] to a length-1
  // sequence): softmax over a single key is exactly 1 for every head, so the context IS the projected value — the query /
  // key projections cannot reach the output (their gradients are exactly zero too, cross_bwd). Two GEMMs + the core skipped.
  const bool one_key = (Lk == 1);
  const float* ctx = w.ctx;
  if (one_key) {
    RET_IF(e.linear_fwd(v, E, h.P(L.mha.inw) + 2L * E * E, h.P(L.mha.inb) + 2 * E, w.vp, E, B, E, E));
    ctx = w.vp;
  } else {
    RET_IF(e.linear_fwd(q, E, h.P(L.mha.inw), h.P(L.mha.inb), w.qp, E, B, E, E));
    RET_IF(e.linear_fwd(k, E, h.P(L.mha.inw) + (long)E * E, h.P(L.mha.inb) + E, w.kp, E, B * Lk, E, E));
    RET_IF(e.linear_fwd(v, E, h.P(L.mha.inw) + 2L * E * E, h.P(L.mha.inb) + 2 * E, w.vp, E, B * Lk, E, E));
    RET_IF(mha_core_fwd(w.qp, E, w.kp, E, w.vp, E, w.ctx, E, w.probs, B, 1, Lk, E, h.c.heads, e.st));
  }
  // cat = [query | attn_out]: the out-projection writes straight into the right half
  RET_IF(ew2d(EW_COPY, q, E, nullptr, 0, w.cat, 2 * E, B, E, e.st));
  RET_IF(e.linear_fwd(ctx, E, h.P(L.mha.outw), h.P(L.mha.outb), w.cat + E, 2 * E, B, E, E));
  RET_IF(h.lin_fwd(L.gate, w.cat, 2 * E, w.g, E, B, MMSA_ACT_SIGMOID));
  RET_IF(gate_mix_fwd(w.g, w.cat, 2 * E, w.cat + E, 2 * E, w.mix, B, E, e.st));
  return layernorm_fwd(MMSA_F32, w.mix, h.P(L.lnw), h.P(L.lnb), out[0], w.mean, w.rstd, B, E, h.c.ln_eps, e.st);
}
// douts: {dout [B,E]}  dinputs: {dquery, dkey, dvalue}
static int cross_bwd(const HCtx& h, const HeadLayout& L, HeadWs& w, const float* const* in, const float* const* dout, float* const* din) {
  const int B = h.c.batch, E = h.c.embed, Lk = h.c.tokens;
  const float *q = in[0], *k = in[1], *v = in[2];
  const Eng& e = h.e;
  RET_IF(layernorm_bwd(MMSA_F32, dout[0], w.mix, w.mean, w.rstd, h.P(L.lnw), w.dmix, h.g ? h.G(L.lnw) : w.dgpre,
                       h.g ? h.G(L.lnb) : w.dgpre, h.g ? h.acc : 0, w.lnws, B, E, e.st));
  // dcat = [dq_mix | da_mix] from the mix, then += gate Linear data gradient
  RET_IF(gate_mix_bwd(w.dmix, w.g, w.cat, 2 * E, w.cat + E, 2 * E, w.dqp, w.dctx, w.dgpre, B, E, e.st));
  RET_IF(h.lin_bwd(L.gate, w.dgpre, E, w.cat, 2 * E, w.dcat, 2 * E, B));
  RET_IF(ew2d(EW_ADD, w.dcat, 2 * E, w.dqp, E, w.dq2, E, B, E, e.st));           // d query (direct paths)
  RET_IF(ew2d(EW_ADD, w.dcat + E, 2 * E, w.dctx, E, w.dmix, E, B, E, e.st));     // d attn_out
  // out-projection
  const bool one_key = (Lk == 1);
  // out-projection backward (one launch): d(context) lands in dvp when the context was the projected value itself
  RET_IF(e.linear_bwd(w.dmix, E, one_key ? w.vp : w.ctx, E, h.P(L.mha.outw), h.g ? h.G(L.mha.outw) : nullptr,
                      h.g ? h.G(L.mha.outb) : nullptr, one_key ? w.dvp : w.dctx, E, B, E, E, h.acc));
  if (one_key) {
    // context == projected value (cross_fwd): d(projected value) = d(context); the query / key projections get exactly zero
    // gradient (softmax over one key: p (dp - p dp) = 0), the key input too; the query only through its direct paths (dq2)
    if (h.g && !h.acc) {  // overwrite mode: the zero gradients must be written
      if (hipMemsetAsync(h.G(L.mha.inw), 0, (size_t)2 * E * E * sizeof(float), e.st) != hipSuccess) return MMSA_ERR_LAUNCH;
      if (hipMemsetAsync(h.G(L.mha.inb), 0, (size_t)2 * E * sizeof(float), e.st) != hipSuccess) return MMSA_ERR_LAUNCH;
    }
    RET_IF(ew2d(EW_COPY, w.dq2, E, nullptr, 0, din[0], E, B, E, e.st));
    if (hipMemsetAsync(din[1], 0, (size_t)B * E * sizeof(float), e.st) != hipSuccess) return MMSA_ERR_LAUNCH;
    return e.linear_bwd(w.dvp, E, v, E, h.P(L.mha.inw) + 2L * E * E, h.g ? h.G(L.mha.inw) + 2L * E * E : nullptr,
                        h.g ? h.G(L.mha.inb) + 2 * E : nullptr, din[2], E, B, E, E, h.acc);
  }
  RET_IF(mha_core_bwd(w.qp, E, w.kp, E, w.vp, E, w.probs, w.dctx, E, w.dqp, E, w.dkp, E, w.dvp, E, B, 1, Lk, E, h.c.heads, e.st));
  // in-projection (three row blocks of the packed weight)
  if (h.g) {
    RET_IF(e.bias_grad(w.dqp, E, h.G(L.mha.inb), B, E, h.acc));
    RET_IF(e.bias_grad(w.dkp, E, h.G(L.mha.inb) + E, B * Lk, E, h.acc));
    RET_IF(e.bias_grad(w.dvp, E, h.G(L.mha.inb) + 2 * E, B * Lk, E, h.acc));
    RET_IF(e.linear_wgrad(w.dqp, E, q, E, h.G(L.mha.inw), B, E, E, h.acc));
    RET_IF(e.linear_wgrad(w.dkp, E, k, E, h.G(L.mha.inw) + (long)E * E, B * Lk, E, E, h.acc));
    RET_IF(e.linear_wgrad(w.dvp, E, v, E, h.G(L.mha.inw) + 2L * E * E, B * Lk, E, E, h.acc));
  }
  GemmParams p = Eng::blank();  // dquery = dqp Wq + dq2
  p.A = w.dqp; p.lda = E; p.B = h.P(L.mha.inw); p.ldb = E; p.b_kmajor = 1; p.C = din[0]; p.ldc = E; p.M = B; p.N = E; p.K = E;
  p.add = w.dq2; p.ldadd = E;
  RET_IF(e.gemm(p));
  RET_IF(e.linear_dgrad(w.dkp, E, h.P(L.mha.inw) + (long)E * E, din[1], E, B * Lk, E, E));
  return e.linear_dgrad(w.dvp, E, h.P(L.mha.inw) + 2L * E * E, din[2], E, B * Lk, E, E);
}

// ------------------------------------------------------------------------------------------------ kind 1: A2
// inputs: {feat_0 .. feat_{M-1}} each [B,E]   outputs: {fused [B,E]}
static int mmf_fwd(const HCtx& h, const HeadLayout& L, HeadWs& w, const float* const* in, float* const* out) {
  const int B = h.c.batch, E = h.c.embed, M = h.c.tokens;
  const Eng& e = h.e;
  // L2-normalise every modality into the token sequence seqn[b][m][:]
  for (int m = 0; m < M; ++m) {
    RET_IF(l2norm_fwd_ld(in[m], w.seqn + (long)m * E, (long)M * E, w.nrm + (long)m * B, B, E, 1e-12f, e.st));
  }
  RET_IF(e.linear_fwd(w.seqn, E, h.P(L.mha.inw), h.P(L.mha.inb), w.qkv, 3 * E, B * M, 3 * E, E));
  RET_IF(mha_core_fwd(w.qkv, 3 * E, w.qkv + E, 3 * E, w.qkv + 2 * E, 3 * E, w.ctx, E, w.probs, B, M, M, E, h.c.heads, e.st));
  RET_IF(e.linear_fwd(w.ctx, E, h.P(L.mha.outw), h.P(L.mha.outb), w.attn, E, B * M, E, E));
  RET_IF(seq_pool_fwd(w.attn, w.pooled, w.pidx, B, M, E, h.c.pool_mode, e.st));
  const float* o;
  return lrb_fwd(h, L.fm0, L.fm2, w.pooled, E, w.fm, B, 0.f, 1, &o, out[0]);
}
static int mmf_bwd(const HCtx& h, const HeadLayout& L, HeadWs& w, const float* const* in, const float* const* dout, float* const* din) {
  const int B = h.c.batch, E = h.c.embed, M = h.c.tokens;
  const Eng& e = h.e;
  RET_IF(lrb_bwd(h, L.fm0, L.fm2, dout[0], w.pooled, E, w.fm, 0.f, w.dpooled, E, B));
  RET_IF(seq_pool_bwd(w.dpooled, w.pidx, w.dattn, B, M, E, h.c.pool_mode, e.st));
  RET_IF(e.linear_bwd(w.dattn, E, w.ctx, E, h.P(L.mha.outw), h.g ? h.G(L.mha.outw) : nullptr, h.g ? h.G(L.mha.outb) : nullptr,
                      w.dctx1, E, B * M, E, E, h.acc));
  RET_IF(mha_core_bwd(w.qkv, 3 * E, w.qkv + E, 3 * E, w.qkv + 2 * E, 3 * E, w.probs, w.dctx1, E, w.dqkv, 3 * E, w.dqkv + E, 3 * E,
                      w.dqkv + 2 * E, 3 * E, B, M, M, E, h.c.heads, e.st));
  RET_IF(e.linear_bwd(w.dqkv, 3 * E, w.seqn, E, h.P(L.mha.inw), h.g ? h.G(L.mha.inw) : nullptr, h.g ? h.G(L.mha.inb) : nullptr,
                      w.dseq, E, B * M, 3 * E, E, h.acc));
  for (int m = 0; m < M; ++m) {
    // modality m's rows sit at stride M*E in the token sequence
    RET_IF(l2norm_bwd_ld(w.dseq + (long)m * E, w.seqn + (long)m * E, (long)M * E, w.nrm + (long)m * B, din[m], B, E, 1e-12f, 0,
                         e.st));
  }
  return MMSA_OK;
}

// ------------------------------------------------------------------------------------------------ kind 2: A4
// inputs: {anchor, raw2, raw3, enh2, enh3} each [B,E]  outputs: {logits [B,C], fused [B,128], valence logits [B,C] (optional)}
static int wh_fwd(const HCtx& h, const HeadLayout& L, HeadWs& w, const float* const* in, float* const* out) {
  const int B = h.c.batch, E = h.c.embed;
  const Eng& e = h.e;
  RET_IF(cat3_fwd(in[0], in[1], in[2], w.cat3, B, E, e.st));
  RET_IF(h.lin_fwd(L.aw0, w.cat3, 3 * E, w.awact, 64, B, MMSA_ACT_GELU, w.awpre));
  RET_IF(h.lin_fwd(L.aw2, w.awact, 64, w.wl, 3, B));
  RET_IF(weighted_concat_fwd(w.wl, in[0], in[3], in[4], w.wsm, w.wcat, B, E, e.st));
  const float *f1, *f2, *a1;
  RET_IF(lbg_fwd(h, L.fu0, L.fu1, w.wcat, 3 * E, w.fu0, B, 2, &f1));
  bool copied = false;
  RET_IF(lbg_fwd(h, L.fu4, L.fu5, f1, 256, w.fu4, B, 3, &f2, out[1], &copied));
  if (!copied) RET_IF(ew2d(EW_COPY, f2, 128, nullptr, 0, out[1], 128, B, 128, e.st));
  RET_IF(lbg_fwd(h, L.ar0, L.ar1, f2, 128, w.ar0, B, 4, &a1));
  RET_IF(h.lin_fwd(L.ar4, a1, 128, out[0], h.c.num_classes, B));
  if (h.c.valence) {
    const float* x = f2;
    int ld = 128;
    for (int i = 0; i < 4; ++i) {
      const float* y;
      RET_IF(lbg_fwd(h, L.va[i], L.vb[i], x, ld, w.va[i], B, 5 + i, &y));
      x = y; ld = L.va[i].out;
    }
    RET_IF(h.lin_fwd(L.va[4], x, ld, out[2], h.c.num_classes, B));
  }
  return MMSA_OK;
}
// douts: {dlogits [B,C], dfused_extra [B,128] or NULL, dvalence [B,C] or NULL}   dinputs: {d anchor, d raw2, d raw3, d enh2, d enh3}
static int wh_bwd(const HCtx& h, const HeadLayout& L, HeadWs& w, const float* const* in, const float* const* dout, float* const* din) {
  const int B = h.c.batch, E = h.c.embed, C = h.c.num_classes;
  const Eng& e = h.e;
  const float* f1 = h.drop() ? w.fu0.yd : w.fu0.y;
  const float* f2 = h.drop() ? w.fu4.yd : w.fu4.y;
  const float* a1 = h.drop() ? w.ar0.yd : w.ar0.y;
  // arousal head
  RET_IF(h.lin_bwd(L.ar4, dout[0], C, a1, 128, w.dlast, 128, B));
  RET_IF(lbg_bwd(h, L.ar0, L.ar1, w.dlast, f2, 128, w.ar0, w.dfused2, 128, B));
  if (dout[1]) RET_IF(ew2d(EW_ADD, w.dfused2, 128, dout[1], 128, w.dfused2, 128, B, 128, e.st));
  if (h.c.valence && dout[2]) {
    const float* xs[5]; int lds[5];
    xs[0] = f2; lds[0] = 128;
    for (int i = 0; i < 4; ++i) { xs[i + 1] = h.drop() ? w.va[i].yd : w.va[i].y; lds[i + 1] = L.va[i].out; }
    // two [B,256] scratch buffers ping-pong the gradient down the valence MLP
    float *ga = w.dlast, *gb = w.dlast2;
    RET_IF(h.lin_bwd(L.va[4], dout[2], C, xs[4], lds[4], ga, lds[4], B));
    for (int i = 3; i >= 0; --i) {
      RET_IF(lbg_bwd(h, L.va[i], L.vb[i], ga, xs[i], lds[i], w.va[i], gb, lds[i], B));
      float* t = ga; ga = gb; gb = t;
    }
    RET_IF(ew2d(EW_ADD, w.dfused2, 128, ga, 128, w.dfused2, 128, B, 128, e.st));
  }
  RET_IF(lbg_bwd(h, L.fu4, L.fu5, w.dfused2, f1, 256, w.fu4, w.dlast, 256, B));
  RET_IF(lbg_bwd(h, L.fu0, L.fu1, w.dlast, w.wcat, 3 * E, w.fu0, w.dwcat, 3 * E, B));
  RET_IF(weighted_concat_bwd(w.dwcat, w.wsm, in[0], in[3], in[4], w.df[0], din[3], din[4], w.dwl, B, E, e.st));
  // attention_weights MLP: Linear(3E,64) GELU Linear(64,3)
  RET_IF(h.lin_bwd(L.aw2, w.dwl, 3, w.awact, 64, w.dawact, 64, B, w.awpre, 64));  // * gelu'(pre) fused
  RET_IF(h.lin_bwd(L.aw0, w.dawact, 64, w.cat3, 3 * E, w.dcat3, 3 * E, B));
  return cat3_bwd(w.dcat3, w.df[0], din[0], din[1], din[2], B, E, e.st);
}

// ------------------------------------------------------------------------------------------------ kind 3: Classifier
// inputs {x [B,E]} outputs {out_a [B,C], out_v [B,C]}; douts {d_a, d_v}; dinputs {dx}
static int cls_fwd(const HCtx& h, const HeadLayout& L, HeadWs& w, const float* const* in, float* const* out) {
  const int B = h.c.batch, E = h.c.embed, Hh = h.c.hidden;
  RET_IF(h.lin_fwd(L.sh, in[0], E, w.sh, Hh, B, MMSA_ACT_RELU));
  const float* s = w.sh;
  if (h.drop()) { RET_IF(dropout_fwd(w.sh, w.shd, w.shmask, (long)B * Hh, h.c.dropout_p, h.seed + 0x9000ull, h.e.st)); s = w.shd; }
  RET_IF(h.lin_fwd(L.fa, s, Hh, out[0], h.c.num_classes, B));
  return h.lin_fwd(L.fv, s, Hh, out[1], h.c.num_classes, B);
}
static int cls_bwd(const HCtx& h, const HeadLayout& L, HeadWs& w, const float* const* in, const float* const* dout, float* const* din) {
  const int B = h.c.batch, E = h.c.embed, Hh = h.c.hidden, C = h.c.num_classes;
  const float* s = h.drop() ? w.shd : w.sh;
  RET_IF(h.lin_bwd(L.fa, dout[0], C, s, Hh, w.dsh, Hh, B));
  RET_IF(h.lin_bwd(L.fv, dout[1], C, s, Hh, w.dsh2, Hh, B));
  RET_IF(ew2d(EW_ADD, w.dsh, Hh, w.dsh2, Hh, w.dsh, Hh, B, Hh, h.e.st));
  const float* d = w.dsh;
  if (h.drop()) { RET_IF(dropout_bwd(w.dsh, w.shmask, w.dsh2, (long)B * Hh, h.c.dropout_p, h.e.st)); d = w.dsh2; }
  RET_IF(ew2d(EW_RELU_BWD, d, Hh, w.sh, Hh, w.dshpre, Hh, B, Hh, h.e.st));
  return h.lin_bwd(L.sh, w.dshpre, Hh, in[0], E, din[0], E, B);
}

// ------------------------------------------------------------------------------------------------ kind 4: ProjectionHead
static int proj_fwd(const HCtx& h, const HeadLayout& L, HeadWs& w, const float* const* in, float* const* out) {
  const int B = h.c.batch, E = h.c.embed;
  const float *a, *b2;
  RET_IF(lrb_fwd(h, L.p0, L.p2, in[0], E, w.p0, B, h.c.dropout_p, 11, &a));
  RET_IF(lrb_fwd(h, L.p4, L.p6, a, h.c.hidden, w.p4, B, h.c.dropout_p, 12, &b2));
  return h.lin_fwd(L.p8, b2, h.c.out_dim, out[0], h.c.out_dim, B);
}
static int proj_bwd(const HCtx& h, const HeadLayout& L, HeadWs& w, const float* const* in, const float* const* dout, float* const* din) {
  const int B = h.c.batch, E = h.c.embed;
  const bool dr = h.c.training && h.c.dropout_p > 0.f;
  const float* a = dr ? w.p0.yd : w.p0.y;
  const float* b2 = dr ? w.p4.yd : w.p4.y;
  RET_IF(h.lin_bwd(L.p8, dout[0], h.c.out_dim, b2, h.c.out_dim, w.dmid, h.c.out_dim, B));
  RET_IF(lrb_bwd(h, L.p4, L.p6, w.dmid, a, h.c.hidden, w.p4, h.c.dropout_p, w.p0.dyb, h.c.hidden, B));
  // p0.dyb holds the incoming gradient of unit 0; move it so lrb_bwd's dropout scratch does not alias it
  RET_IF(ew2d(EW_COPY, w.p0.dyb, h.c.hidden, nullptr, 0, w.dmid, h.c.hidden, B, h.c.hidden, h.e.st));
  return lrb_bwd(h, L.p0, L.p2, w.dmid, in[0], E, w.p0, h.c.dropout_p, din[0], E, B);
}

extern "C" {

int mmsa_head_param_count(int32_t kind, const mmsa_head_cfg* c, int32_t buffers) {
  if (!c || !head_cfg_ok(kind, *c)) return -1;
  const HeadLayout L = head_layout(kind, *c);
  return (int)(buffers ? L.bt.entries.size() : L.t.entries.size());
}
int64_t mmsa_head_param_total(int32_t kind, const mmsa_head_cfg* c, int32_t buffers) {
  if (!c || !head_cfg_ok(kind, *c)) return -1;
  const HeadLayout L = head_layout(kind, *c);
  return buffers ? L.bt.total : L.t.total;
}
int mmsa_head_param_info(int32_t kind, const mmsa_head_cfg* c, int32_t buffers, int idx, char* name, int name_cap,
                         int64_t* offset, int32_t* ndim, int64_t* shape) {
  if (!c || !head_cfg_ok(kind, *c)) return MMSA_ERR_ARG;
  const HeadLayout L = head_layout(kind, *c);
  const ParamTable& t = buffers ? L.bt : L.t;
  if (idx < 0 || idx >= (int)t.entries.size()) return MMSA_ERR_ARG;
  const ParamEntry& e = t.entries[idx];
  if ((int)e.name.size() + 1 > name_cap) return MMSA_ERR_ARG;
  strcpy(name, e.name.c_str());
  *offset = e.offset;
  *ndim = e.ndim;
  for (int i = 0; i < 4; ++i) shape[i] = e.shape[i];
  return MMSA_OK;
}
size_t mmsa_head_ws_bytes(int32_t kind, const mmsa_head_cfg* c) {
  if (!c || !head_cfg_ok(kind, *c)) return 0;
  return head_ws(kind, *c, nullptr).total;
}

int mmsa_head_fwd(int32_t kind, const mmsa_head_cfg* cp, const float* w, float* bnbuf, const float* const* inputs,
                  float* const* outputs, void* ws_base, void* stream) {
  if (!cp || !head_cfg_ok(kind, *cp) || !w || !inputs || !outputs || !ws_base) return MMSA_ERR_ARG;
  const HeadLayout L = head_layout(kind, *cp);
  if (L.bt.total > 0 && !bnbuf) return MMSA_ERR_ARG;
  HeadWs ws = head_ws(kind, *cp, ws_base);
  HCtx h{*cp, Eng{MMSA_F32, (hipStream_t)stream, ws.splitk, ws.splitk_bytes, ws.colws}, w, bnbuf, nullptr, 0, ws.bnws, ws.lnws,
         (unsigned long long)cp->seed};
  switch (kind) {
    case 0: return cross_fwd(h, L, ws, inputs, outputs);
    case 1: return mmf_fwd(h, L, ws, inputs, outputs);
    case 2: return wh_fwd(h, L, ws, inputs, outputs);
    case 3: return cls_fwd(h, L, ws, inputs, outputs);
    default: return proj_fwd(h, L, ws, inputs, outputs);
  }
}

int mmsa_head_bwd(int32_t kind, const mmsa_head_cfg* cp, const float* w, const float* const* inputs, const float* const* douts,
                  float* const* dinputs, float* grad, int32_t accumulate, void* ws_base, void* stream) {
  if (!cp || !head_cfg_ok(kind, *cp) || !w || !inputs || !douts || !dinputs || !ws_base) return MMSA_ERR_ARG;
  const HeadLayout L = head_layout(kind, *cp);
  HeadWs ws = head_ws(kind, *cp, ws_base);
  HCtx h{*cp, Eng{MMSA_F32, (hipStream_t)stream, ws.splitk, ws.splitk_bytes, ws.colws}, w, nullptr, grad, accumulate ? 1 : 0,
         ws.bnws, ws.lnws, (unsigned long long)cp->seed};
  switch (kind) {
    case 0: return cross_bwd(h, L, ws, inputs, douts, dinputs);
    case 1: return mmf_bwd(h, L, ws, inputs, douts, dinputs);
    case 2: return wh_bwd(h, L, ws, inputs, douts, dinputs);
    case 3: return cls_bwd(h, L, ws, inputs, douts, dinputs);
    default: return proj_bwd(h, L, ws, inputs, douts, dinputs);
  }
}

// fused cross-entropy forward + backward entry (Trainer.py:68 + the start of loss.backward())
int mmsa_ce_fwd_bwd(const float* logits, const int64_t* labels, float* loss, float* dlogits, float* probs, int32_t B, int32_t C,
                    float grad_scale, void* stream) {
  if (!logits || !labels || !loss || B <= 0 || C <= 0) return MMSA_ERR_ARG;
  return ce_fwd_bwd(logits, (const long long*)labels, loss, dlogits, probs, B, C, grad_scale, (hipStream_t)stream);
}

}  // extern "C"
