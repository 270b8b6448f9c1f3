// ResNet (bottleneck, v1.5) image encoder engine: whole forward / backward as kernel sequences on the caller's stream.
//
// NOT IN THE REFERENCE: ResNet-50 fills the reference's encoder slot (MultimodalModel.py:264-266). Activations are
// NHWC ([B*H*W][C] matrices); every convolution is an implicit GEMM on the MFMA kernel (1x1 stride 1: plain GEMM;
// 3x3 and strided 1x1: row gather at staging time; the 7x7 stem: explicit im2col, K padded 147 -> 192);
// BatchNorm uses batch statistics in training (two-stage column reductions) with the affine + ReLU (+ residual)
// applied in one streaming kernel. Weights are stored [Cout][KH][KW][Cin] (the physical layout of a channels_last
// torch tensor of logical shape [Cout][Cin][KH][KW], so state_dicts stay torchvision-compatible).
#include "../../include/mmsa.h"
#include <stdlib.h>
#include "engine_common.h"

struct ConvDef {
  int Cin, Cout, k, stride, pad, Hin, Win, Hout, Wout;
  long w, g, b;     // offsets into the flat parameter buffer: weight, bn gamma, bn beta
  long rm, rv;      // offsets into the flat BN buffer: running mean / var
};
struct BlockDef {
  ConvDef c1, c2, c3, ds;
  bool has_ds;
};
struct ResLayout {
  ParamTable t;   // parameters
  ParamTable bt;  // BN running buffers (fp32)
  ConvDef stem;
  std::vector<BlockDef> blocks;
  long wproj, bproj;
  int feat_c, Hf, Wf;  // final feature map
  int Kstem_pad;
};

static ConvDef make_conv(ResLayout& L, const std::string& wname, const std::string& bnname, int Cin, int Cout, int k, int s,
                         int p, int Hin, int Win) {
  ConvDef c;
  c.Cin = Cin; c.Cout = Cout; c.k = k; c.stride = s; c.pad = p; c.Hin = Hin; c.Win = Win;
  c.Hout = (Hin + 2 * p - k) / s + 1;
  c.Wout = (Win + 2 * p - k) / s + 1;
  c.w = L.t.add(wname, {Cout, Cin, k, k});  // logical OIHW; physical [Cout][k][k][Cin]
  c.g = L.t.add(bnname + ".weight", {Cout});
  c.b = L.t.add(bnname + ".bias", {Cout});
  c.rm = L.bt.add(bnname + ".running_mean", {Cout});
  c.rv = L.bt.add(bnname + ".running_var", {Cout});
  return c;
}

static ResLayout res_layout(const mmsa_resnet_cfg& c) {
  ResLayout L;
  L.Kstem_pad = 192;
  L.stem = make_conv(L, "resnet.conv1.weight", "resnet.bn1", 3, 64, 7, 2, 3, c.height, c.width);
  int H = (L.stem.Hout + 2 - 3) / 2 + 1, W = (L.stem.Wout + 2 - 3) / 2 + 1;  // after the 3x3/2 max-pool
  int inp = 64;
  for (int si = 0; si < 4; ++si) {
    const int w = c.widths[si];
    for (int b = 0; b < c.blocks[si]; ++b) {
      const std::string p = "resnet.layer" + std::to_string(si + 1) + "." + std::to_string(b) + ".";
      const int s = (b == 0 && si > 0) ? 2 : 1;
      BlockDef B;
      B.c1 = make_conv(L, p + "conv1.weight", p + "bn1", inp, w, 1, 1, 0, H, W);
      B.c2 = make_conv(L, p + "conv2.weight", p + "bn2", w, w, 3, s, 1, H, W);
      B.c3 = make_conv(L, p + "conv3.weight", p + "bn3", w, w * 4, 1, 1, 0, B.c2.Hout, B.c2.Wout);
      B.has_ds = (b == 0);
      if (B.has_ds) B.ds = make_conv(L, p + "downsample.0.weight", p + "downsample.1", inp, w * 4, 1, s, 0, H, W);
      L.blocks.push_back(B);
      H = B.c2.Hout; W = B.c2.Wout;
      inp = w * 4;
    }
  }
  L.feat_c = inp; L.Hf = H; L.Wf = W;
  L.wproj = L.t.add("proj.weight", {c.out_dim, inp});
  L.bproj = L.t.add("proj.bias", {c.out_dim});
  return L;
}

static bool res_cfg_ok(const mmsa_resnet_cfg& c) {
  if (c.batch <= 0 || c.height < 32 || c.width < 32 || c.out_dim % 64) return false;
  for (int i = 0; i < 4; ++i)
    if (c.blocks[i] <= 0 || c.widths[i] % 64) return false;
  return c.dtype == MMSA_F32 || c.dtype == MMSA_BF16;
}

struct ConvWs {
  void *z, *y;  // conv output (pre-BN), BN(+act) output
  float *mean, *invstd;
  unsigned char* mask;  // bn3 only: sign bits of the block output relu(bn3 + identity), [M][C/8] (bnops.hip)
};
struct BlockWs {
  ConvWs c1, c2, c3, ds;  // c3.y is the block output relu(bn3 + identity); ds.y the projected identity
};
struct ResWs {
  void *col, *stem_w;
  ConvWs stem;
  void* pool;
  unsigned char* pool_idx;
  std::vector<BlockWs> blocks;
  void *pooled, *dfeat_t, *dpooled;
  void *g0, *g1, *g2, *g3;  // gradient ping-pong buffers (largest activation size)
  void* foldw;  // inference: one convolution's weights with its BatchNorm scale folded in (largest weight)
  void* dzarena;  // backward: the pre-BatchNorm gradients dz of one stage's convolutions, kept until the stage's weight gradients
  size_t dzarena_bytes;  // go out as grouped launches (Eng::wgrad_batch)
  float *stem_dw, *splitk, *splitk2, *colws, *bnws;  // splitk2: slabs of the weight-gradient groups when they run on their own stream
  long bnws_floats;  // capacity of the partial-sum part of bnws (conv-epilogue statistics: GemmParams::colstat_cap)
  size_t splitk_bytes;
  size_t total;
};

static ConvWs conv_ws(Bump& b, const ConvDef& c, int B, size_t es, bool need_y = true, bool need_mask = false) {
  ConvWs w;
  const size_t n = (size_t)B * c.Hout * c.Wout * c.Cout;
  w.z = b.take(n * es);
  w.y = need_y ? b.take(n * es) : nullptr;
  w.mask = need_mask ? (unsigned char*)b.take(n / 8) : nullptr;
  w.mean = (float*)b.take((size_t)c.Cout * 4);
  w.invstd = (float*)b.take((size_t)c.Cout * 4);
  return w;
}

static ResWs res_ws(const mmsa_resnet_cfg& c, const ResLayout& L, void* base) {
  ResWs w;
  Bump b(base);
  const size_t es = c.dtype == MMSA_BF16 ? 2 : 4;
  const int B = c.batch;
  const size_t M0 = (size_t)B * L.stem.Hout * L.stem.Wout;
  w.col = b.take(M0 * L.Kstem_pad * es);
  w.stem_w = b.take((size_t)64 * L.Kstem_pad * es);
  w.stem = conv_ws(b, L.stem, B, es);
  const int PH = (L.stem.Hout + 2 - 3) / 2 + 1, PW = (L.stem.Wout + 2 - 3) / 2 + 1;
  const size_t np = (size_t)B * PH * PW * 64;
  w.pool = b.take(np * es);
  w.pool_idx = (unsigned char*)b.take(np);
  size_t maxact = M0 * 64;
  size_t maxw = (size_t)64 * L.Kstem_pad;
  int maxc = 64;
  for (const BlockDef& bd : L.blocks) {
    BlockWs x;
    x.c1 = conv_ws(b, bd.c1, B, es);
    x.c2 = conv_ws(b, bd.c2, B, es);
    x.c3 = conv_ws(b, bd.c3, B, es, true, true);
    if (bd.has_ds) x.ds = conv_ws(b, bd.ds, B, es);
    w.blocks.push_back(x);
    const size_t a_in = (size_t)B * bd.c1.Hin * bd.c1.Win * (bd.c1.Cin > bd.c1.Cout ? bd.c1.Cin : bd.c1.Cout);
    const size_t a_out = (size_t)B * bd.c3.Hout * bd.c3.Wout * bd.c3.Cout;
    if (a_in > maxact) maxact = a_in;
    if (a_out > maxact) maxact = a_out;
    const size_t w2 = (size_t)bd.c2.Cout * 9 * bd.c2.Cin, w3 = (size_t)bd.c3.Cout * bd.c3.Cin;
    if (w2 > maxw) maxw = w2;
    if (w3 > maxw) maxw = w3;
    if (bd.has_ds && (size_t)bd.ds.Cout * bd.ds.Cin > maxw) maxw = (size_t)bd.ds.Cout * bd.ds.Cin;
    if (bd.c3.Cout > maxc) maxc = bd.c3.Cout;
  }
  w.pooled = b.take((size_t)B * L.feat_c * es);
  w.dfeat_t = b.take((size_t)B * c.out_dim * es);
  w.dpooled = b.take((size_t)B * L.feat_c * es);
  w.g0 = b.take(maxact * es);
  w.g1 = b.take(maxact * es);
  w.g2 = b.take(maxact * es);
  w.g3 = b.take(maxact * es);
  {
    // sum of the convolution outputs of the whole net (a stage's weight gradients may still be reading their dz slots on the
    // weight-gradient stream while the next stage's backward fills its own: no slot is reused inside one backward; 1.4 GB of the
    // 288 at B = 64). `run` / `best` = the largest stage: the flush threshold when everything runs on one stream.
    size_t run = 0, best = 0, all = 0;
    for (int i = (int)L.blocks.size() - 1; i >= 0; --i) {
      const BlockDef& bd = L.blocks[i];
      const ConvDef* cs[4] = {&bd.c1, &bd.c2, &bd.c3, bd.has_ds ? &bd.ds : nullptr};
      for (const ConvDef* cd : cs)
        if (cd) { run += align_up((size_t)B * cd->Hout * cd->Wout * cd->Cout * es, 256); all += align_up((size_t)B * cd->Hout * cd->Wout * cd->Cout * es, 256); }
      if (run > best) best = run;
      if (bd.has_ds) run = 0;
    }
    (void)best;
    w.dzarena_bytes = all;
    w.dzarena = b.take(all);
  }
  w.stem_dw = (float*)b.take((size_t)64 * L.Kstem_pad * 4);
  w.foldw = b.take(maxw * es);
  // split-K slabs for the weight gradients: up to 64 slabs of the small early-stage weights, fewer of the large ones
  w.splitk_bytes = (size_t)16 * maxw * sizeof(float);
  w.splitk = (float*)b.take(w.splitk_bytes);
  w.splitk2 = (float*)b.take(w.splitk_bytes);
  w.colws = (float*)b.take(colsum_ws_bytes(maxc > c.out_dim ? maxc : c.out_dim));
  w.bnws = (float*)b.take(bn_ws_bytes(maxc));
  w.bnws_floats = (long)(bn_ws_bytes(maxc) / sizeof(float)) - 2L * maxc;  // the partial-sum part (the tail holds the backward's sums)
  w.total = b.off;
  return w;
}

struct ResCtx {
  const mmsa_resnet_cfg& c;
  Eng e;
  const float* w32;
  const void* wt;
  float* bnbuf;
  float* grad;
  int acc;
  size_t es;
  const void* W(long off) const { return (const char*)wt + (size_t)off * es; }
  const float* P(long off) const { return w32 + off; }
  float* G(long off) const { return grad + off; }
};

static void set_geom(ConvGeom& g, int SH, int SW, int GH, int GW, int k, int mul, int kmul, int off, int div, int cper) {
  g.SH = SH; g.SW = SW; g.GH = GH; g.GW = GW; g.KH = k; g.KW = k;
  g.mul = mul; g.kmul = kmul; g.off = off; g.offx = off; g.div = div; g.cper = cper; g.src_pix_stride = cper;
  g.fd_gw = make_fastdiv(GW); g.fd_ghw = make_fastdiv(GH * GW); g.fd_kw = make_fastdiv(k); g.fd_cper = make_fastdiv(cper);
}

// BatchNorm statistics taken by the convolution GEMM's epilogue (GemmParams::colstat): `rows` > 0 after a conv_fwd that produced
// them into `part` (training mode, bf16 MFMA path, no K split), 0 when the BatchNorm has to run its own statistics pass.
// MMSA_NO_CONV_STATS=1 turns the fusion off (A/B hook).
struct ConvStats {
  float* part;
  long cap;
  int rows;
};
static bool conv_stats_on() {
  return !mmsa_disabled("conv_stats");  // (read per call so that a test can compare both paths in one process)
}
// z[B*Ho*Wo][Cout] = conv(x)
// fold (inference): the convolution's weights with the BatchNorm scale folded in, its shift as the bias, activation, residual:
// y = act(conv(x, w') + shift (+ res))
struct ConvFold {
  const void* w;
  const float* shift;
  int act;
  const void* res;
};
static int conv_fwd(const ResCtx& r, const ConvDef& c, const void* x, void* z, ConvStats* cs = nullptr,
                    const ConvFold* fold = nullptr) {
  const int B = r.c.batch, M = B * c.Hout * c.Wout, K = c.k * c.k * c.Cin;
  GemmParams p = Eng::blank();
  p.A = x; p.lda = c.Cin; p.B = r.W(c.w); p.ldb = K; p.C = z; p.ldc = c.Cout;
  p.M = M; p.N = c.Cout; p.K = K;
  if (fold) {
    p.B = fold->w; p.bias = fold->shift; p.act = fold->act; p.add = fold->res; p.ldadd = c.Cout;
    p.act_after_add = fold->res ? 1 : 0;
  }
  if (cs) {
    cs->rows = 0;
    if (r.c.training == 1 && r.c.dtype == MMSA_BF16 && conv_stats_on()) { p.colstat = cs->part; p.colstat_cap = cs->cap; p.colstat_rows = &cs->rows; }
  }
  if (!(c.k == 1 && c.stride == 1)) {
    p.gather = 1;
    set_geom(p.g, c.Hin, c.Win, c.Hout, c.Wout, c.k, c.stride, 1, -c.pad, 1, c.Cin);
  }
  return r.e.gemm(p);
}
// Strided 1x1 data gradient (the downsample branch) as a dense GEMM over the Hout*Wout output-gradient rows whose epilogue
// scatters each row to its pixel (s*y, s*x) (GemmParams c_gw..c_colpitch): is that form available for this convolution?
static bool strided1x1_params(const ResCtx& r, const ConvDef& c, const void* dz, void* dx, GemmParams* out) {
  if (!(c.k == 1 && c.stride > 1 && c.pad == 0 && r.c.dtype == MMSA_BF16 && !Eng::force_simt() && !Eng::v1_only() &&
        c.Hin == c.Hout * c.stride && c.Win == c.Wout * c.stride))
    return false;
  GemmParams p = Eng::blank();
  p.A = dz; p.lda = c.Cout; p.B = r.W(c.w); p.ldb = c.Cin; p.b_kmajor = 1; p.C = dx; p.ldc = c.Cin;
  p.M = r.c.batch * c.Hout * c.Wout; p.N = c.Cin; p.K = c.Cout;
  p.c_gw = c.Wout; p.c_gh = c.Hout;
  p.c_imgpitch = (long)c.Hin * c.Win * c.Cin;
  p.c_rowpitch = (long)c.stride * c.Win * c.Cin;
  p.c_colpitch = (long)c.stride * c.Cin;
  p.fd_c_ghw = make_fastdiv((uint32_t)(c.Hout * c.Wout));
  p.fd_c_gw = make_fastdiv((uint32_t)c.Wout);
  if (!gemm2_eligible(p)) return false;
  *out = p;
  return true;
}
// ... ADDED to what dx already holds at those pixels (GemmParams::c_rmw): the block's main-path data gradient is written first, the
// projected-skip gradient lands on top of it — no zero fill of dx, no 3/4-zero side operand for the main path's GEMM.
// MMSA_ERR_UNSUPPORTED: the form does not exist for this convolution (the caller takes the zero-fill + side-operand order).
static int strided1x1_dgrad_onto(const ResCtx& r, const ConvDef& c, const void* dz, void* dx) {
  static const bool off = mmsa_disabled("ds_rmw");
  GemmParams p;
  if (off || !strided1x1_params(r, c, dz, dx, &p)) return MMSA_ERR_UNSUPPORTED;
  p.c_rmw = 1;
  return gemm_bf16_launch(p, r.e.st);
}
// dx[B*Hi*Wi][Cin] = conv_transpose(dz) (+ add)
static int conv_dgrad(const ResCtx& r, const ConvDef& c, const void* dz, void* dx, const void* add) {
  const int B = r.c.batch, M = B * c.Hin * c.Win, K = c.k * c.k * c.Cout;
  // Strided 1x1 on its own: only the pixels (s*y, s*x) receive gradient. As a row gather over all Hin*Win pixels 3/4 of the MFMA
  // rows multiply zeros; instead: zero-fill dx, then the dense GEMM with the scattering plain-store epilogue.
  if (!add) {
    GemmParams p;
    if (strided1x1_params(r, c, dz, dx, &p)) {
      if (hipMemsetAsync(dx, 0, (size_t)M * c.Cin * r.es, r.e.st) != hipSuccess) return MMSA_ERR_LAUNCH;
      return gemm_bf16_launch(p, r.e.st);
    }
  }
  // Strided 3x3 (stride 2, pad 1: conv2 of the first block of stages 2-4). As one row gather over all Hin*Win pixels,
  // 3/4 of the (pixel, tap) pairs are stride holes that multiply zeros. Decomposed by pixel parity (py, px) = (y % 2,
  // x % 2): the pixels of a class only see the taps with ky = py + 1 (mod 2), kx likewise — 1, 2, 2 and 4 taps — and
  // for those the source is a plain stride-1 gather of dY: sy = y' + py - ky' with y = 2y' + py and ky = 2ky' (py = 1)
  // or ky = 1 (py = 0). Four GEMMs over a quarter of the rows each, 9 taps in total, instead of 9 taps on every row:
  // 4x fewer MFMAs. Each class stores through the output row map to its own pixels; together they cover dx once.
  if (c.k == 3 && c.stride == 2 && c.pad == 1 && !add && r.c.dtype == MMSA_BF16 && !Eng::force_simt() && !Eng::v1_only() &&
      c.Hin == 2 * c.Hout && c.Win == 2 * c.Wout && !mmsa_disabled("parity_dgrad")) {
    GemmParams ps[4];
    bool ok = true;
    for (int cls = 0; cls < 4; ++cls) {
      const int py = cls >> 1, px = cls & 1;
      const int kh = py ? 2 : 1, kw = px ? 2 : 1, ky0 = py ? 0 : 1, kx0 = px ? 0 : 1;
      GemmParams& p = ps[cls];
      p = Eng::blank();
      p.A = dz; p.lda = c.Cout;
      p.B = (const char*)r.W(c.w) + (size_t)(ky0 * 3 + kx0) * c.Cin * r.es;
      p.ldb = 9L * c.Cin; p.b_kmajor = 1; p.b_tap_stride = 2L * c.Cin; p.b_tap_stride_y = 6L * c.Cin;
      p.C = (char*)dx + (size_t)(py * c.Win + px) * c.Cin * r.es; p.ldc = c.Cin;
      p.M = B * c.Hout * c.Wout; p.N = c.Cin; p.K = kh * kw * c.Cout;
      p.gather = 1;
      set_geom(p.g, c.Hout, c.Wout, c.Hout, c.Wout, 1, 1, -1, py, 1, c.Cout);
      p.g.KH = kh; p.g.KW = kw; p.g.offx = px; p.g.fd_kw = make_fastdiv((uint32_t)kw);
      p.c_gw = c.Wout; p.c_gh = c.Hout;
      p.c_imgpitch = (long)c.Hin * c.Win * c.Cin;
      p.c_rowpitch = 2L * c.Win * c.Cin;
      p.c_colpitch = 2L * c.Cin;
      p.fd_c_ghw = make_fastdiv((uint32_t)(c.Hout * c.Wout));
      p.fd_c_gw = make_fastdiv((uint32_t)c.Wout);
      ok = ok && gemm2_eligible(p);
    }
    if (ok) {
      for (int cls = 0; cls < 4; ++cls) {
        const int rc = gemm_bf16_launch(ps[cls], r.e.st);
        if (rc) return rc;
      }
      return MMSA_OK;
    }
  }
  GemmParams p = Eng::blank();
  p.A = dz; p.lda = c.Cout; p.B = r.W(c.w); p.ldb = (long)c.k * c.k * c.Cin; p.b_kmajor = 1; p.C = dx; p.ldc = c.Cin;
  p.M = M; p.N = c.Cin; p.K = K;
  p.add = add; p.ldadd = c.Cin;
  if (!(c.k == 1 && c.stride == 1)) {
    p.gather = 1;
    p.b_tap_stride = c.Cin;
    set_geom(p.g, c.Hout, c.Wout, c.Hin, c.Win, c.k, 1, -1, c.pad, c.stride, c.Cout);
  }
  return r.e.gemm(p);
}
// dW[Cout][k*k*Cin] (+)= dz^T gathered(x)
static GemmParams conv_wgrad_params(const ResCtx& r, const ConvDef& c, const void* dz, const void* x, float* dW, int accumulate) {
  const int B = r.c.batch, Kred = B * c.Hout * c.Wout, N = c.k * c.k * c.Cin;
  GemmParams p = Eng::blank();
  p.A = dz; p.lda = c.Cout; p.a_kmajor = 1; p.B = x; p.ldb = c.Cin; p.b_kmajor = 1; p.C = dW; p.ldc = N;
  p.M = c.Cout; p.N = N; p.K = Kred;
  p.out_f32 = 1; p.accumulate = accumulate;
  if (!(c.k == 1 && c.stride == 1)) {
    p.gather = 2;
    set_geom(p.g, c.Hin, c.Win, c.Hout, c.Wout, c.k, c.stride, 1, -c.pad, 1, c.Cin);
  }
  p.split_k = r.e.pick_split(c.Cout, N, Kred);
  p.ws = r.e.splitk_ws;
  return p;
}
static int conv_wgrad(const ResCtx& r, const ConvDef& c, const void* dz, const void* x, float* dW, int accumulate) {
  return r.e.gemm(conv_wgrad_params(r, c, dz, x, dW, accumulate));
}

// the ReLU sign bits of a block output (MMSA_NO_BN_MASK=1: the backward reads the saved output instead; A/B hook)
static unsigned char* bn_mask(const ConvWs& w, int act) {
  static const bool off = mmsa_disabled("bn_mask");
  return (off || act != MMSA_ACT_RELU) ? nullptr : w.mask;
}
static int bn_fwd(const ResCtx& r, const ConvDef& c, const ConvWs& w, const void* res, void* y, int act, float* bnws,
                  const ConvStats* cs = nullptr) {
  const int M = r.c.batch * c.Hout * c.Wout;
  return bn_forward(r.c.dtype, w.z, r.P(c.g), r.P(c.b), r.bnbuf + c.rm, r.bnbuf + c.rv, w.mean, w.invstd, res, y, bnws, M,
                    c.Cout, r.c.bn_eps, r.c.bn_momentum, act, r.c.training == 1, r.e.st, bn_mask(w, act),
                    cs && cs->rows > 0 ? cs->part : nullptr, cs ? cs->rows : 0);
}
// param_grads = false (a wholly frozen bottleneck): only the data gradient is produced, dgamma / dbeta are not written
static int bn_bwd(const ResCtx& r, const ConvDef& c, const ConvWs& w, const void* dy, const void* y, void* dz, void* dres,
                  int act, float* bnws, bool param_grads = true) {
  const int M = r.c.batch * c.Hout * c.Wout;
  return bn_backward(r.c.dtype, dy, w.z, y, w.mean, w.invstd, r.P(c.g), r.P(c.b), dz, dres, param_grads ? r.G(c.g) : nullptr,
                     param_grads ? r.G(c.b) : nullptr, r.acc, bnws, M, c.Cout, act, r.c.training == 1, r.e.st, bn_mask(w, act));
}

extern "C" {

int mmsa_resnet_param_count(const mmsa_resnet_cfg* c, int32_t buffers) {
  if (!c || !res_cfg_ok(*c)) return -1;
  const ResLayout L = res_layout(*c);
  return (int)(buffers ? L.bt.entries.size() : L.t.entries.size());
}
int64_t mmsa_resnet_param_total(const mmsa_resnet_cfg* c, int32_t buffers) {
  if (!c || !res_cfg_ok(*c)) return -1;
  const ResLayout L = res_layout(*c);
  return buffers ? L.bt.total : L.t.total;
}
int mmsa_resnet_param_info(const mmsa_resnet_cfg* c, int32_t buffers, int idx, char* name, int name_cap, int64_t* offset,
                           int32_t* ndim, int64_t* shape) {
  if (!c || !res_cfg_ok(*c)) return MMSA_ERR_ARG;
  const ResLayout L = res_layout(*c);
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
// Test/diagnostic aid: byte offset inside the workspace of a saved tensor. block = -1: stem (which 0 = conv out z,
// 1 = BN+ReLU out y, 2 = pooled); block >= 0: which 0..7 = c1.z, c1.y, c2.z, c2.y, c3.z, c3.y(block output), ds.z, ds.y;
// block = -2: which 0..3 = gradient ping-pong buffers g0..g3. Returns -1 when absent.
int64_t mmsa_resnet_ws_offset(const mmsa_resnet_cfg* c, int32_t block, int32_t which) {
  if (!c || !res_cfg_ok(*c)) return -1;
  const ResLayout L = res_layout(*c);
  char* base = (char*)0x100000;
  ResWs ws = res_ws(*c, L, base);
  const void* p = nullptr;
  if (block == -1) p = which == 0 ? ws.stem.z : which == 1 ? ws.stem.y : which == 2 ? ws.pool : nullptr;
  else if (block == -2) p = which == 0 ? ws.g0 : which == 1 ? ws.g1 : which == 2 ? ws.g2 : which == 3 ? ws.g3 : nullptr;
  else if (block >= 0 && block < (int)ws.blocks.size()) {
    const BlockWs& b = ws.blocks[block];
    const void* t[8] = {b.c1.z, b.c1.y, b.c2.z, b.c2.y, b.c3.z, b.c3.y, L.blocks[block].has_ds ? b.ds.z : nullptr,
                        L.blocks[block].has_ds ? b.ds.y : nullptr};
    if (which >= 0 && which < 8) p = t[which];
  }
  return p ? (int64_t)((const char*)p - base) : -1;
}

size_t mmsa_resnet_ws_bytes(const mmsa_resnet_cfg* c) {
  if (!c || !res_cfg_ok(*c)) return 0;
  const ResLayout L = res_layout(*c);
  return res_ws(*c, L, nullptr).total;
}

int mmsa_resnet_fwd(const mmsa_resnet_cfg* cp, const float* w32, const void* wt, float* bnbuf, const float* image,
                    void* ws_base, float* feat, void* stream) {
  if (!cp || !res_cfg_ok(*cp) || !w32 || !wt || !bnbuf || !image || !ws_base || !feat) return MMSA_ERR_ARG;
  const mmsa_resnet_cfg& c = *cp;
  const ResLayout L = res_layout(c);
  ResWs ws = res_ws(c, L, ws_base);
  hipStream_t st = (hipStream_t)stream;
  ResCtx r{c, Eng{c.dtype, st, ws.splitk, ws.splitk_bytes, ws.colws}, w32, wt, bnbuf, nullptr, 0, (size_t)(c.dtype == MMSA_BF16 ? 2 : 4)};
  const int B = c.batch;
  // stem: im2col (reads the NCHW fp32 image) + GEMM against the zero-padded weight copy
  const ConvDef& s = L.stem;
  const int M0 = B * s.Hout * s.Wout;
  RET_IF(stem_im2col(c.dtype, image, ws.col, B, 3, c.height, c.width, s.Hout, s.Wout, 7, 7, 2, 3, L.Kstem_pad, st));
  RET_IF(pad_rows(c.dtype, w32 + s.w, ws.stem_w, 64, 147, L.Kstem_pad, st));
  // every convolution's GEMM also sums the columns of the z it stores (per 64-row slice, into the BatchNorm scratch), so the
  // BatchNorm that follows starts at its finalize: one streamed pass over z less per convolution
  // training == 2 (inference: eval mode, forward only): every BatchNorm is FOLDED into the convolution that feeds it
  // (conv + BN + ReLU (+ residual) = one MFMA-tiled kernel, no BatchNorm launch, z never stored): per convolution two tiny
  // kernels turn the running statistics into scale / shift and write w' = w * scale[cout] (from the fp32 master weights) into a
  // scratch weight buffer; the GEMM runs on w' with the shift as its bias, the ReLU and the residual add in its epilogue.
  // MMSA_DISABLE=bn_fold: the unfolded eval path (A/B hook).
  const bool infer = c.training == 2 && !mmsa_disabled("bn_fold");
  auto fold_of = [&](const ConvDef& cd, const ConvWs& cw, int act, const void* res, ConvFold* f) -> int {
    const int K = cd.k * cd.k * cd.Cin;
    RET_IF(bn_fold(c.dtype, r.P(cd.g), r.P(cd.b), bnbuf + cd.rm, bnbuf + cd.rv, c.bn_eps, cd.Cout, cw.invstd, cw.mean,
                   w32 + cd.w, ws.foldw, K, K, st));
    f->w = ws.foldw; f->shift = cw.mean; f->act = act; f->res = res;
    return MMSA_OK;
  };
  if (infer) {
    ConvFold f;
    // the stem's weight copy is zero-padded to K = 192: fold into it in place of pad_rows' plain copy (the pad columns stay zero)
    RET_IF(bn_fold(c.dtype, r.P(s.g), r.P(s.b), bnbuf + s.rm, bnbuf + s.rv, c.bn_eps, 64, ws.stem.invstd, ws.stem.mean, w32 + s.w,
                   ws.stem_w, 147, L.Kstem_pad, st));
    GemmParams p = Eng::blank();
    p.A = ws.col; p.lda = L.Kstem_pad; p.B = ws.stem_w; p.ldb = L.Kstem_pad; p.C = ws.stem.y; p.ldc = 64;
    p.M = M0; p.N = 64; p.K = L.Kstem_pad;
    p.bias = ws.stem.mean; p.act = MMSA_ACT_RELU;
    RET_IF(r.e.gemm(p));
    RET_IF(maxpool_fwd(c.dtype, ws.stem.y, ws.pool, ws.pool_idx, B, s.Hout, s.Wout, 64, st));
    const void* x = ws.pool;
    for (size_t i = 0; i < L.blocks.size(); ++i) {
      const BlockDef& bd = L.blocks[i];
      BlockWs& bw = ws.blocks[i];
      const void* idn = x;
      if (bd.has_ds) {
        RET_IF(fold_of(bd.ds, bw.ds, MMSA_ACT_NONE, nullptr, &f));
        RET_IF(conv_fwd(r, bd.ds, x, bw.ds.y, nullptr, &f));
        idn = bw.ds.y;
      }
      RET_IF(fold_of(bd.c1, bw.c1, MMSA_ACT_RELU, nullptr, &f));
      RET_IF(conv_fwd(r, bd.c1, x, bw.c1.y, nullptr, &f));
      RET_IF(fold_of(bd.c2, bw.c2, MMSA_ACT_RELU, nullptr, &f));
      RET_IF(conv_fwd(r, bd.c2, bw.c1.y, bw.c2.y, nullptr, &f));
      RET_IF(fold_of(bd.c3, bw.c3, MMSA_ACT_RELU, idn, &f));
      RET_IF(conv_fwd(r, bd.c3, bw.c2.y, bw.c3.y, nullptr, &f));
      x = bw.c3.y;
    }
    RET_IF(avgpool_fwd(c.dtype, x, ws.pooled, B, L.Hf * L.Wf, L.feat_c, st));
    RET_IF(r.e.linear_fwd(ws.pooled, L.feat_c, r.W(L.wproj), r.P(L.bproj), feat, c.out_dim, B, c.out_dim, L.feat_c,
                          MMSA_ACT_NONE, nullptr, nullptr, 0, 1));
    return MMSA_OK;
  }
  ConvStats cs{ws.bnws, ws.bnws_floats, 0};
  {
    GemmParams p = Eng::blank();
    p.A = ws.col; p.lda = L.Kstem_pad; p.B = ws.stem_w; p.ldb = L.Kstem_pad; p.C = ws.stem.z; p.ldc = 64;
    p.M = M0; p.N = 64; p.K = L.Kstem_pad;
    if (c.training == 1 && c.dtype == MMSA_BF16 && conv_stats_on()) { p.colstat = cs.part; p.colstat_cap = cs.cap; p.colstat_rows = &cs.rows; }
    RET_IF(r.e.gemm(p));
  }
  RET_IF(bn_fwd(r, s, ws.stem, nullptr, ws.stem.y, MMSA_ACT_RELU, ws.bnws, &cs));
  RET_IF(maxpool_fwd(c.dtype, ws.stem.y, ws.pool, ws.pool_idx, B, s.Hout, s.Wout, 64, st));
  const void* x = ws.pool;
  for (size_t i = 0; i < L.blocks.size(); ++i) {
    const BlockDef& bd = L.blocks[i];
    BlockWs& bw = ws.blocks[i];
    const void* idn = x;
    if (bd.has_ds) {  // the projection shortcut first: its BatchNorm and conv3's share the statistics scratch
      RET_IF(conv_fwd(r, bd.ds, x, bw.ds.z, &cs));
      RET_IF(bn_fwd(r, bd.ds, bw.ds, nullptr, bw.ds.y, MMSA_ACT_NONE, ws.bnws, &cs));
      idn = bw.ds.y;
    }
    RET_IF(conv_fwd(r, bd.c1, x, bw.c1.z, &cs));
    RET_IF(bn_fwd(r, bd.c1, bw.c1, nullptr, bw.c1.y, MMSA_ACT_RELU, ws.bnws, &cs));
    RET_IF(conv_fwd(r, bd.c2, bw.c1.y, bw.c2.z, &cs));
    RET_IF(bn_fwd(r, bd.c2, bw.c2, nullptr, bw.c2.y, MMSA_ACT_RELU, ws.bnws, &cs));
    RET_IF(conv_fwd(r, bd.c3, bw.c2.y, bw.c3.z, &cs));
    RET_IF(bn_fwd(r, bd.c3, bw.c3, idn, bw.c3.y, MMSA_ACT_RELU, ws.bnws, &cs));
    x = bw.c3.y;
  }
  RET_IF(avgpool_fwd(c.dtype, x, ws.pooled, B, L.Hf * L.Wf, L.feat_c, st));
  RET_IF(r.e.linear_fwd(ws.pooled, L.feat_c, r.W(L.wproj), r.P(L.bproj), feat, c.out_dim, B, c.out_dim, L.feat_c,
                        MMSA_ACT_NONE, nullptr, nullptr, 0, 1));
  return MMSA_OK;
}

int mmsa_resnet_bwd(const mmsa_resnet_cfg* cp, const float* w32, const void* wt, void* ws_base, const float* dfeat, float* grad,
                    int32_t accumulate, void* stream) {
  return mmsa_resnet_bwd_cb2(cp, w32, wt, ws_base, dfeat, grad, accumulate, stream, nullptr, nullptr, nullptr, nullptr);
}

// The backward with a "gradient range ready" callback (see mmsa_bert_bwd_cb): the projection first, then one range per
// stage from stage 4 down (announced when the stage's first bottleneck — the last one the backward reaches — is done),
// the stem last.
int mmsa_resnet_bwd_cb(const mmsa_resnet_cfg* cp, const float* w32, const void* wt, void* ws_base, const float* dfeat,
                       float* grad, int32_t accumulate, void* stream, mmsa_range_cb cb, void* user, const uint8_t* frozen) {
  return mmsa_resnet_bwd_cb2(cp, w32, wt, ws_base, dfeat, grad, accumulate, stream, nullptr, cb, user, frozen);
}

// wgrad_stream (optional, another stream of the same device): the stage-wise weight-gradient groups are enqueued there —
// ordered after the stage's backward by an event, joined into `stream` by an event before the call returns (on EVERY exit path,
// errors included: the caller recycles the workspace) — so that these throughput-bound launches overlap the latency-bound
// BatchNorm / data-gradient chain of the following stages. With a range callback AND a weight-gradient stream (round 4: the
// data-parallel step is the single-GPU step) every announced range is complete on wgrad_stream — it waits for an event recorded on
// `stream` right before each announcement — so the caller records the range's event THERE.
int mmsa_resnet_bwd_cb2(const mmsa_resnet_cfg* cp, const float* w32, const void* wt, void* ws_base, const float* dfeat,
                        float* grad, int32_t accumulate, void* stream, void* wgrad_stream, mmsa_range_cb cb, void* user,
                        const uint8_t* frozen) {
  if (!cp || !res_cfg_ok(*cp) || !w32 || !wt || !ws_base || !dfeat || !grad) return MMSA_ERR_ARG;
  if (cp->training == 2) return MMSA_ERR_ARG;  // inference mode stored nothing a backward could use (include/mmsa.h)
  const mmsa_resnet_cfg& c = *cp;
  const ResLayout L = res_layout(c);
  ResWs ws = res_ws(c, L, ws_base);
  hipStream_t st = (hipStream_t)stream;
  ResCtx r{c, Eng{c.dtype, st, ws.splitk, ws.splitk_bytes, ws.colws}, w32, wt, nullptr, grad, accumulate ? 1 : 0,
           (size_t)(c.dtype == MMSA_BF16 ? 2 : 4)};
  const int B = c.batch, D = c.out_dim, acc = r.acc;
  // Frozen parameter groups (see mmsa_bert_bwd_cb): parameter-table entries are 3 for the stem (conv, BN weight, BN bias),
  // 9 (+3 with a downsample branch) per bottleneck, 2 for the projection. A wholly frozen bottleneck skips its weight-gradient
  // GEMMs and writes no BatchNorm gradients (its BatchNorm backward still runs for the data gradient); the backward stops below the lowest
  // trainable group.
  const int nb = (int)L.blocks.size();
  std::vector<int> first(nb + 1);
  {
    int idx = 3;
    for (int i = 0; i < nb; ++i) { first[i] = idx; idx += L.blocks[i].has_ds ? 12 : 9; }
    first[nb] = idx;
  }
  auto all_frozen = [&](int lo, int hi) {
    if (!frozen) return false;
    for (int i = lo; i < hi; ++i)
      if (!frozen[i]) return false;
    return true;
  };
  const bool stem_frozen = all_frozen(0, 3), tail_frozen = all_frozen(first[nb], first[nb] + 2);
  std::vector<char> blk_frozen(nb);
  int lowest = stem_frozen ? nb : -1;  // -1: stem trainable; i: bottleneck i is the lowest trainable group; nb: none below the tail
  for (int i = nb - 1; i >= 0; --i) {
    blk_frozen[i] = all_frozen(first[i], first[i + 1]);
    if (!blk_frozen[i] && stem_frozen) lowest = i;
  }
  // projection
  RET_IF(cast_f32(c.dtype, dfeat, ws.dfeat_t, (long)B * D, st));
  if (!tail_frozen) {
    RET_IF(r.e.bias_grad(ws.dfeat_t, D, r.G(L.bproj), B, D, acc));
    RET_IF(r.e.linear_wgrad(ws.dfeat_t, D, ws.pooled, L.feat_c, r.G(L.wproj), B, D, L.feat_c, acc));
  }
  // (announced below, once the weight-gradient stream is known: a range is announced on the stream it is complete on)
  const bool defer = c.dtype == MMSA_BF16 && !Eng::force_simt() && !Eng::v1_only() && !mmsa_disabled("wgrad_defer");
  // ast: the stream the announcements are complete on (the caller's wgrad_stream whenever one is passed, also in the modes that
  // launch every weight gradient in place); wst: where the deferred weight-gradient groups go
  hipStream_t ast = (wgrad_stream && wgrad_stream != stream) ? (hipStream_t)wgrad_stream : nullptr;
  hipStream_t wst = defer ? ast : nullptr;
  bool wst_used = false;
  auto wst_follows_st = [&]() -> int {  // what `stream` holds so far happens before what the weight-gradient stream does next
    if (!ast) return MMSA_OK;
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return MMSA_ERR_LAUNCH;
    const bool ok = hipEventRecord(ev, st) == hipSuccess && hipStreamWaitEvent(ast, ev, 0) == hipSuccess;
    (void)hipEventDestroy(ev);  // (released once the recorded work has completed)
    wst_used = true;
    return ok ? MMSA_OK : MMSA_ERR_LAUNCH;
  };
  auto join_wst = [&]() -> int {  // everything enqueued on the weight-gradient stream happens before what `stream` does next
    if (!wst_used) return MMSA_OK;
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return MMSA_ERR_LAUNCH;
    const bool ok = hipEventRecord(ev, ast) == hipSuccess && hipStreamWaitEvent(st, ev, 0) == hipSuccess;
    (void)hipEventDestroy(ev);
    wst_used = false;
    return ok ? MMSA_OK : MMSA_ERR_LAUNCH;
  };
  auto announce = [&](long off, long len) -> int {
    if (!cb) return MMSA_OK;
    RET_IF(wst_follows_st());
    cb(user, off, len);
    return MMSA_OK;
  };
  // everything below may have work in flight on the weight-gradient stream: it is joined on every exit path
  auto run = [&]() -> int {
  if (!tail_frozen) RET_IF(announce(L.wproj, L.t.total - L.wproj));
  if (lowest == nb) return MMSA_OK;  // stem and every bottleneck frozen
  RET_IF(r.e.linear_dgrad(ws.dfeat_t, D, r.W(L.wproj), ws.dpooled, L.feat_c, B, D, L.feat_c));
  void *dOut = ws.g0, *t1 = ws.g1, *t2 = ws.g2, *t3 = ws.g3;
  RET_IF(avgpool_bwd(c.dtype, ws.dpooled, dOut, B, L.Hf * L.Wf, L.feat_c, st));
  long chunk_end = L.wproj;
  bool chunk_live = false;
  // Weight gradients are deferred to the end of their stage and launched in groups (Eng::wgrad_batch): every dz then lives in its
  // own slot of the arena instead of the ping-pong buffer. MMSA_DISABLE=wgrad_defer (A/B hook) and the non-bf16 modes launch them in place.
  std::vector<GemmParams> pend;
  size_t arena_off = 0;
  Eng ew = r.e;  // the engine view of the weight-gradient stream: its own stream and its own slab workspace
  ew.st = wst;
  ew.splitk_ws = ws.splitk2;
  auto flush = [&]() -> int {
    int rc = MMSA_OK;
    if (!pend.empty()) {
      if (wst) {
        RET_IF(wst_follows_st());
        for (GemmParams& q : pend) q.ws = ws.splitk2;
        rc = ew.wgrad_batch(pend.data(), (int)pend.size());
      } else {
        rc = r.e.wgrad_batch(pend.data(), (int)pend.size());
      }
    }
    pend.clear();
    if (!wst) arena_off = 0;  // (with a weight-gradient stream no slot is reused: the arena spans the whole net)
    return rc;
  };
  // where the dz of convolution cd goes: its arena slot (deferred) or the ping-pong buffer `fallback`
  auto dz_slot = [&](const ConvDef& cd, void* fallback, bool wg, void** out) -> int {
    *out = fallback;
    if (!defer || !wg) return MMSA_OK;
    const size_t need = align_up((size_t)B * cd.Hout * cd.Wout * cd.Cout * r.es, 256);
    if (need > ws.dzarena_bytes) return MMSA_OK;
    if (arena_off + need > ws.dzarena_bytes) RET_IF(flush());
    *out = (char*)ws.dzarena + arena_off;
    arena_off += need;
    return MMSA_OK;
  };
  auto wgrad = [&](const ConvDef& cd, const void* dz, const void* x) -> int {
    if (defer && dz >= ws.dzarena && dz < (const void*)((const char*)ws.dzarena + ws.dzarena_bytes)) {
      pend.push_back(conv_wgrad_params(r, cd, dz, x, r.G(cd.w), acc));
      return MMSA_OK;
    }
    return conv_wgrad(r, cd, dz, x, r.G(cd.w), acc);
  };
  for (int i = (int)L.blocks.size() - 1; i >= 0 && i >= lowest; --i) {
    const BlockDef& bd = L.blocks[i];
    const bool wg = !blk_frozen[i], last_needed = (i == lowest);
    chunk_live = chunk_live || wg;
    BlockWs& bw = ws.blocks[i];
    const void* xin = i == 0 ? ws.pool : ws.blocks[i - 1].c3.y;
    // out = relu(bn3(z3) + idn): dz3 -> t1, masked gradient of the identity branch -> t2
    void *dz3, *dz2, *dz1, *dzd;  // (t1 / t3 unless the weight gradient is deferred: then a slot of the arena)
    RET_IF(dz_slot(bd.c3, t1, wg, &dz3));
    RET_IF(bn_bwd(r, bd.c3, bw.c3, dOut, bw.c3.y, dz3, t2, MMSA_ACT_RELU, ws.bnws, wg));
    if (wg) RET_IF(wgrad(bd.c3, dz3, bw.c2.y));
    RET_IF(conv_dgrad(r, bd.c3, dz3, t3, nullptr));                                   // dy2 -> t3
    RET_IF(dz_slot(bd.c2, t1, wg, &dz2));
    RET_IF(bn_bwd(r, bd.c2, bw.c2, t3, nullptr, dz2, nullptr, MMSA_ACT_RELU, ws.bnws, wg));  // dz2
    if (wg) RET_IF(wgrad(bd.c2, dz2, bw.c1.y));
    RET_IF(conv_dgrad(r, bd.c2, dz2, t3, nullptr));                                   // dy1 -> t3
    RET_IF(dz_slot(bd.c1, t1, wg, &dz1));
    RET_IF(bn_bwd(r, bd.c1, bw.c1, t3, nullptr, dz1, nullptr, MMSA_ACT_RELU, ws.bnws, wg));  // dz1
    if (const char* dbg = MMSA_EXP_ENV("MMSA_RESNET_BWD_STOP_BLOCK")) {  // diagnostic: leave t3 = dy1, t1 = dz1 of block i intact
      if (atoi(dbg) == i) {
        if (dz1 != t1 && hipMemcpyAsync(t1, dz1, (size_t)B * bd.c1.Hout * bd.c1.Wout * bd.c1.Cout * r.es, hipMemcpyDeviceToDevice, st) != hipSuccess)
          return MMSA_ERR_LAUNCH;
        return flush();
      }
    }
    if (wg) RET_IF(wgrad(bd.c1, dz1, xin));
    const void* skip = t2;  // identity block: the skip gradient is added to conv1's data gradient
    if (bd.has_ds) {
      RET_IF(dz_slot(bd.ds, t3, wg, &dzd));
      RET_IF(bn_bwd(r, bd.ds, bw.ds, t2, nullptr, dzd, nullptr, MMSA_ACT_NONE, ws.bnws, wg));  // dzd
      if (wg) RET_IF(wgrad(bd.ds, dzd, xin));
      if (!last_needed) {
        GemmParams probe;
        static const bool rmw_off = mmsa_disabled("ds_rmw");
        if (!rmw_off && strided1x1_params(r, bd.ds, dzd, t2, &probe)) {
          // main path first (plain store), then the strided projection's gradient added onto its pixels
          RET_IF(conv_dgrad(r, bd.c1, dz1, t2, nullptr));    // dx -> t2
          RET_IF(strided1x1_dgrad_onto(r, bd.ds, dzd, t2));
        } else {
          RET_IF(conv_dgrad(r, bd.ds, dzd, dOut, nullptr));  // dOut is free: holds the projected-skip data gradient
          skip = dOut;
          RET_IF(conv_dgrad(r, bd.c1, dz1, t2, skip));       // dx -> t2
        }
        void* t = dOut; dOut = t2; t2 = t;
      }
    } else if (!last_needed) {
      RET_IF(conv_dgrad(r, bd.c1, dz1, dOut, skip));     // dx -> dOut (its old content was consumed by bn3's backward)
    }
    if (bd.has_ds || last_needed) {  // first bottleneck of a stage (or the last one needed): the stage's gradients are enqueued
      RET_IF(flush());
      if (chunk_live) RET_IF(announce(bd.c1.w, chunk_end - bd.c1.w));
      chunk_end = bd.c1.w;
      chunk_live = false;
    }
  }
  RET_IF(flush());
  if (lowest >= 0) return MMSA_OK;  // the stem is frozen
  // max-pool, stem BN + ReLU, stem conv (weight gradient only: the image needs none)
  const ConvDef& s = L.stem;
  RET_IF(maxpool_bwd(c.dtype, dOut, ws.pool_idx, t1, B, s.Hout, s.Wout, 64, st));
  RET_IF(bn_bwd(r, s, ws.stem, t1, nullptr, t2, nullptr, MMSA_ACT_RELU, ws.bnws));
  const int M0 = B * s.Hout * s.Wout;
  {
    GemmParams p = Eng::blank();
    p.A = t2; p.lda = 64; p.a_kmajor = 1; p.B = ws.col; p.ldb = L.Kstem_pad; p.b_kmajor = 1; p.C = ws.stem_dw;
    p.ldc = L.Kstem_pad; p.M = 64; p.N = L.Kstem_pad; p.K = M0; p.out_f32 = 1;
    p.split_k = r.e.pick_split(64, L.Kstem_pad, M0);
    p.ws = ws.splitk;
    RET_IF(r.e.gemm(p));
  }
  RET_IF(unpad_rows(ws.stem_dw, r.G(s.w), 64, L.Kstem_pad, 147, acc, st));
  return announce(0, chunk_end);
  };
  const int rc = run();
  const int jrc = join_wst();
  return rc ? rc : jrc;
}

}  // extern "C"
