"""GPU parity: the C++ engines (BERT, ResNet, fusion-head modules, whole model) against the CPU oracle.

fp32 storage mode must match the fp32 oracle tightly (accumulation order only); bf16 mode is compared with the
oracle run under the bf16 storage policy (same rounding points) and, loosely, with the fp32 oracle.
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

import multimodal_sentiment_aanalysis_amd as mm
from multimodal_sentiment_aanalysis_amd import _lib
from multimodal_sentiment_aanalysis_amd.engine import BertTextNet, ResNetImageNet, materialize
from oracle import fusion as OF
from oracle import model as OM
from oracle.bert import bert_forward
from oracle.policy import BF16, BF16G, FP32, Policy
from oracle.resnet import resnet_forward
RESNET50_NARROW = dict(blocks=(3, 4, 6, 3), widths=(64, 64, 64, 64))  # ResNet-50 block structure, 64-wide stages

from util import MINI_BERT, MINI_RESNET, MINI_RESNET2, cpu_state, oracle_cfg, rel_err, synth_batch


def _grads(module):
    return {n: p.grad.detach().cpu().clone() for n, p in module.named_parameters() if p.grad is not None}


def _oracle_grads(sd, names, loss_fn):
    params = {n: sd[n].clone().requires_grad_(True) for n in names}
    work = dict(sd)
    work.update(params)
    loss = loss_fn(work)
    gs = torch.autograd.grad(loss, [params[n] for n in names], allow_unused=True)
    return {n: (g if g is not None else torch.zeros_like(sd[n])) for n, g in zip(names, gs)}


def _check_grads(got, ref, tol, what, skip=(), l2=False):
    """Per-tensor gradient error relative to that tensor's own scale, floored at 1e-3 of the largest gradient so that
    analytically-zero gradients (the key bias under softmax) are compared absolutely.
    l2=False: max-norm (smooth networks: BERT, fusion head).
    l2=True : relative L2 norm (ReLU networks): one pre-activation within rounding of 0 may take the other branch on
              the GPU (accumulation order) and moves a whole dy element between the two sides; that is a measure-zero
              discontinuity of the function, not an arithmetic error, and max-norm would report it as one."""
    gmax = max(r.abs().max().item() for r in ref.values())
    worst = ("", 0.0)
    for n, r in ref.items():
        if any(s in n for s in skip):
            continue
        assert n in got, f"{what}: no gradient produced for {n}"
        d = got[n].double() - r.double()
        if l2:
            scale = max(r.double().norm().item(), 1e-3 * gmax * (r.numel() ** 0.5), 1e-12)
            e = d.norm().item() / scale
        else:
            scale = max(r.abs().max().item(), 1e-3 * gmax, 1e-12)
            e = d.abs().max().item() / scale
        if e > worst[1]:
            worst = (n, e)
    assert worst[1] < tol, f"{what}: worst grad rel err {worst[1]:.3e} at {worst[0]} (tol {tol})"


@pytest.mark.parametrize("precision,pol,tol_f,tol_g", [("fp32", FP32, 2e-5, 2e-4), ("bf16", BF16G, 2e-2, 6e-2)])
# S = 128 is the benchmark sequence length (MFMA attention both ways); S = 256 is BASELINE.json configs[3]
# (MFMA attention both ways: the backward is the recompute variant that keeps no S x S image in LDS)
@pytest.mark.parametrize("B,S,masked", [(4, 32, False), (3, 16, True), (2, 64, True), (2, 128, True), (1, 256, False)])
def test_bert_engine(dev, precision, pol, tol_f, tol_g, B, S, masked):
    torch.manual_seed(0)
    cfg = MINI_BERT if S <= MINI_BERT["max_pos"] else dict(MINI_BERT, max_pos=256)
    net = BertTextNet(cfg)
    net.precision = precision
    sd = cpu_state(net)
    _, ids, mask, _ = synth_batch(B, S, 32, 32, cfg["vocab"], seed=5)
    if masked:
        mask[:, S - 5:] = 0
    wgt = torch.randn(B, 256, generator=torch.Generator().manual_seed(9))

    def feat_fn(work):
        _, pooled = bert_forward(work, "bert.", ids, mask if masked else None, cfg, pol)
        return pooled @ pol.qw(work["proj.weight"]).t() + work["proj.bias"]

    ref = feat_fn(sd)
    net.to(dev)
    out = net(ids.to(dev), mask.to(dev) if masked else None)
    assert rel_err(out, ref) < tol_f, f"bert feat rel err {rel_err(out, ref)}"
    (out * wgt.to(dev)).sum().backward()
    names = [n for n, _ in net.named_parameters()]
    ref_g = _oracle_grads(sd, names, lambda w: (feat_fn(w) * wgt).sum())
    _check_grads(_grads(net), ref_g, tol_g, f"bert {precision}")


@pytest.mark.parametrize("precision,pol,tol_f,tol_g", [("fp32", FP32, 5e-5, 1e-2), ("bf16", BF16G, 3e-2, 5e-1)])
@pytest.mark.parametrize("rcfg,B,HW", [(MINI_RESNET, 4, 96), (MINI_RESNET2, 3, 96), (MINI_RESNET, 2, 64),
                                       (RESNET50_NARROW, 2, 64)])
def test_resnet_engine(dev, precision, pol, tol_f, tol_g, rcfg, B, HW):
    torch.manual_seed(0)
    if sum(rcfg["blocks"]) > 8:
        # 16 bottleneck blocks = 53 train-mode BatchNorms over as few as 8 samples: each one re-amplifies the fp32 /
        # bf16 rounding of the one before. Measured 9e-5 (fp32 features); the north-star bound is 1e-3 on logits.
        tol_f, tol_g = 4 * tol_f, min(4 * tol_g, 0.9)
        if precision == "bf16":
            # 53 train-mode BatchNorms over 8-sample statistics: chaotic under bf16 storage (20 % feature distance between
            # two correct implementations). The full-depth bf16 checks are test_resnet_backward_teacher_forced (tight, at
            # the device's own forward) and test_resnet_engine_full_depth_bf16 (free-running, well conditioned: B = 16).
            pytest.skip("full-depth bf16: covered by test_resnet_backward_teacher_forced / test_resnet_engine_full_depth_bf16")
    net = ResNetImageNet(rcfg)
    net.precision = precision
    # non-trivial BN affine so its gradients are exercised
    with torch.no_grad():
        for n, p in net.named_parameters():
            if "bn" in n or "downsample.1" in n:
                p.add_(0.1 * torch.randn_like(p))
    sd = cpu_state(net)
    ocfg = dict(blocks=tuple(rcfg["blocks"]), widths=tuple(rcfg["widths"]), expansion=4)
    image, _, _, _ = synth_batch(B, 8, HW, HW, 10, seed=3)
    wgt = torch.randn(B, 256, generator=torch.Generator().manual_seed(9))

    def feat_fn(work):
        f = resnet_forward(work, "resnet.", image, ocfg, True, pol)
        return f @ pol.qw(work["proj.weight"]).t() + work["proj.bias"]

    sd_ref = {k: v.clone() for k, v in sd.items()}
    ref = feat_fn(sd_ref)
    net.to(dev)
    net.train()
    out = net(image.to(dev))
    assert rel_err(out, ref) < tol_f, f"resnet feat rel err {rel_err(out, ref)}"
    # running statistics follow torch's update rule (momentum 0.1, unbiased variance). Full depth: the same amplification
    # as tol_f above (variance of 8 samples in the last stage: 1.5e-4 with the batch-row GEMM kernel, 0.9e-4 with the VALU
    # one - two fp32 summation orders, both within rounding of the fp64 product)
    post = cpu_state(net)
    tol_s = (4e-4 if sum(rcfg["blocks"]) > 8 else 1e-4) if precision == "fp32" else 3e-2
    for k in sd_ref:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(post[k], sd_ref[k]) < tol_s, k
        if k.endswith("num_batches_tracked"):
            assert int(post[k]) == 1
    (out * wgt.to(dev)).sum().backward()
    names = [n for n, _ in net.named_parameters()]
    ref_g = _oracle_grads({k: v.clone() for k, v in sd.items()}, names, lambda w: (feat_fn(w) * wgt).sum())
    _check_grads(_grads(net), ref_g, tol_g, f"resnet {precision}", l2=True)


def _resnet_case(rcfg, B, HW, dev, precision="bf16", seed=3):
    torch.manual_seed(0)
    net = ResNetImageNet(rcfg)
    net.precision = precision
    with torch.no_grad():  # non-trivial BN affine so its gradients are exercised
        for n, p in net.named_parameters():
            if "bn" in n or "downsample.1" in n:
                p.add_(0.1 * torch.randn_like(p))
    sd = cpu_state(net)
    ocfg = dict(blocks=tuple(rcfg["blocks"]), widths=tuple(rcfg["widths"]), expansion=4)
    image, _, _, _ = synth_batch(B, 8, HW, HW, 10, seed=seed)
    wgt = torch.randn(B, 256, generator=torch.Generator().manual_seed(9))
    return net, sd, ocfg, image, wgt


def _saved_resnet_activations(out, ocfg, shapes):
    """The tensors the device forward stored in its workspace (bf16 NHWC), as fp32 NCHW CPU tensors keyed by the oracle's
    names (oracle/resnet.py). `out` is the engine's output (its grad_fn holds the workspace); `shapes`: name -> NCHW shape."""
    fn = out.grad_fn
    L = _lib.load()
    ws, cfg = fn.ws, fn.cfg
    which = {"c1.z": 0, "c1.y": 1, "c2.z": 2, "c2.y": 3, "c3.z": 4, "c3.y": 5, "ds.z": 6, "ds.y": 7}
    got = {}
    blk = 0
    index = {}
    for si, nb in enumerate(ocfg["blocks"]):
        for b in range(nb):
            index[f"layer{si + 1}.{b}"] = blk
            blk += 1
    for name, shp in shapes.items():
        if name in ("stem.z", "stem.y"):
            off = L.mmsa_resnet_ws_offset(ctypes.byref(cfg), -1, 0 if name == "stem.z" else 1)
        elif name.startswith("layer"):
            lay, w = name.rsplit(".", 2)[0], ".".join(name.rsplit(".", 2)[1:])
            off = L.mmsa_resnet_ws_offset(ctypes.byref(cfg), index[lay], which[w])
        else:
            continue
        assert off >= 0, name
        Bn, C, H, W = shp
        n = Bn * C * H * W
        t = ws[off:off + 2 * n].view(torch.bfloat16).view(Bn, H, W, C).permute(0, 3, 1, 2).float().cpu().contiguous()
        got[name] = t
    return got


@pytest.mark.parametrize("rcfg,B,HW,tol", [(MINI_RESNET, 4, 96, 2e-2), (RESNET50_NARROW, 16, 96, 4e-2),
                                           (dict(blocks=(3, 4, 6, 3), widths=(64, 128, 256, 512)), 16, 96, 4e-2),
                                           # J1 (BASELINE configs[3]): ResNet-101's block structure (a 23-bottleneck stage),
                                           # 64-wide; the random walk of stored-gradient roundings is 33 blocks long
                                           (dict(blocks=(3, 4, 23, 3), widths=(64, 64, 64, 64)), 16, 96, 6e-2)])
def test_resnet_backward_teacher_forced(dev, rcfg, B, HW, tol):
    """THE tight check of the bf16 (MFMA) ResNet backward, at full depth (53 BatchNorms; the last case is ResNet-50 itself).

    A free-running comparison of two bf16 ResNets is limited by chaos, not by arithmetic: 1-ulp bf16 flips of the stored
    activations decorrelate within a few layers, ~1 % of the ReLU decisions near zero then differ, and each such layer's
    gradient differs by sqrt(2 x 1 %) ~ 14 % whatever the kernels do (test_resnet_engine_full_depth_bf16 measures that
    end-to-end distance). So here the oracle's forward is FORCED to the values the device itself stored (read back from
    the engine workspace: every conv output z, every BN+ReLU output y, every block output): what is compared is the
    backward map at the same forward point — linear in the incoming gradient, no decisions left to differ on — with the
    oracle rounding every activation gradient to bf16 where the device stores one (policy BF16G). What remains is the
    independent bf16 rounding of ~3 stored gradient tensors per bottleneck on either side (~1.6e-3 each, accumulating as a
    random walk from the output to the stem): measured per-tensor relative L2 1.8e-3 at the projection growing smoothly
    to 1.0e-2 at the stem of the 4-block net, cosine > 0.9999 everywhere (tools/debug/tf_resnet.py prints the table).
    A wgrad / dgrad / BN-backward kernel that is wrong by a few per cent in any layer fails this.
    One tensor is checked at a looser bound: the stem's BN bias gradient is a badly conditioned sum (the next BatchNorms
    are invariant to most of a per-channel shift, so the true sum over 10^5 pixels nearly cancels: |g| is 10x smaller than
    its weight twin's) — measured 5-8e-2."""
    net, sd, ocfg, image, wgt = _resnet_case(rcfg, B, HW, dev)
    # shapes of every named activation from a dry oracle pass
    tr = {}
    with torch.no_grad():
        resnet_forward({k: v.clone() for k, v in sd.items()}, "resnet.", image, ocfg, True, Policy("bf16", trace=tr))
    shapes = {k: tuple(v.shape) for k, v in tr.items()}
    net.to(dev)
    net.train()
    out = net(image.to(dev))
    forced = _saved_resnet_activations(out, ocfg, shapes)
    assert len(forced) == len([k for k in shapes if k not in ("image", "pooled")])
    # the device forward itself: each stored tensor against the oracle applied to the stored tensor before it is a local
    # check of one conv / BN kernel; done here for the last one (pooled -> feature) through the output
    (out * wgt.to(dev)).sum().backward()
    pol = Policy("bf16", round_grads=True, forced=forced)

    def feat_fn(work):
        f = resnet_forward(work, "resnet.", image, ocfg, True, pol)
        return f @ pol.qw(work["proj.weight"]).t() + work["proj.bias"]

    names = [n for n, _ in net.named_parameters()]
    work0 = {k: v.clone() for k, v in sd.items()}
    ref_out = feat_fn(work0)
    assert rel_err(out, ref_out) < 5e-3, f"feature (from the forced last block output) {rel_err(out, ref_out)}"
    # per-layer forward check: every tensor the device stored vs the oracle's value of that layer computed from the
    # device's stored INPUTS of the layer (conv + bf16 store; BN with batch statistics + ReLU (+ residual) + store):
    # one bf16 rounding (2^-9, relative L2 ~1.1e-3) + fp32 accumulation order, for all 53 conv/BN pairs
    worst = max(pol.local_err.items(), key=lambda kv: kv[1])
    print(f"per-layer forward (device tensor vs oracle from the device's inputs): worst {worst[1]:.3e} at {worst[0]}, "
          f"{len(pol.local_err)} tensors")
    assert worst[1] < 4e-3, worst
    ref_g = _oracle_grads({k: v.clone() for k, v in sd.items()}, names, lambda w: (feat_fn(w) * wgt).sum())
    got = _grads(net)
    _check_grads(got, ref_g, tol, "resnet bf16 backward at the device's own forward", skip=("resnet.bn1.bias",), l2=True)
    _check_grads({"resnet.bn1.bias": got["resnet.bn1.bias"]}, {"resnet.bn1.bias": ref_g["resnet.bn1.bias"]}, 0.15,
                 "stem BN bias (ill-conditioned sum)", l2=True)


@pytest.mark.parametrize("gamma3", [1.0, 0.25])
def test_resnet_engine_full_depth_bf16(dev, gamma3):
    """The FREE-RUNNING end-to-end distance of the bf16 engine (replaces round 1's skip): ResNet-50 itself, B = 16, 96x96
    (BatchNorm statistics over 9216 ... 144 samples), against the oracle under the bf16 storage policy with gradients
    rounded at the same points (BF16G) and against the fp32 oracle — next to the distance between those two ORACLES.
    gamma3 = 1: default init; gamma3 = 0.25: every bottleneck's last BN weight scaled (damped residual branches, the
    regime of trained nets; tests/golden/make_c0_golden.py).

    What this shows, and asserts: under bf16 storage this graph is chaotic — the oracle's own bf16 policy is 31 % (default
    init) / ~2 % (damped) from its fp32 features and 150 % / ~50 % from its fp32 gradients at this size (1-ulp flips of
    stored activations decorrelate within a few layers; ~1 % of the ReLU decisions then differ and each flips a whole dy
    element). The device must be no further from fp32 than the bf16-policy oracle is (x1.5 + floor): its deviation is
    bf16 storage, not kernel arithmetic. The tight statements about the kernels are the teacher-forced test above and
    the fp32 mode."""
    rcfg = dict(blocks=(3, 4, 6, 3), widths=(64, 128, 256, 512))
    net, sd, ocfg, image, wgt = _resnet_case(rcfg, 16, 96, dev)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("bn3.weight"):
                p.mul_(gamma3)
    sd = cpu_state(net)

    def feat_fn(work, pol):
        f = resnet_forward(work, "resnet.", image, ocfg, True, pol)
        return f @ pol.qw(work["proj.weight"]).t() + work["proj.bias"]

    ref = feat_fn({k: v.clone() for k, v in sd.items()}, BF16G)
    ref32 = feat_fn({k: v.clone() for k, v in sd.items()}, FP32)
    net.to(dev)
    net.train()
    out = net(image.to(dev))
    e_f, e_f32, e_o = rel_err(out, ref), rel_err(out, ref32), rel_err(ref, ref32)
    (out * wgt.to(dev)).sum().backward()
    names = [n for n, _ in net.named_parameters()]
    got = _grads(net)
    rl2 = lambda a, b: (a.double() - b.double()).norm().item() / max(b.double().norm().item(), 1e-12)  # noqa: E731
    g_b = _oracle_grads({k: v.clone() for k, v in sd.items()}, names, lambda w: (feat_fn(w, BF16G) * wgt).sum())
    g_32 = _oracle_grads({k: v.clone() for k, v in sd.items()}, names, lambda w: (feat_fn(w, FP32) * wgt).sum())
    med = lambda xs: sorted(xs)[len(xs) // 2]  # noqa: E731
    d_b = [rl2(got[n], g_b[n]) for n in names]
    d_32 = [rl2(got[n], g_32[n]) for n in names]
    o_o = [rl2(g_b[n], g_32[n]) for n in names]
    print(f"full-depth bf16 ResNet-50 B=16 96x96 gamma3={gamma3}: features: device vs bf16g oracle {e_f:.3e}, device vs fp32 "
          f"oracle {e_f32:.3e}, bf16g oracle vs fp32 oracle {e_o:.3e}; gradient rel-L2 (median / worst over tensors): device "
          f"vs bf16g {med(d_b):.3e} / {max(d_b):.3e}, device vs fp32 {med(d_32):.3e} / {max(d_32):.3e}, bf16g oracle vs fp32 "
          f"oracle {med(o_o):.3e} / {max(o_o):.3e}")
    assert e_f32 < 1.5 * e_o + 2e-2, (e_f32, e_o)
    assert med(d_32) < 1.5 * med(o_o) + 5e-2 and max(d_32) < 1.5 * max(o_o) + 5e-2, (med(d_32), med(o_o), max(d_32), max(o_o))


def test_resnet_strided_dgrad_parity_classes(dev, monkeypatch):
    """The stride-2 3x3 data gradient as four parity-class GEMMs (resnet_engine.hip conv_dgrad) against the single row
    gather over all pixels (MMSA_DISABLE=parity_dgrad): same forward, same bf16 rounding points, only the fp32 summation
    order of a dx element differs, so every parameter gradient must agree to bf16 resolution."""
    image, _, _, _ = synth_batch(4, 8, 96, 96, 10, seed=3)
    wgt = torch.randn(4, 256, generator=torch.Generator().manual_seed(9)).to(dev)

    def run(no_parity):
        if no_parity:
            monkeypatch.setenv("MMSA_DISABLE", "parity_dgrad")
        else:
            monkeypatch.delenv("MMSA_DISABLE", raising=False)
        torch.manual_seed(0)
        net = ResNetImageNet(MINI_RESNET)
        net.precision = "bf16"
        net.to(dev)
        net.train()
        out = net(image.to(dev))
        L = _lib.load()
        L.mmsa_prof_mode(0)
        L.mmsa_prof_begin(4096)
        (out * wgt).sum().backward()
        torch.cuda.synchronize()
        ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        L.mmsa_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
        return out.detach().cpu(), _grads(net), n.value

    out_a, ga, na = run(False)
    out_b, gb, nb = run(True)
    assert torch.equal(out_a, out_b)
    assert na == nb + 9, f"three strided 3x3 convolutions x (4 class GEMMs instead of 1): {na} vs {nb} MFMA launches"
    _check_grads(ga, gb, 2e-2, "parity-class dgrad vs row gather", l2=True)


@pytest.mark.parametrize("rcfg,B,HW", [(MINI_RESNET, 4, 96), (dict(blocks=(3, 4, 6, 3), widths=(64, 128, 256, 512)), 4, 128)])
def test_resnet_strided_projection_gradient_added_onto_the_main_path(dev, monkeypatch, rcfg, B, HW):
    """The first block of stages 2-4: the strided 1x1 projection's data gradient is ADDED onto the main path's data gradient at its
    pixels by the GEMM's mapped read-modify-write epilogue (GemmParams::c_rmw; no zero fill, no 3/4-zero side operand) against the
    zero-fill + side-operand order (MMSA_DISABLE=ds_rmw): same forward, same sums with the bf16 rounding point on the other addend
    (round(acc_ds + bf16(acc_main)) instead of round(acc_main + bf16(acc_ds)) on a quarter of the pixels): the gradients agree to
    bf16 resolution; what no strided projection's backward precedes (the tail, the last stage's last block) is untouched."""

    def run(off):
        monkeypatch.setenv("MMSA_DISABLE", off)
        net, sd, ocfg, image, wgt = _resnet_case(rcfg, B, HW, dev)
        net.to(dev).train()
        out = net(image.to(dev))
        (out * wgt.to(dev)).sum().backward()
        return out.detach().float().cpu(), _grads(net)

    out_a, g_a = run("")
    out_b, g_b = run("ds_rmw")
    monkeypatch.delenv("MMSA_DISABLE", raising=False)
    assert torch.equal(out_a, out_b)
    _check_grads(g_a, g_b, 2e-2, "projection gradient added in place vs zero fill + side operand", l2=True)
    last = "resnet.layer4.%d." % (rcfg["blocks"][3] - 1)
    same = [n for n in g_a if n.startswith("proj.") or n.startswith(last)]
    assert same
    for n in same:
        assert torch.equal(g_a[n], g_b[n]), n


@pytest.mark.parametrize("rcfg,B,HW", [(MINI_RESNET, 4, 96), (dict(blocks=(3, 4, 6, 3), widths=(64, 128, 256, 512)), 8, 128)])
def test_resnet_weight_gradients_grouped_per_stage(dev, monkeypatch, rcfg, B, HW):
    """The weight gradients of a stage deferred to its end and launched in groups with one common K split (Eng::wgrad_batch ->
    mmsa_gemm_group_split's kernel) against launching each in place (MMSA_DISABLE=wgrad_defer): same forward, same dz values (they
    only live in another buffer), so every gradient agrees to the fp32 summation order of its K slices; fewer MFMA launches."""
    image, _, _, _ = synth_batch(B, 8, HW, HW, 10, seed=3)
    wgt = torch.randn(B, 256, generator=torch.Generator().manual_seed(9)).to(dev)

    def run(defer):
        monkeypatch.setenv("MMSA_DISABLE", "" if defer else "wgrad_defer")
        torch.manual_seed(0)
        net = ResNetImageNet(rcfg)
        net.precision = "bf16"
        net.to(dev)
        net.train()
        out = net(image.to(dev))
        L = _lib.load()
        L.mmsa_prof_mode(0)
        L.mmsa_prof_begin(4096)
        (out * wgt).sum().backward()
        torch.cuda.synchronize()
        ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        L.mmsa_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
        return out.detach().cpu(), _grads(net), n.value, fl.value

    out_a, ga, na, fa = run(True)
    out_b, gb, nb, fb = run(False)
    assert torch.equal(out_a, out_b)
    assert na < nb, f"grouped: {na} MFMA launches, in place: {nb}"
    # (a problem small enough for the batch-row kernel is outside the MFMA record set when launched alone, inside as a group member)
    assert abs(fa - fb) < 0.02 * fb, "the grouped launches account for the same algorithmic FLOPs"
    _check_grads(ga, gb, 2e-5, "grouped vs in-place weight gradients", l2=True)


@pytest.mark.parametrize("rcfg,B,HW", [(MINI_RESNET, 4, 96), (dict(blocks=(3, 4, 6, 3), widths=(64, 128, 256, 512)), 8, 128)])
def test_conv_epilogue_statistics_match_the_statistics_pass(dev, monkeypatch, rcfg, B, HW):
    """BatchNorm batch statistics taken in the convolution GEMM's epilogue (GemmParams::colstat: per 64-row slice column sums of
    the bf16 values the epilogue stores) against the separate streamed statistics pass over z (MMSA_DISABLE=conv_stats): the same
    numbers summed in another order — mean / variance agree to fp32 rounding, so the normalised activations are the same bf16
    values up to rare one-ulp flips, through all 53 BatchNorms of ResNet-50; fewer kernels are launched."""
    L = _lib.load()

    def run(fused):
        monkeypatch.setenv("MMSA_DISABLE", "" if fused else "conv_stats")
        net, sd, ocfg, image, wgt = _resnet_case(rcfg, B, HW, dev)
        net.to(dev).train()
        out = net(image.to(dev))
        torch.cuda.synchronize()
        post = {k: v.detach().float().cpu().clone() for k, v in net.state_dict().items() if "running" in k}
        (out * wgt.to(dev)).sum().backward()
        return out.detach().float().cpu(), post, _grads(net)

    out_a, st_a, g_a = run(True)
    out_b, st_b, g_b = run(False)
    deep = sum(rcfg["blocks"]) > 8
    # The statistics themselves are pinned where both runs see the same input: the stem (identical z: fp32 summation order only)
    # and the first bottleneck (its inputs differ by rare one-ulp flips of the stem's bf16 output). Further down the two runs are
    # two free-running bf16 forwards of a random-init BatchNorm net: they decorrelate like any two correct implementations do
    # (DESIGN.md section 4), so only a loose bound on the output remains; test_resnet_backward_teacher_forced and
    # test_resnet_engine_full_depth_bf16 run on the fused path and pin it against the oracle.
    for k in st_a:
        if k.startswith("resnet.bn1."):
            assert rel_err(st_a[k], st_b[k]) < 2e-6, k
        elif k.startswith("resnet.layer1.0."):
            assert rel_err(st_a[k], st_b[k]) < 5e-3, k
    assert rel_err(out_a, out_b) < (0.3 if deep else 3e-2), rel_err(out_a, out_b)
    assert all(torch.isfinite(g).all() for g in g_a.values())


@pytest.mark.parametrize("precision,tol", [("fp32", 5e-5), ("bf16", 3e-2)])
def test_resnet_inference_folds_batchnorm_into_the_convolutions(dev, monkeypatch, precision, tol):
    """Eval mode without a backward (torch.no_grad): mmsa_resnet_fwd runs with training = 2 — every BatchNorm folded into the GEMM
    epilogue of the convolution in front of it (conv + BN + ReLU (+ residual add) = one kernel; no BatchNorm launch, z never
    stored). Against the oracle's eval-mode forward, against the unfolded eval path of the same engine (MMSA_DISABLE=bn_fold), and
    with fewer kernel launches than it (GEMM launch count equal: the fold adds none)."""
    torch.manual_seed(1)
    rcfg = dict(blocks=(2, 1, 1, 2), widths=(64, 64, 128, 128))
    net = ResNetImageNet(rcfg)
    net.precision = precision
    with torch.no_grad():
        for n, b in net.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(0.1 * torch.randn_like(b))
            if n.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand_like(b))
        for n, p in net.named_parameters():
            if "bn" in n or "downsample.1" in n:
                p.add_(0.1 * torch.randn_like(p))
    sd = cpu_state(net)
    image, _, _, _ = synth_batch(4, 8, 96, 96, 10, seed=4)
    ocfg = dict(blocks=rcfg["blocks"], widths=rcfg["widths"], expansion=4)
    pol = FP32 if precision == "fp32" else BF16
    ref = resnet_forward(sd, "resnet.", image, ocfg, False, pol) @ pol.qw(sd["proj.weight"]).t() + sd["proj.bias"]
    net.to(dev).eval()
    with torch.no_grad():
        out = net(image.to(dev))
        monkeypatch.setenv("MMSA_DISABLE", "bn_fold")
        unfolded = net(image.to(dev))
        monkeypatch.setenv("MMSA_DISABLE", "")
    assert rel_err(out, ref) < tol, f"folded inference vs oracle ({precision}): {rel_err(out, ref)}"
    assert rel_err(out, unfolded) < tol, f"folded vs unfolded eval path: {rel_err(out, unfolded)}"
    # eval mode WITH a backward keeps the unfolded path (the backward needs z and the statistics) and still matches
    out_g = net(image.to(dev))
    assert rel_err(out_g, unfolded) < 1e-6
    (out_g * torch.ones_like(out_g)).sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_resnet_eval_mode(dev):
    torch.manual_seed(1)
    net = ResNetImageNet(MINI_RESNET)
    net.precision = "fp32"
    with torch.no_grad():
        for n, b in net.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(0.1 * torch.randn_like(b))
            if n.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand_like(b))
    sd = cpu_state(net)
    image, _, _, _ = synth_batch(2, 8, 64, 64, 10, seed=4)
    ocfg = dict(blocks=MINI_RESNET["blocks"], widths=MINI_RESNET["widths"], expansion=4)
    ref = resnet_forward(sd, "resnet.", image, ocfg, False, FP32) @ sd["proj.weight"].t() + sd["proj.bias"]
    net.to(dev).eval()
    with torch.no_grad():
        out = net(image.to(dev))
    assert rel_err(out, ref) < 5e-5


# ------------------------------------------------------------------------------------------------ fusion-head modules
def _feat(B, seed, dev=None):
    t = torch.randn(B, 256, generator=torch.Generator().manual_seed(seed))
    return t if dev is None else t.to(dev)


@pytest.mark.parametrize("B,Lk", [(16, 1), (5, 4), (64, 1), (37, 1), (7, 1)])
def test_cross_modal_transformer(dev, B, Lk):
    torch.manual_seed(0)
    m = mm.CrossModalTransformer()
    with torch.no_grad():
        m.multihead_attn.in_proj_bias.normal_(0, 0.1)
        m.norm.weight.add_(0.1 * torch.randn(256))
        m.norm.bias.add_(0.1 * torch.randn(256))
    sd = cpu_state(m)
    q = _feat(B, 1)
    g = torch.Generator().manual_seed(2)
    k = torch.randn(B, Lk, 256, generator=g) if Lk > 1 else _feat(B, 2)
    v = torch.randn(B, Lk, 256, generator=g) if Lk > 1 else _feat(B, 3)
    wgt = _feat(B, 4)
    ins = [t.clone().requires_grad_(True) for t in (q, k, v)]
    names = [n for n, _ in m.named_parameters()]
    params = {n: sd[n].clone().requires_grad_(True) for n in names}
    ref = OF.cross_modal_transformer(params, "", *ins) if False else None
    work = dict(sd); work.update(params)
    ref = OF.cross_modal_transformer({("x." + kk): vv for kk, vv in work.items()}, "x", *ins)
    gs = torch.autograd.grad((ref * wgt).sum(), ins + [params[n] for n in names])
    m.to(dev)
    dins = [t.to(dev).requires_grad_(True) for t in (q, k, v)]
    out = m(*dins)
    assert rel_err(out, ref) < 2e-5
    (out * wgt.to(dev)).sum().backward()
    for a, b, nm in zip(dins, gs[:3], "qkv"):
        assert rel_err(a.grad, b) < 2e-4, f"d{nm}"
    got = _grads(m)
    _check_grads(got, dict(zip(names, gs[3:])), 3e-4, "cross modal")


def _prefixed(sd, p):
    return {p + "." + k: v for k, v in sd.items()}


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("B", [16, 7])
def test_weighted_head(dev, train, B):
    torch.manual_seed(0)

    class Head(mm.MultimodalTransformerModel.__mro__[1]):  # HeadEngine with the weighted-head layout only
        kind = 2

        def __init__(self):
            super().__init__()
            self._init_head()

        def _base_cfg(self):
            c = super()._base_cfg()
            c.update(embed=256, num_classes=3, valence=1, dropout_p=0.0)
            return c

        def _out_dims(self):
            return [3, 128, 3]

    m = Head()
    m.train(train)
    with torch.no_grad():
        for n, b in m.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(0.1 * torch.randn_like(b))
            if n.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand_like(b))
    sd = cpu_state(m)
    feats = [_feat(B, 10 + i) for i in range(5)]
    wa, wv = torch.randn(B, 3, generator=torch.Generator().manual_seed(1)), torch.randn(B, 3, generator=torch.Generator().manual_seed(2))
    names = [n for n, _ in m.named_parameters()]
    params = {n: sd[n].clone().requires_grad_(True) for n in names}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(params)
    ins = [f.clone().requires_grad_(True) for f in feats]
    logits, fused = OF.weighted_fusion_logits(work, ins[0], ins[1], ins[2], ins[3], ins[4], train)
    val = OF.valence_head(work, "valence_head", fused, train)
    gs = torch.autograd.grad((logits * wa).sum() + (val * wv).sum(), ins + [params[n] for n in names])
    m.to(dev)
    dins = [f.to(dev).requires_grad_(True) for f in feats]
    out = m._run(*dins)
    assert rel_err(out[0], logits) < 5e-5 and rel_err(out[2], val) < 5e-5 and rel_err(out[1], fused) < 5e-5
    ((out[0] * wa.to(dev)).sum() + (out[2] * wv.to(dev)).sum()).backward()
    for i in range(5):
        assert rel_err(dins[i].grad, gs[i]) < 5e-4, f"dinput {i}"
    _check_grads(_grads(m), dict(zip(names, gs[5:])), 1e-3, "weighted head")
    if train:
        post = cpu_state(m)
        for k in work:
            if k.endswith("running_var") or k.endswith("running_mean"):
                assert rel_err(post[k], work[k]) < 1e-4, k


@pytest.mark.parametrize("B", [64, 37])
def test_head_units_one_launch_match_the_separate_launches(dev, monkeypatch, B):
    """head_fused.hip's Linear -> BatchNorm1d -> activation -> Dropout unit (one launch) and the unit backward up to its Linear
    (Dropout + BatchNorm [+ ReLU] backward in one launch) against the separate launches (MMSA_DISABLE=head_units), in training mode
    with Dropout on: same counter-based keep decisions (seed, element index), the Linear summed in the batch-row kernel's order,
    the statistics in bn_small_fwd_kernel's partition -> equal up to the contraction of a few fp32 expressions (1e-6)."""
    from multimodal_sentiment_aanalysis_amd.engine import HeadEngine

    class Head(mm.MultimodalTransformerModel.__mro__[1]):
        kind = 2

        def __init__(self):
            super().__init__()
            self._init_head()

        def _base_cfg(self):
            c = super()._base_cfg()
            c.update(embed=256, num_classes=3, valence=1, dropout_p=0.3)
            return c

        def _out_dims(self):
            return [3, 128, 3]

    res = []
    for off in ("", "head_units"):
        monkeypatch.setenv("MMSA_DISABLE", off)
        torch.manual_seed(0)
        HeadEngine._seed_counter = 0
        m = Head()
        m.train(True)
        m.to(dev)
        dins = [_feat(B, 10 + i, dev).requires_grad_(True) for i in range(5)]
        out = m._run(*dins)
        wa, wv, wf = (torch.randn(B, n, generator=torch.Generator().manual_seed(s)).to(dev) for n, s in ((3, 1), (3, 2), (128, 3)))
        ((out[0] * wa).sum() + (out[2] * wv).sum() + (out[1] * wf).sum()).backward()
        res.append(([o.detach() for o in out], [t.grad for t in dins], _grads(m), cpu_state(m)))
    monkeypatch.delenv("MMSA_DISABLE", raising=False)
    (o1, d1, g1, s1), (o2, d2, g2, s2) = res
    for a, b in zip(o1, o2):
        assert rel_err(a, b) < 1e-6
    assert (o1[1] == 0).float().mean().item() > 0.2  # the Dropout is on (fused features: 30 % zeros)
    assert torch.equal(o1[1] == 0, o2[1] == 0)
    for a, b in zip(d1, d2):
        assert rel_err(a, b) < 2e-6
    for n in g1:
        assert rel_err(g1[n], g2[n]) < 2e-6, n
    for k in s1:
        if k.endswith("running_var") or k.endswith("running_mean"):
            assert rel_err(s1[k], s2[k]) < 1e-6, k


@pytest.mark.parametrize("pool", ["max", "mean"])
def test_full_model_fp32(dev, pool):
    torch.manual_seed(0)
    m = mm.MultimodalTransformerModel(bert_config=MINI_BERT, resnet_config=MINI_RESNET, dropout=0.0)
    m.encoder.pool = pool
    sd = cpu_state(m)
    B = 8
    image, ids, mask, labels = synth_batch(B, 32, 64, 64, MINI_BERT["vocab"])
    cfg = oracle_cfg(MINI_BERT, MINI_RESNET, pool=pool)
    names = [n for n, p in m.named_parameters() if n not in ("contrastive_weight", "temperature")]

    def loss_fn(work):
        logits, _ = OM.model_forward(work, image, ids, mask, cfg, True, FP32)
        return OF.cross_entropy(logits, labels)

    sd_ref = {k: v.clone() for k, v in sd.items()}
    logits_ref, _ = OM.model_forward(sd_ref, image, ids, mask, cfg, True, FP32)
    loss_ref = OF.cross_entropy(logits_ref, labels)
    materialize(m, dev, "fp32")
    m.train()
    logits, aux = m(image.to(dev), ids.to(dev).float(), mask.to(dev), labels.to(dev))
    loss = mm.CrossEntropyLoss()(logits, labels.to(dev))
    assert (logits.cpu() - logits_ref).abs().max().item() < 1e-3, "logits tolerance 1e-3 (north_star)"
    assert abs(loss.item() - loss_ref.item()) < 1e-4, "loss tolerance 1e-4 (north_star)"
    loss.backward()
    ref_g = _oracle_grads({k: v.clone() for k, v in sd.items()}, names, loss_fn)
    _check_grads(_grads(m), ref_g, 1e-2, "full model fp32", l2=True)
    # eval contract: bare logits (Tester.py:53)
    m.eval()
    with torch.no_grad():
        ev = m(image.to(dev), ids.to(dev), mask.to(dev))
    post = cpu_state(m)
    ev_ref, _ = OM.model_forward(post, image, ids, mask, cfg, False, FP32)
    assert (ev.cpu() - ev_ref).abs().max().item() < 1e-3


def test_full_model_bf16(dev):
    torch.manual_seed(0)
    m = mm.MultimodalTransformerModel(bert_config=MINI_BERT, resnet_config=MINI_RESNET, dropout=0.0)
    sd = cpu_state(m)
    B = 8
    image, ids, mask, labels = synth_batch(B, 32, 64, 64, MINI_BERT["vocab"])
    cfg = oracle_cfg(MINI_BERT, MINI_RESNET)
    ref_bf, _ = OM.model_forward({k: v.clone() for k, v in sd.items()}, image, ids, mask, cfg, True, BF16)
    ref_32, _ = OM.model_forward({k: v.clone() for k, v in sd.items()}, image, ids, mask, cfg, True, FP32)
    materialize(m, dev, "bf16")
    m.train()
    logits, _ = m(image.to(dev), ids.to(dev), mask.to(dev), labels.to(dev))
    e_bf = (logits.cpu() - ref_bf).abs().max().item()
    e_32 = (logits.cpu() - ref_32).abs().max().item()
    print(f"bf16 path: |dlogits| vs bf16-policy oracle {e_bf:.3e}, vs fp32 oracle {e_32:.3e}")
    assert e_bf < 3e-2


def _run_engine_grads(make, run, env):
    import os
    old = os.environ.get("MMSA_BF16_SIMT")
    os.environ["MMSA_BF16_SIMT"] = env
    try:
        torch.manual_seed(0)
        net = make()
        out = run(net)
        torch.cuda.synchronize()
        return out.detach().cpu(), _grads(net)
    finally:
        if old is None:
            os.environ.pop("MMSA_BF16_SIMT", None)
        else:
            os.environ["MMSA_BF16_SIMT"] = old


def test_bf16_mfma_engines_match_simt_engines(dev):
    """The bf16 engines on the MFMA kernels vs the same engines forced onto the SIMT kernels (MMSA_BF16_SIMT=1): same
    storage rounding, independent GEMM / attention code. This is the tight check of the bf16 backward; the oracle
    comparison above is loose because the oracle does not round gradients to bf16 between layers."""
    image, ids, mask, _ = synth_batch(4, 32, 96, 96, MINI_BERT["vocab"], seed=11)
    wgt = torch.randn(4, 256, generator=torch.Generator().manual_seed(9)).to(dev)

    def make_r():
        n = ResNetImageNet(MINI_RESNET)
        n.precision = "bf16"
        return n.to(dev).train()

    def run_r(n):
        o = n(image.to(dev))
        (o * wgt).sum().backward()
        return o

    def make_b():
        n = BertTextNet(MINI_BERT)
        n.precision = "bf16"
        return n.to(dev)

    def run_b(n):
        o = n(ids.to(dev), mask.to(dev))
        (o * wgt).sum().backward()
        return o

    # ResNet tolerance: bf16 storage of activation gradients through train-mode BatchNorm is intrinsically noisy — the
    # BN backward subtracts the batch mean of a gradient whose entries are nearly equal (the broadcast of the
    # average-pool gradient), so a 2^-9 relative rounding of each entry becomes a 10-20 % relative change of the
    # difference. Measured: MFMA vs SIMT engines 9-24 %, either vs the fp32 engine 17-39 % ("bf16 backward").
    for nm, mk, rn, tol in (("resnet", make_r, run_r, 0.35), ("bert", make_b, run_b, 6e-2)):
        o1, g1 = _run_engine_grads(mk, rn, "0")
        o2, g2 = _run_engine_grads(mk, rn, "1")
        assert rel_err(o1, o2) < 2e-2, f"{nm} forward MFMA vs SIMT {rel_err(o1, o2)}"
        _check_grads(g1, g2, tol, f"{nm} bf16 MFMA vs SIMT", l2=True)


def test_full_size_resnet50_mfma_vs_simt_forward(dev):
    """ResNet-50 at its real widths (64..2048 channels, K up to 4608, 224x224: every tile shape, K split, the 147->192
    padded stem) on the MFMA kernels against the same engine on the SIMT kernels: same bf16 rounding points, independent
    GEMM code. Forward only: 53 train-mode BatchNorms make the bf16 backward of two correct implementations diverge
    (see test_bf16_mfma_engines_match_simt_engines); the backward at these widths is covered per convolution in
    test_gemm2_gpu.py::test_g2_conv and test_kernels_gpu.py::test_conv_implicit_gemm."""
    g = torch.Generator().manual_seed(21)
    image = torch.randn(4, 3, 224, 224, generator=g)

    def make():
        n = ResNetImageNet(None)  # default configuration = ResNet-50
        n.precision = "bf16"
        return n.to(dev).eval()  # eval: running statistics (identity at init) keep the comparison well conditioned

    def run(n):
        with torch.no_grad():
            return n(image.to(dev))

    outs = []
    for flag in ("0", "1"):
        import os
        os.environ["MMSA_BF16_SIMT"] = flag
        try:
            torch.manual_seed(0)
            outs.append(run(make()).float().cpu())
        finally:
            os.environ.pop("MMSA_BF16_SIMT", None)
    assert torch.isfinite(outs[0]).all()
    assert rel_err(outs[0], outs[1]) < 2e-2, f"ResNet-50 features MFMA vs SIMT {rel_err(outs[0], outs[1])}"


@pytest.mark.parametrize("B", [16, 64])
def test_full_size_train_step_properties(dev, B):
    """The benchmark's model (BERT-base + ResNet-50 + fusion head, bf16) at B = 16 and at B = 64 (BASELINE.json configs[1]'s
    own per-GPU batch: every planner decision, tile shape and K split of the benchmarked step): properties that need no oracle.
    (1) the loss of a fixed batch goes down over optimizer steps; (2) a second run from the same seed reproduces the
    first loss exactly (the forward has no atomics) and the later ones closely (every reduction is order-fixed except
    the word-embedding scatter, whose fp32 atomics reorder the sums of repeated tokens)."""
    from multimodal_sentiment_aanalysis_amd.fused import FusedTrainStep
    image, ids, mask, labels = synth_batch(B, 128, 224, 224, 30522, seed=5)
    batch = (image.to(dev), ids.to(dev), mask.to(dev), labels.to(dev))

    def run():
        torch.manual_seed(0)
        model = mm.MultimodalTransformerModel(dropout=0.0)
        step = FusedTrainStep(model, dev, precision="bf16", lr=1e-4)
        losses = []
        for _ in range(4):
            loss, _ = step.step(*batch)
            losses.append(float(loss))
        torch.cuda.synchronize()
        return losses

    l1 = run()
    l2 = run()
    assert all(v == v and abs(v) < 1e4 for v in l1), l1
    assert l1[-1] < l1[0], f"loss did not decrease on a fixed batch: {l1}"
    assert l1[0] == l2[0], (l1, l2)
    for a, b in zip(l1, l2):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (l1, l2)


@pytest.mark.parametrize("full", [False, True])
def test_image_encoder_on_its_own_stream_changes_nothing(dev, monkeypatch, full):
    """FusedTrainStep runs the image encoder on a side HIP stream beside the text encoder (two_streams, the default). The two
    encoders touch disjoint gradient ranges, workspaces and statistics buffers and meet only at the fusion head and at the
    optimizer, both behind event joins: losses, logits, every updated weight, both AdamW moments and the BatchNorm running
    statistics must be BIT-identical to the single-stream run — any missing ordering edge shows up as a difference (the
    word-embedding scatter's fp32 atomics are inside one encoder on one stream in both runs; a batch without repeated tokens
    keeps even them exact). Mini configuration, and the benchmark's own model at B = 16."""
    from multimodal_sentiment_aanalysis_amd.fused import FusedTrainStep
    if full:
        image, ids, mask, labels = synth_batch(16, 128, 224, 224, 30522, seed=5)
        mk = lambda: mm.MultimodalTransformerModel(dropout=0.0)  # noqa: E731
    else:
        image, ids, mask, labels = synth_batch(8, 32, 64, 64, MINI_BERT["vocab"], seed=3)
        mk = lambda: mm.MultimodalTransformerModel(bert_config=MINI_BERT, resnet_config=MINI_RESNET, dropout=0.0)  # noqa: E731
    ids = torch.stack([torch.randperm(int(ids.max()) + 1, generator=torch.Generator().manual_seed(40 + r))[:ids.shape[1]]
                       for r in range(ids.shape[0])]).to(ids.dtype)  # no token twice in a row: no atomic meets another
    batch = (image.to(dev), ids.to(dev), mask.to(dev), labels.to(dev))

    def run(two, wgrad_stream=False):
        monkeypatch.setenv("MMSA_WGRAD_STREAM", "1" if wgrad_stream else "0")
        torch.manual_seed(0)
        step = FusedTrainStep(mk(), dev, precision="bf16", lr=1e-3, two_streams=two)
        assert step.two_streams == two and (getattr(step._image_net, "_side", None) is not None) == two
        assert (getattr(step._image_net, "_wgrad_stream", None) is not None) == wgrad_stream
        out = []
        for _ in range(3):
            loss, logits = step.step(*batch)
            out.append((loss.clone(), logits.clone()))
        torch.cuda.synchronize()
        st = step.state
        return out, st.flat_w.clone(), step.opt.m.clone(), step.opt.v.clone(), st.flat_bn.clone()

    a, b = run(True), run(False)
    # ... and with the image encoder's stage-wise weight-gradient groups on a third stream (mmsa_resnet_bwd_cb2)
    c = run(True, wgrad_stream=True)
    for other, name in ((a, "two-stream"), (c, "three-stream")):
        for (la, ga), (lb, gb) in zip(other[0], b[0]):
            assert torch.equal(la, lb) and torch.equal(ga, gb), name
        for x, y, what in zip(other[1:], b[1:], ("weights", "exp_avg", "exp_avg_sq", "BatchNorm buffers")):
            assert torch.equal(x, y), f"{what} differ between the {name} and the single-stream step"


def test_c3_bert_large_resnet101_train_steps(dev):
    """BASELINE.json configs[3] (BERT-large S = 256 + ResNet-101): the largest configuration runs through the same engines
    (attention backward at S = 256: the recompute MFMA kernel) at its stated per-GPU batch of 32 — a fixed batch's loss is
    finite and goes down."""
    from multimodal_sentiment_aanalysis_amd.fused import FusedTrainStep
    image, ids, mask, labels = synth_batch(32, 256, 224, 224, 30522, seed=9)
    batch = (image.to(dev), ids.to(dev), mask.to(dev), labels.to(dev))
    torch.manual_seed(0)
    model = mm.MultimodalTransformerModel(bert_config=mm.BERT_LARGE, resnet_config=mm.RESNET101, dropout=0.0)
    step = FusedTrainStep(model, dev, precision="bf16", lr=1e-4)
    losses = [float(step.step(*batch)[0]) for _ in range(4)]
    torch.cuda.synchronize()
    assert all(v == v and abs(v) < 1e4 for v in losses), losses
    assert losses[-1] < losses[0], losses
    del step, model
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------ step tail rules
def _mini_step(dev, precision="bf16"):
    from multimodal_sentiment_aanalysis_amd.fused import FusedTrainStep
    torch.manual_seed(0)
    model = mm.MultimodalTransformerModel(bert_config=MINI_BERT, resnet_config=MINI_RESNET, dropout=0.0)
    step = FusedTrainStep(model, dev, precision=precision)
    image, ids, mask, labels = synth_batch(8, 32, 64, 64, MINI_BERT["vocab"], seed=3)
    return step, (image.to(dev), ids.to(dev), mask.to(dev), labels.to(dev))


def test_fused_step_skips_non_finite_batch(dev):
    """Trainer.py:74-76: a NaN loss skips the update. The fused step decides on the device (no host sync): a batch whose
    loss / gradient norm is not finite must leave the weights, both AdamW moments, the bf16 working copy and the
    bias-correction step count exactly as they were, and the next good batch must train on as if nothing happened."""
    step, batch = _mini_step(dev)
    st, opt = step.state, step.opt
    loss0, _ = step.step(*batch)
    assert torch.isfinite(loss0) and opt.t == 1
    snap = [t.clone() for t in (st.flat_w, st.flat_wt, opt.m, opt.v)]
    # (a) an inf pixel: the ReLU of the image encoder squashes the NaN channel in the forward (v > 0 ? v : 0), so the loss
    #     stays finite, but the BatchNorm backward of that channel is NaN -> the gradient norm is not finite;
    # (b) a NaN loss (Trainer.py:74 `torch.isnan(loss)`): the last bias of the arousal head is poisoned for one step
    eq = lambda a, b: torch.equal(a.nan_to_num(7.0), b.nan_to_num(7.0))  # noqa: E731  (NaN-tolerant bitwise compare)
    bad_img = batch[0].clone()
    bad_img[1, 0, 5, 7] = float("inf")
    loss_bad, _ = step.step(bad_img, *batch[1:])
    assert not torch.isfinite(opt.norm_out[0]) and float(opt.norm_out[1]) == -1.0, "inf pixel: skip flag"
    for a, b, tn in zip((st.flat_w, st.flat_wt, opt.m, opt.v), snap, ("w", "bf16 copy", "m", "v")):
        assert eq(a, b), f"inf pixel: {tn} changed on a skipped step"
    assert opt.t == 1
    bias = [q for n, q in step.model.named_parameters() if n.startswith("arousal_head") and n.endswith("bias")][-1]
    with torch.no_grad():
        keep = bias.detach().clone()
        bias.fill_(float("nan"))
    snap_nan = st.flat_w.clone()
    loss_bad, _ = step.step(*batch)
    assert torch.isnan(loss_bad) and float(opt.norm_out[1]) == -1.0, "NaN loss: skip flag"
    for a, b, tn in zip((st.flat_w, st.flat_wt, opt.m, opt.v), [snap_nan] + snap[1:], ("w", "bf16 copy", "m", "v")):
        assert eq(a, b), f"NaN loss: {tn} changed on a skipped step"
    assert opt.t == 1, "bias-correction step count advanced on a skipped step"
    with torch.no_grad():
        bias.copy_(keep)
    assert eq(st.flat_w, snap[0])
    loss1, _ = step.step(*batch)
    assert torch.isfinite(loss1) and opt.t == 2
    assert torch.isfinite(st.flat_w).all() and not torch.equal(st.flat_w, snap[0])
    # the same two good steps without the bad batch in between give the same weights (BN running buffers aside, which the
    # forward of the skipped batch did touch — as the reference's forward does before its NaN check)
    step_b, _ = _mini_step(dev)
    step_b.step(*batch)
    step_b.step(*batch)
    assert float(loss1) == pytest.approx(float(step_b.loss), rel=1e-4)


def test_flat_adamw_over_sub_ranges(dev):
    """FlatAdamW(ranges=...) — a curriculum phase's optimizer (MultiTaskTrainer.py:50-177) — takes the clip norm over its
    ranges only and updates nothing outside them; against torch.optim.AdamW + clip_grad_norm_ over the same parameters."""
    from multimodal_sentiment_aanalysis_amd.fused import FlatAdamW
    torch.manual_seed(0)
    model = mm.MultimodalTransformerModel(bert_config=MINI_BERT, resnet_config=MINI_RESNET, dropout=0.0, multitask=True)
    state = materialize(model, dev, "fp32")
    state.flat_g.copy_(torch.randn_like(state.flat_g) * 0.01)
    # the valence head's parameters (phase 3 of the reference optimizes exactly these)
    sub = [(n, p) for n, p in model.named_parameters() if n.startswith("valence_head.")]
    base = state.flat_w.data_ptr()
    ranges = [((p.data_ptr() - base) // 4, p.numel()) for _, p in sub]
    ref_p = [torch.nn.Parameter(p.detach().clone()) for _, p in sub]
    for rp, (_, p) in zip(ref_p, sub):
        rp.grad = p.grad.detach().clone()
    ref_opt = torch.optim.AdamW(ref_p, lr=1e-3, weight_decay=1e-4)
    w_before = state.flat_w.clone()
    opt = FlatAdamW(state, lr=1e-3, weight_decay=1e-4, max_norm=1.0, ranges=ranges)
    for _ in range(3):
        total = torch.nn.utils.clip_grad_norm_(ref_p, 1.0)
        ref_opt.step()
        for rp, (_, p) in zip(ref_p, sub):  # clip_grad_norm_ scaled the reference's grads in place: restore for the next round
            rp.grad = p.grad.detach().clone()
        opt.step()
        assert float(opt.norm_out[0]) == pytest.approx(float(total), rel=1e-5)
    for rp, (n, p) in zip(ref_p, sub):
        assert rel_err(p, rp) < 1e-5, n
    touched = torch.zeros_like(state.flat_w, dtype=torch.bool)
    for a, n in opt.ranges:
        touched[a:a + n] = True
    assert torch.equal(state.flat_w[~touched], w_before[~touched]), "parameters outside the optimizer's ranges moved"


def test_working_copy_follows_any_parameter_edit(dev):
    """The bf16 working copy must be refreshed when ANY parameter of an engine changes (an optimizer over a few layers,
    a partial load_state_dict, an in-place edit): round 1 watched three tensors only."""
    torch.manual_seed(0)
    net = BertTextNet(MINI_BERT)
    net.precision = "bf16"
    net.to(dev)
    _, ids, mask, _ = synth_batch(2, 32, 32, 32, MINI_BERT["vocab"], seed=5)
    with torch.no_grad():
        a = net(ids.to(dev), mask.to(dev)).clone()
        p = dict(net.named_parameters())["bert.encoder.layer.1.attention.output.dense.weight"]  # neither first, middle nor last
        p.mul_(1.5)
        b = net(ids.to(dev), mask.to(dev)).clone()
    assert not torch.equal(a, b), "stale bf16 working copy after an in-place edit of one mid-layer tensor"


@pytest.mark.parametrize("which", ["bert", "resnet"])
def test_frozen_groups_skip_kernels_and_keep_gradients(dev, which):
    """N2 inside an encoder: with the lower layers (BERT: embeddings + layer 0; ResNet: stem + the first two bottlenecks) frozen,
    the backward launches fewer matrix-core GEMMs (no weight gradients for frozen groups, no data gradient below the lowest
    trainable one), the trainable parameters get EXACTLY the gradients of the all-trainable run, and frozen ones get none."""
    image, ids, mask, _ = synth_batch(4, 32, 96, 96, MINI_BERT["vocab"], seed=11)
    wgt = torch.randn(4, 256, generator=torch.Generator().manual_seed(9)).to(dev)
    L = _lib.load()

    def run(freeze):
        torch.manual_seed(0)
        if which == "bert":
            net = BertTextNet(dict(MINI_BERT, layers=3))
            frozen = lambda n: n.startswith("bert.embeddings.") or n.startswith("bert.encoder.layer.0.")  # noqa: E731
            fwd = lambda: net(ids.to(dev), mask.to(dev))  # noqa: E731
        else:
            net = ResNetImageNet(MINI_RESNET2)
            frozen = lambda n: n.startswith(("resnet.conv1.", "resnet.bn1.", "resnet.layer1.0.", "resnet.layer1.1."))  # noqa: E731
            fwd = lambda: net(image.to(dev))  # noqa: E731
        net.precision = "bf16"
        net.to(dev)
        net.train()
        if freeze:
            for n, p in net.named_parameters():
                if frozen(n):
                    p.requires_grad = False
        out = fwd()
        L.mmsa_prof_mode(0)
        L.mmsa_prof_begin(4096)
        (out * wgt).sum().backward()
        torch.cuda.synchronize()
        ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        L.mmsa_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
        return {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in net.named_parameters()}, n.value, frozen

    g_all, n_all, frozen = run(False)
    g_fr, n_fr, _ = run(True)
    assert n_fr < n_all, f"{which}: {n_fr} GEMM launches with frozen groups vs {n_all}"
    for k, g in g_all.items():
        if frozen(k):
            assert g_fr[k] is None, k
        else:
            assert g_fr[k] is not None and torch.equal(g_fr[k], g), f"{which}: gradient of trainable {k} changed"



def test_bert_engine_fp8(dev):
    """BASELINE.json configs[4] at the engine level: the text encoder with fp8 (e4m3) operands in its forward Linears (precision
    "fp8": bf16 storage, per-tensor current scaling, bf16 backward) against the ORACLE under the same policy (oracle/policy.py
    BF16G_FP8: the same scales, the same e4m3 rounding, exact products, straight-through bf16 backward). The two quantize
    identical tensors identically, so they differ like two bf16 implementations do (~1e-2), not by e4m3's 2^-4 steps; how far fp8
    moves the feature from bf16 (the price of the format) is printed beside it."""
    from oracle.policy import BF16G_FP8
    torch.manual_seed(0)
    cfg = dict(MINI_BERT, hidden=256, heads=4, intermediate=1024)  # K = 256 / 1024: multiples of the fp8 K step (128)
    _, ids, mask, _ = synth_batch(8, 32, 32, 32, cfg["vocab"], seed=5)
    wgt = torch.randn(8, 256, generator=torch.Generator().manual_seed(9))
    net = BertTextNet(cfg)
    net.precision = "fp8"
    sd = cpu_state(net)

    def feat_fn(work, pol):
        _, pooled = bert_forward(work, "bert.", ids, mask, cfg, pol)
        return pooled @ pol.qw(work["proj.weight"]).t() + work["proj.bias"]

    ref8, ref16 = feat_fn(sd, BF16G_FP8), feat_fn(sd, BF16G)
    net.to(dev)
    out = net(ids.to(dev), mask.to(dev))
    assert torch.isfinite(out).all()
    e_oracle = ((out.detach().cpu() - ref8).norm() / ref8.norm()).item()
    price = ((ref8 - ref16).norm() / ref16.norm()).item()
    print(f"fp8 engine vs fp8-policy oracle: rel L2 {e_oracle:.3e}; fp8-policy vs bf16-policy oracle (price of e4m3): {price:.3e}")
    assert price > 5e-3, "the fp8 policy really quantizes"
    assert e_oracle < 0.35 * price + 5e-3, (e_oracle, price)  # well inside the format's own error: the same function
    (out * wgt.to(dev)).sum().backward()
    names = [n for n, _ in net.named_parameters()]
    ref_g = _oracle_grads(sd, names, lambda w: (feat_fn(w, BF16G_FP8) * wgt).sum())
    _check_grads(_grads(net), ref_g, 8e-2, "fp8 forward / bf16 backward vs the fp8-policy oracle", l2=True)


def test_fp8_linear_propagates_non_finite_inputs(dev):
    """A NaN / Inf in a quantized Linear's input must reach its output (the amax pass turns the scale into Inf): otherwise the
    clamp of the e4m3 conversion would launder it into +-448 and the trainer's NaN rule (Trainer.py:74-76) could never fire for
    a fault upstream of an fp8 GEMM."""
    from multimodal_sentiment_aanalysis_amd import kernels as K
    torch.manual_seed(0)
    x = torch.randn(256, 256, device=dev).bfloat16()
    w = torch.randn(128, 256, device=dev).bfloat16()
    wq, sw = K.fp8_quantize(w)

    def linear(xin, rows):
        xq, sx = K.fp8_quantize_rows(xin) if rows else K.fp8_quantize(xin)
        y = torch.empty(256, 128, dtype=torch.bfloat16, device=dev)
        assert K.gemm_fp8(xq, sx, wq, sw, y, row_scales=rows) == 0
        return y.float()

    for rows in (False, True):  # per-tensor scale; per-token scales (the engine's form: only the faulty token's row goes non-finite)
        assert torch.isfinite(linear(x, rows)).all()
        for bad in (float("nan"), float("inf"), float("-inf")):
            xb = x.clone()
            xb[17, 3] = bad
            assert not torch.isfinite(linear(xb, rows)).all(), f"{bad} was laundered into finite values (rows={rows})"
    wb = w.clone()
    wb[5, 9] = float("nan")
    flat = torch.cat([w.reshape(-1), wb.reshape(-1)])
    img, sc = K.fp8_quantize_batch(flat, [0, w.numel()], [w.numel(), w.numel()])
    assert torch.isfinite(sc[0]) and not torch.isfinite(sc[1]), "a non-finite weight must make its tensor's scale non-finite"


@pytest.mark.parametrize("precision,pol,tol_f,tol_g", [("fp32", FP32, 5e-5, 5e-4), ("bf16", BF16G, 2e-2, 8e-2)])
def test_bert_large_dimensions_engine_vs_oracle(dev, precision, pol, tol_f, tol_g):
    """J1 (BASELINE configs[3]) at BERT-large's OWN dimensions — hidden 1024, 16 heads, FFN 4096, S = 256 — two layers, B = 2:
    forward and every parameter gradient against the oracle (fp32 tight; bf16 against the bf16 storage policy with rounded
    gradients). The tile / K-split choices for N in {1024, 3072, 4096}, K in {1024, 4096}, the S = 256 attention forward and its
    recompute backward, and LayerNorm at H = 1024 are exactly the code BERT-large runs; depth adds nothing new (test_c3 runs all
    24 layers)."""
    torch.manual_seed(0)
    cfg = dict(hidden=1024, layers=2, heads=16, intermediate=4096, vocab=2000, max_pos=256, type_vocab=2, ln_eps=1e-12)
    net = BertTextNet(cfg)
    net.precision = precision
    sd = cpu_state(net)
    B, S = 2, 256
    _, ids, mask, _ = synth_batch(B, S, 32, 32, cfg["vocab"], seed=5)
    mask[1, S - 37:] = 0
    wgt = torch.randn(B, 256, generator=torch.Generator().manual_seed(9))

    def feat_fn(work):
        _, pooled = bert_forward(work, "bert.", ids, mask, cfg, pol)
        return pooled @ pol.qw(work["proj.weight"]).t() + work["proj.bias"]

    ref = feat_fn(sd)
    net.to(dev)
    out = net(ids.to(dev), mask.to(dev))
    assert rel_err(out, ref) < tol_f, f"BERT-large-dims feature ({precision}) {rel_err(out, ref)}"
    (out * wgt.to(dev)).sum().backward()
    names = [n for n, _ in net.named_parameters()]
    ref_g = _oracle_grads(sd, names, lambda w: (feat_fn(w) * wgt).sum())
    _check_grads(_grads(net), ref_g, tol_g, f"BERT-large dims {precision}", l2=True)


def test_bert_base_full_size_backward_bf16(dev):
    """E1 at its real size on the benchmarked kernels: BERT-base (12 layers, H = 768, 12 heads, FFN 3072), S = 128, B = 4, bf16
    MFMA GEMMs / attention / grouped weight gradients, forward AND backward against the oracle under the bf16 storage policy
    with gradients rounded where the device stores them (BF16G). BERT has no ReLU / BatchNorm, so — unlike the ResNet — a
    free-running comparison is meaningful (round 1 had no full-size backward check of the text encoder)."""
    from multimodal_sentiment_aanalysis_amd.engine import BERT_BASE
    torch.manual_seed(0)
    net = BertTextNet(BERT_BASE)
    net.precision = "bf16"
    sd = cpu_state(net)
    B, S = 4, 128
    _, ids, mask, _ = synth_batch(B, S, 32, 32, BERT_BASE["vocab"], seed=5)
    mask[:, S - 9:] = 0
    wgt = torch.randn(B, 256, generator=torch.Generator().manual_seed(9))

    def feat_fn(work):
        _, pooled = bert_forward(work, "bert.", ids, mask, BERT_BASE, BF16G)
        return pooled @ BF16G.qw(work["proj.weight"]).t() + work["proj.bias"]

    ref = feat_fn(sd)
    net.to(dev)
    out = net(ids.to(dev), mask.to(dev))
    assert rel_err(out, ref) < 2e-2, f"BERT-base feature {rel_err(out, ref)}"
    (out * wgt.to(dev)).sum().backward()
    names = [n for n, _ in net.named_parameters()]
    ref_g = _oracle_grads(sd, names, lambda w: (feat_fn(w) * wgt).sum())
    _check_grads(_grads(net), ref_g, 8e-2, "BERT-base bf16 (full size)", l2=True)
