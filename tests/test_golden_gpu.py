"""GPU: the HIP product path against the golden vectors generated from the reference's own modules
(tests/golden/*.npz) — the same fixtures that pin the oracle on the CPU (test_oracle_golden.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import multimodal_sentiment_aanalysis_amd as mm
from multimodal_sentiment_aanalysis_amd.engine import BertTextNet, HeadEngine, materialize
from multimodal_sentiment_aanalysis_amd.fused import FlatAdamW
from multimodal_sentiment_aanalysis_amd._lib import HEAD_MM_FUSION, HEAD_WEIGHTED

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    d = np.load(os.path.join(G, name))
    return {k: torch.from_numpy(np.array(d[k])) for k in d.files}


def sub(d, prefix):
    return {k[len(prefix):]: v.clone() for k, v in d.items() if k.startswith(prefix)}


def close(a, b, tol, what, floor=0.0):
    scale = max(b.double().abs().max().item(), floor, 1e-12)
    err = (a.detach().cpu().double() - b.double()).abs().max().item() / scale
    assert err < tol, f"{what}: {err:.3e}"


def check_param_grads(module, d, prefix, tol, what):
    gs = {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}
    gmax = max(v.abs().max().item() for v in gs.values())
    for n, p in module.named_parameters():
        if n in gs:
            assert p.grad is not None, n
            close(p.grad, gs[n], tol, f"{what} grad {n}", floor=1e-3 * gmax)


@pytest.mark.parametrize("tag", ["l1", "l4"])
def test_a1_cross_modal_transformer(dev, tag):
    d = load(f"a1_cross_modal_{tag}.npz")
    m = mm.CrossModalTransformer()
    m.load_state_dict(sub(d, "w."))
    m.to(dev)
    q, k, v = (d[n].to(dev).requires_grad_(True) for n in ("q", "k", "v"))
    out = m(q, k, v)
    close(out, d["out"], 5e-6, "A1 out")
    (out * d["wgt"].to(dev)).sum().backward()
    close(q.grad, d["dq"], 5e-5, "dq"); close(k.grad, d["dk"], 5e-5, "dk"); close(v.grad, d["dv"], 5e-5, "dv")
    check_param_grads(m, d, "g.", 1e-4, "A1")


class _Fusion3(HeadEngine):
    """The reference's 3-modality ME-MHACL fusion (MultimodalModel.py:388-404) on the kind-1 engine."""
    kind = HEAD_MM_FUSION

    def __init__(self, pool):
        super().__init__()
        self.pool = pool
        self._init_head()

    def _base_cfg(self):
        c = super()._base_cfg()
        c.update(embed=256, heads=8, tokens=3, pool_mode=0 if self.pool == "max" else 1)
        return c

    def _out_dims(self):
        return [256]


def test_a2_mm_fusion(dev):
    d = load("a2_mm_fusion.npz")
    for mode in ("train", "eval"):
        m = _Fusion3("max")
        m.load_state_dict(sub(d, "w."))
        m.to(dev).train(mode == "train")
        feats = [d[f"f{i}"].to(dev).requires_grad_(True) for i in range(3)]
        out = m._run(*feats)[0]
        # 3e-5: train mode ends in BatchNorm over 16 rows; the fp32 summation order of the batch statistics differs
        # from torch's (8-channel lanes, fixed tree), worth 1-2e-5 relative on the normalised output. The north-star
        # bound is 1e-3 on logits.
        close(out, d[f"{mode}.out"], 3e-5, f"A2 {mode} out")
        (out * d["wgt"].to(dev)).sum().backward()
        for i in range(3):
            close(feats[i].grad, d[f"{mode}.df{i}"], 2e-4, f"A2 {mode} df{i}")
        check_param_grads(m, d, f"{mode}.g.", 2e-4, f"A2 {mode}")
        if mode == "train":
            sd = m.state_dict()
            for k2 in ("fusion_mlp.2.running_mean", "fusion_mlp.2.running_var"):
                close(sd[k2], d["train.post." + k2], 1e-5, k2)
    m = _Fusion3("mean")
    m.load_state_dict(sub(d, "w."))
    m.to(dev).eval()
    with torch.no_grad():
        out = m._run(*[d[f"f{i}"].to(dev) for i in range(3)])[0]
    close(out, d["eval.mean_pool_out"], 1e-5, "A2 mean-pool twin")


class _RefHead(HeadEngine):
    """Reference MultimodalTransformerModel fusion path (:287-313) with identity encoders: two CrossModalTransformers
    + the weighted head (with the valence head), using the reference's module names so its state_dict loads."""
    kind = HEAD_WEIGHTED

    def __init__(self, valence=True):
        super().__init__()
        self.valence = valence
        self.cross_attn_e2p = mm.CrossModalTransformer()
        self.cross_attn_p2e = mm.CrossModalTransformer()
        self._init_head()

    def _base_cfg(self):
        c = super()._base_cfg()
        c.update(embed=256, num_classes=3, valence=int(self.valence), dropout_p=0.0)
        return c

    def _out_dims(self):
        return [3, 128] + ([3] if self.valence else [])

    def forward(self, eeg, eye, pps):
        e2p = self.cross_attn_e2p(eeg, eye, eye)
        p2e = self.cross_attn_p2e(eeg, pps, pps)
        return self._run(eeg, eye, pps, e2p, p2e)


def test_a4_fusion_head(dev):
    d = load("a4_fusion_head.npz")
    for mode in ("train", "eval"):
        m = _RefHead()
        m.load_state_dict(sub(d, "w."), strict=False)  # contrastive_weight / temperature are not part of this path
        materialize(m, dev)
        m.train(mode == "train")
        f = [d[f"f{i}"].to(dev).requires_grad_(True) for i in range(3)]
        outs = m(*f)
        close(outs[0], d[f"{mode}.arousal"], 2e-5, f"A4 {mode} arousal")
        close(outs[2], d[f"{mode}.valence"], 2e-5, f"A4 {mode} valence")
        ((outs[0] * d["wa"].to(dev)).sum() + (outs[2] * d["wv"].to(dev)).sum()).backward()
        for i in range(3):
            close(f[i].grad, d[f"{mode}.df{i}"], 5e-4, f"A4 {mode} df{i}")
        check_param_grads(m, d, f"{mode}.g.", 1e-3, f"A4 {mode}")
        if mode == "train":
            sd = m.state_dict()
            for k2 in [k for k in d if k.startswith("train.post.")]:
                close(sd[k2[len("train.post."):]], d[k2], 2e-5, k2)


def test_a5_a6_heads_and_ce(dev):
    d = load("a5_a6_heads_ce.npz")
    c = mm.Classifier()
    c.dropout_p = 0.0
    c.load_state_dict(sub(d, "cls.w."))
    c.to(dev).train()
    x = d["x"].to(dev).requires_grad_(True)
    a, v = c(x)
    close(a, d["cls_a"], 5e-6, "cls a"); close(v, d["cls_v"], 5e-6, "cls v")
    ((a * d["wa"].to(dev)).sum() + (v * d["wv"].to(dev)).sum()).backward()
    close(x.grad, d["cls_dx"], 5e-5, "cls dx")
    check_param_grads(c, d, "cls.g.", 5e-5, "classifier")
    p = mm.ProjectionHead()
    p.dropout_p = 0.0
    p.load_state_dict(sub(d, "proj.w."))
    p.to(dev).train()
    x = d["x"].to(dev).requires_grad_(True)
    z = p(x)
    close(z, d["proj_z"], 1e-5, "proj z")
    (z * d["wz"].to(dev)).sum().backward()
    close(x.grad, d["proj_dx"], 2e-4, "proj dx")
    check_param_grads(p, d, "proj.g.", 2e-4, "projection head")
    lg = d["ce_logits"].to(dev).requires_grad_(True)
    loss = mm.CrossEntropyLoss()(lg, d["ce_labels"].to(dev))
    assert abs(loss.item() - d["ce_loss"].item()) < 1e-6
    loss.backward()
    close(lg.grad, d["ce_dlogits"], 1e-6, "CE dlogits")


def test_a7_train_step_fused_optimizer(dev):
    """Two reference train steps (Trainer.py:59-81) through the HIP CE / grad-norm / AdamW kernels on the flat buffers."""
    d = load("a7_train_step.npz")
    m = _RefHead(valence=False)  # the older single-head contract trains the arousal head only
    m.load_state_dict({k: v for k, v in sub(d, "w0.").items() if not k.startswith("valence_head")}, strict=False)
    state = materialize(m, dev)
    m.train()
    opt = FlatAdamW(state, lr=1e-4, weight_decay=0.01, max_norm=1.0)
    f = [d[f"f{i}"].to(dev) for i in range(3)]
    labels = d["labels"].to(dev)
    crit = mm.CrossEntropyLoss()
    for step in (1, 2):
        state.flat_g.zero_()
        loss = crit(m(*f)[0], labels)
        loss.backward()
        opt.step()
        assert abs(loss.item() - d[f"loss{step}"].item()) < 1e-5
        assert abs(opt.norm_out[0].item() - d[f"norm{step}"].item()) / d[f"norm{step}"].item() < 1e-4
        grads = {n: p.grad.detach().cpu() for n, p in m.named_parameters()}
        gmax = max(g.abs().max().item() for g in grads.values())
        for n, p in m.named_parameters():
            diff = (p.detach().cpu().double() - d[f"w{step}.{n}"].double()).abs()
            assert diff.max().item() <= 2.2e-4 * step, f"step {step} {n}: {diff.max().item():.3e}"
            if grads[n].abs().max().item() >= 1e-5 * gmax:
                assert (diff > 2e-6).double().mean().item() < 2e-3, f"step {step} {n}"


def test_e1_bert_mini_against_transformers(dev):
    d = load("e1_bert_mini.npz")
    cfg = dict(hidden=128, layers=2, heads=2, intermediate=512, vocab=1000, max_pos=64, type_vocab=2, ln_eps=1e-12)
    net = BertTextNet(cfg)
    net.precision = "fp32"
    hf = {"bert." + k[2:]: v for k, v in d.items() if k.startswith("w.")}
    # make the projection read the pooled output back: proj = [I_128 ; 0] so feat[:, :128] == pooled
    proj_w = torch.zeros(256, 128)
    proj_w[:128] = torch.eye(128)
    hf["proj.weight"], hf["proj.bias"] = proj_w, torch.zeros(256)
    net.load_state_dict(hf)
    net.to(dev)
    with torch.no_grad():
        out = net(d["ids"].to(dev))
        close(out[:, :128], d["pooled_nomask"], 5e-5, "pooled (HF golden)")
        out = net(d["ids"].to(dev), d["mask"].to(dev))
        close(out[:, :128], d["pooled_mask"], 5e-5, "pooled masked (HF golden)")


def test_e1_bert_base_seed_regenerated(dev):
    from multimodal_sentiment_aanalysis_amd.engine import BERT_BASE
    d = load("e1_bert_base_seed1234.npz")
    torch.manual_seed(1234)
    net = BertTextNet(BERT_BASE)
    net.precision = "fp32"
    with torch.no_grad():
        net._pmap["proj.weight"].zero_()
        net._pmap["proj.weight"][:256, :256] = torch.eye(256)
        net._pmap["proj.bias"].zero_()
    net.to(dev)
    with torch.no_grad():
        out = net(d["ids"].to(dev))
    close(out, d["pooled"][:, :256], 2e-4, "BERT-base pooled[:, :256] (HF golden), fp32 engine")
    net.precision = "bf16"
    materialize(net, dev)
    with torch.no_grad():
        out16 = net(d["ids"].to(dev))
    err = (out16.cpu() - d["pooled"][:, :256]).abs().max().item()
    print(f"BERT-base bf16 engine vs HF fp32 pooled: max abs err {err:.3e}")
    assert err < 5e-2


def _identity_proj(net, width):
    with torch.no_grad():
        net._pmap["proj.weight"].copy_(torch.eye(width))
        net._pmap["proj.bias"].zero_()


def test_e2_resnet_mini_against_transformers(dev):
    """E2 on the device against the transformers.ResNetModel (float64) golden, stored weights: train-mode features, every
    parameter gradient, running statistics after the forward, eval-mode features with those statistics. The projection is
    the identity so that the engine's [B, out_dim] output IS the pooled feature vector."""
    from multimodal_sentiment_aanalysis_amd.engine import ResNetImageNet
    d = load("e2_resnet_mini.npz")
    rcfg = dict(blocks=(1, 2, 1, 1), widths=(64, 64, 128, 128))
    # gradient bounds: train-mode BatchNorm makes dbeta / dgamma badly conditioned sums (the fp32 CPU oracle itself is 3e-3 from
    # the float64 values on this net; the device's summation order 1.3e-2 on one BN bias, others below 5e-3)
    for precision, tol_f, tol_g, tol_s in (("fp32", 1e-4, 3e-2, 1e-4), ("bf16", 8e-2, 3e-1, 3e-2)):
        net = ResNetImageNet(rcfg, out_dim=512)
        net.precision = precision
        sd = {"resnet." + k: v.clone() for k, v in sub(d, "w.").items()}
        sd["proj.weight"], sd["proj.bias"] = torch.eye(512), torch.zeros(512)
        net.load_state_dict(sd)
        net.to(dev).train()
        out = net(d["image"].to(dev))
        close(out, d["pooled_train"], tol_f, f"{precision} pooled (train)")
        (out * d["wgt"].to(dev)).sum().backward()
        gs = sub(d, "g.")
        gmax = max(v.abs().max().item() for v in gs.values())
        worst, errs = ("", 0.0), []
        for n, p in net.named_parameters():
            if not n.startswith("resnet."):
                continue
            r = gs[n[len("resnet."):]].double()
            e = ((p.grad.detach().cpu().double() - r).norm() / r.norm().clamp_min(1e-3 * gmax * r.numel() ** 0.5)).item()
            errs.append(e)
            worst = max(worst, (n, e), key=lambda t: t[1])
        if precision == "fp32":
            # the loose bound is for the BatchNorm scale / shift gradients only (badly conditioned column sums over the batch: the
            # fp32 CPU oracle itself is 3e-3 from the float64 values on this net; measured on MI355X: 1.3e-2 on one shift, 9.7e-3 on
            # one scale); the convolution weights keep a tight one. The stem's BN bias is ALSO pinned at 0.15 in the teacher-forced
            # backward test (tests/test_engines_gpu.py::test_resnet_backward_teacher_forced).
            dims = {n: p.dim() for n, p in net.named_parameters()}
            named = list(zip([n for n in dims if n.startswith("resnet.")], errs))
            loose = [(n, e) for n, e in named if dims[n] == 1]
            tight = [(n, e) for n, e in named if dims[n] != 1]
            wl, wt_ = max(loose, key=lambda t: t[1]), max(tight, key=lambda t: t[1])
            print(f"e2 mini fp32 gradients: worst BatchNorm scale/shift {wl[1]:.2e} at {wl[0]}; worst convolution weight {wt_[1]:.2e} at {wt_[0]}")
            assert wl[1] < tol_g, f"{precision}: worst BatchNorm gradient rel-L2 {wl[1]:.3e} at {wl[0]}"
            assert wt_[1] < 1e-2, f"{precision}: worst convolution-weight gradient rel-L2 {wt_[1]:.3e} at {wt_[0]}"
        else:
            # Free-running bf16 on a random-init BatchNorm net with 4 samples (36 of them per channel in the last stage): two correct
            # implementations decorrelate. The yardstick is the ORACLE under the bf16 storage policy on the
            # same weights and batch: the device must be no further from the float64 golden than that is (x2 + a floor), on the
            # typical and on the worst tensor. The bf16 backward is pinned tensor by tensor in the teacher-forced test.
            from oracle.policy import BF16G
            from oracle.resnet import resnet_forward
            ocfg = dict(blocks=rcfg["blocks"], widths=rcfg["widths"], expansion=4)
            osd = {"r." + k: v.clone() for k, v in sub(d, "w.").items()}
            names = [k for k in gs]
            params = {k: osd["r." + k].requires_grad_(True) for k in names}
            pooled_pol = resnet_forward(osd, "r.", d["image"], ocfg, True, BF16G)
            (pooled_pol * d["wgt"]).sum().backward()
            pol = sorted(((params[k].grad.double() - gs[k].double()).norm() /
                          gs[k].double().norm().clamp_min(1e-3 * gmax * gs[k].numel() ** 0.5)).item() for k in names)
            med, pmed, pworst = sorted(errs)[len(errs) // 2], pol[len(pol) // 2], pol[-1]
            print(f"e2 mini bf16 gradients vs float64 golden: device median {med:.3f} worst {worst[1]:.3f}; bf16-policy oracle median "
                  f"{pmed:.3f} worst {pworst:.3f}")
            assert med < 2.0 * pmed + 0.05 and worst[1] < 2.0 * pworst + 0.1, (med, worst, pmed, pworst)
        post = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        for k, v in sub(d, "w1.").items():
            if "num_batches" in k:
                assert int(post["resnet." + k]) == int(v)
            else:
                close(post["resnet." + k], v, tol_s, f"{precision} running statistic {k}")
        net.eval()
        with torch.no_grad():
            ev = net(d["image"].to(dev))
        close(ev, d["pooled_eval"], tol_f, f"{precision} pooled (eval)")
        print(f"e2 mini {precision}: worst gradient rel-L2 {worst[1]:.2e} ({worst[0]})")


def test_e2_resnet50_seed_regenerated(dev):
    """ResNet-50 at full size (B=4, 224x224) against transformers.ResNetModel in float64: weights regenerated from seed 1234 on
    both sides. fp32 engine: pooled features <= 1e-4 (train and eval); gradients within the fp32 summation-order floor of this
    random-init network (the fp32 CPU oracle itself is 2.5e-2 from the float64 values: tests/test_oracle_golden.py), asserted at
    2x that. bf16 engine: features bounded by the bf16-policy oracle's own distance (~0.3 free-running)."""
    from multimodal_sentiment_aanalysis_amd.engine import RESNET50, ResNetImageNet
    d = load("e2_resnet50_seed1234.npz")
    image = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(int(d["image_seed"])))
    for precision, tol_f, tol_g in (("fp32", 1e-4, 5e-2), ("bf16", 0.6, None)):
        torch.manual_seed(1234)
        net = ResNetImageNet(RESNET50, out_dim=2048)
        net.precision = precision
        _identity_proj(net, 2048)
        net.to(dev).train()
        out = net(image.to(dev))
        close(out, d["pooled_train"], tol_f, f"ResNet-50 {precision} pooled (train)")
        if tol_g is not None:
            (out * d["wgt"].float().to(dev)).sum().backward()
            worst, nworst = ("", 0.0), 0.0
            for n, p in net.named_parameters():
                if not n.startswith("resnet."):
                    continue
                k = n[len("resnet."):]
                g = p.grad.detach().cpu().reshape(-1)  # logical OIHW order (the storage is channels_last)
                stride = max(1, g.numel() // 512)
                e = ((g[::stride][:512].double() - d["gs." + k]).norm() / (d["gs." + k].norm() + 1e-30)).item()
                worst = max(worst, (n, e), key=lambda t: t[1])
                nworst = max(nworst, abs(g.double().norm().item() - d["gn." + k].item()) / d["gn." + k].item())
            print(f"ResNet-50 {precision} vs transformers float64: worst gradient-sample rel-L2 {worst[1]:.2e} ({worst[0]}), "
                  f"worst norm error {nworst:.2e}")
            assert worst[1] < tol_g and nworst < 2e-2, (worst, nworst)
        net.eval()
        with torch.no_grad():
            ev = net(image.to(dev))
        close(ev, d["pooled_eval"], 1e-4 if precision == "fp32" else 3e-2, f"ResNet-50 {precision} pooled (eval)")


def test_trainer_and_tester_contracts(dev, tmp_path):
    from multimodal_sentiment_aanalysis_amd.dataLoader import MultimodalDataLoader
    from multimodal_sentiment_aanalysis_amd.Tester import Tester
    from multimodal_sentiment_aanalysis_amd.Trainer import Trainer
    from util import MINI_BERT, MINI_RESNET
    torch.manual_seed(0)
    dl = MultimodalDataLoader(None, batch_size=8, n=48, seq_len=16, image_size=64, vocab=1000)
    train, test = dl.dict_loader(1, True), dl.dict_loader(1, False)
    for fused in (True, False):
        model = mm.MultimodalTransformerModel(bert_config=MINI_BERT, resnet_config=MINI_RESNET)
        tr = Trainer(model, train, test, device="cuda", fused=fused, precision="fp32")
        l0 = tr.train_epoch(1)
        l1 = tr.train_epoch(2)
        assert all(x == x for x in l0 + l1) and 0.0 <= l1[3] <= 1.0
        te = tr.test()
        assert te[0] == te[0]
    os.chdir(tmp_path)
    torch.save({"module." + k: v for k, v in model.state_dict().items()}, "ddp_style.pth")
    t = Tester(model, test, device="cuda")
    res = t.run("ddp_style.pth")  # `module.` prefix is stripped (Tester.py:32-33)
    assert set(res) == {"loss", "accuracy", "predictions", "labels", "probabilities"}
    assert res["probabilities"].shape == (2, 3) and abs(res["probabilities"].sum(1) - 1).max() < 1e-5
    single = t.predict_single({k: v[0] for k, v in next(iter(test))[0].items()})
    assert single["probabilities"].shape == (3,)


def test_n1_contrastive_losses(dev):
    """N1 on the fused HIP kernels (mmsa_infonce_fwd_bwd / mmsa_supcon_fwd_bwd) through the model-level mirrors
    (MultimodalTransformerModel.compute_contrastive_loss, train.contrastive_loss) against the reference's own values and
    autograd gradients (tests/golden/n1_contrastive.npz). fp32; the [B,B] soft-max at T = 0.01 spans e^±200, so the
    tolerance is on the scale of the largest gradient entry."""
    from multimodal_sentiment_aanalysis_amd import train as T
    from multimodal_sentiment_aanalysis_amd.engine import supervised_infonce
    d = load("n1_contrastive.npz")
    for tag in "abcd":
        same = bool(d[f"infonce.{tag}.same"].item())
        f1 = d[f"infonce.{tag}.f1"].to(dev).requires_grad_(True)
        f2 = f1 if same else d[f"infonce.{tag}.f2"].to(dev).requires_grad_(True)
        temp = d[f"infonce.{tag}.T"].to(dev).requires_grad_(True)
        loss = supervised_infonce(f1, f2, d[f"infonce.{tag}.labels"].to(dev), temp)
        loss.backward()
        close(loss, d[f"infonce.{tag}.loss"], 2e-5, f"infonce {tag} loss")
        close(f1.grad, d[f"infonce.{tag}.df1"], 1e-4, f"infonce {tag} df1")
        close(temp.grad, d[f"infonce.{tag}.dT"], 1e-4, f"infonce {tag} dT")
        if not same:
            close(f2.grad, d[f"infonce.{tag}.df2"], 1e-4, f"infonce {tag} df2")
    for tag in "abc":
        z1 = d[f"supcon.{tag}.z1"].to(dev).requires_grad_(True)
        z2 = d[f"supcon.{tag}.z2"].to(dev).requires_grad_(True)
        loss = T.contrastive_loss(z1, z2, d[f"supcon.{tag}.labels"].to(dev))
        (2.0 * loss).backward()  # a non-unit upstream gradient
        close(loss, d[f"supcon.{tag}.loss"], 2e-5, f"supcon {tag} loss")
        close(z1.grad / 2.0, d[f"supcon.{tag}.dz1"], 1e-4, f"supcon {tag} dz1")
        close(z2.grad / 2.0, d[f"supcon.{tag}.dz2"], 1e-4, f"supcon {tag} dz2")


def test_n1_multitask_forward_returns_contrastive_terms(dev):
    """MultiTaskTrainer contract (MultimodalModel.py:262,319-322): with labels the model returns
    (arousal, valence, c1, c2, c3), c_k = contrastive_weight * InfoNCE(feature_k, feature_k | arousal labels)."""
    from util import MINI_BERT, MINI_RESNET, synth_batch
    from oracle import fusion as OF
    torch.manual_seed(0)
    m = mm.MultimodalTransformerModel(bert_config=MINI_BERT, resnet_config=MINI_RESNET, multitask=True, dropout=0.0)
    with torch.no_grad():
        m.temperature.fill_(0.2)
        m.contrastive_weight.fill_(0.7)
    image, ids, mask, labels = synth_batch(8, 16, 64, 64, MINI_BERT["vocab"], seed=3)
    m.precision = "fp32"
    m.to(dev).train()
    out = m(image.to(dev), ids.to(dev), mask.to(dev), (labels.to(dev), labels.to(dev)))
    assert len(out) == 5 and out[0].shape == (8, 3) and out[1].shape == (8, 3)
    i, t = m.encoder.features(image.to(dev), ids.to(dev), mask.to(dev))
    ref_t = 0.7 * OF.supervised_infonce(t.detach().cpu(), t.detach().cpu(), labels, torch.tensor(0.2))
    ref_i = 0.7 * OF.supervised_infonce(i.detach().cpu(), i.detach().cpu(), labels, torch.tensor(0.2))
    assert abs(out[3].item() - ref_t.item()) < 2e-5 * max(1.0, abs(ref_t.item())), (out[3].item(), ref_t.item())
    assert abs(out[4].item() - ref_i.item()) < 2e-5 * max(1.0, abs(ref_i.item())), (out[4].item(), ref_i.item())
    (out[0].sum() + out[2] + out[3] + out[4]).backward()
    assert m.temperature.grad is not None and torch.isfinite(m.temperature.grad).all()
    assert m.contrastive_weight.grad is not None


class _RefHeadMT(_RefHead):
    """_RefHead with the reference model's MultiTaskTrainer contract (MultimodalModel.py:262-322): identity encoder slots
    under the reference's attribute names, `(arousal, valence, c1, c2, c3) = model(eeg, eye, pps, labels=(a, v))`."""

    def __init__(self):
        super().__init__(valence=True)
        import torch.nn as nn
        self.eeg_net, self.eye_net, self.pps_net = nn.Identity(), nn.Identity(), nn.Identity()
        self.contrastive_weight = nn.Parameter(torch.ones(1))
        self.temperature = nn.Parameter(torch.tensor(0.01))

    def forward(self, eeg, eye, pps, labels=None):
        from multimodal_sentiment_aanalysis_amd.engine import supervised_infonce
        outs = super().forward(eeg, eye, pps)
        if labels is None:
            return outs[0], outs[2]
        c = [self.contrastive_weight * supervised_infonce(f, f, labels[0], self.temperature) for f in (eeg, eye, pps)]
        return outs[0], outs[2], c[0], c[1], c[2]


@pytest.mark.parametrize("hip_optimizer", [True, False])
def test_n2_multitask_phases_against_reference(dev, hip_optimizer):
    """N2 pinned on the reference's OWN MultiTaskTrainer (tests/golden/make_golden.py::gen_multitask_phases calls it on the
    identity-encoder fusion head): one epoch of phase 2 and one of phase 3, two batches each, then evaluate(). Phase 3 is the
    tricky one: cross-attention / weighting / fusion modules are trainable but not handed to the optimizer, so their gradients
    are never zeroed, accumulate across batches (scaled in place by every clip) and enter every clip norm, while only the
    valence head moves. hip_optimizer=True: clip + AdamW as HIP kernels over the phase's sub-ranges of the flat buffers
    (fused.PhaseOptimizer); False: torch.optim.AdamW on the same parameter views."""
    from torch.utils.data import DataLoader, TensorDataset
    from multimodal_sentiment_aanalysis_amd.dataLoader import MultiTaskTrainer
    from multimodal_sentiment_aanalysis_amd.fused import PhaseOptimizer
    d = load("n2_multitask_phases.npz")
    m = _RefHeadMT()
    missing = m.load_state_dict(sub(d, "w0."), strict=False)
    assert not missing.missing_keys, missing
    ds = TensorDataset(d["f0"], d["f1"], d["f2"], d["arousal"], d["valence"])
    loader = DataLoader(ds, batch_size=16, shuffle=False)

    class FeatTrainer(MultiTaskTrainer):
        def _feeds(self, batch):  # the three "modalities" are feature vectors here (identity encoder slots)
            x1, x2, x3, a, v = batch
            return (x1.to(self.device).float(), x2.to(self.device).float(), x3.to(self.device).float()), (a.to(self.device), v.to(self.device))

    tr = FeatTrainer(m, loader, loader, device=dev, hip_optimizer=hip_optimizer)

    def check_state(tag, steps):
        sd = m.state_dict()
        for k, ref in sub(d, tag).items():
            if "num_batches" in k:
                assert int(sd[k]) == int(ref), k
                continue
            diff = (sd[k].detach().cpu().double() - ref.double()).abs().max().item()
            # an AdamW step moves every element by ~lr = 1e-4 (first steps: update = +-lr whatever the gradient's size), so a
            # gradient element within rounding of 0 may step the other way: bound = 2.2 lr per step taken, as in A7
            assert diff <= 2.2e-4 * steps + 1e-6 * ref.abs().max().item(), f"{tag}{k}: {diff:.3e}"

    def check_metrics(tag, got):
        for k, v in got.items():
            ref = float(d[f"{tag}.{k}"])
            assert abs(v - ref) <= 2e-4 * max(1.0, abs(ref)), f"{tag}.{k}: {v} vs {ref}"

    tr.clip_norms = []
    r2 = tr.train_epoch_phase2(1)
    assert isinstance(tr.phase2_optimizer, PhaseOptimizer) == hip_optimizer
    check_metrics("train_p2", r2)
    check_state("w_p2.", 2)
    check_metrics("eval_p2", tr.evaluate())
    m.train()
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    r3 = tr.train_epoch_phase3(1)
    check_metrics("train_p3", r3)
    check_state("w_p3.", 4)
    moved = {k for k, v in m.named_parameters() if not torch.equal(v.detach(), before[k])}
    assert moved and all(k.startswith("valence_head.") for k in moved), sorted(moved)[:5]
    # tight check of what phase 3 moved (the accumulate-and-rescale quirk changes its clip coefficient on the 2nd batch)
    for k in moved:
        ref = d["w_p3." + k]
        delta_ref = (ref - d["w_p2." + k]).double()
        delta = (m.state_dict()[k].detach().cpu() - d["w_p2." + k]).double()
        # (tensors in front of a BatchNorm have an analytically zero gradient: their AdamW update is rounding noise / (noise +
        #  eps), ~1e-6, and differs between any two implementations — only tensors that really moved are compared tightly)
        if delta_ref.abs().median() > 2e-5:
            assert (delta - delta_ref).norm() <= 0.05 * delta_ref.norm() + 1e-7, k
    check_metrics("eval_p3", tr.evaluate())
    # the total norm of every batch against what the reference's own clip_grad_norm_(self.model.parameters(), 1.0) returned:
    # phase 3's norms include the stale gradient the (now frozen) arousal head kept from phase 2's last batch, rescaled in place
    # by every clip since — a clip over the phase's trainable modules only reads 12.3 -> wrong by that head's share
    got = torch.stack([t.reshape(()) for t in tr.clip_norms]).double().cpu()
    assert got.shape == d["clip_norms"].shape
    assert ((got - d["clip_norms"]).abs() <= 2e-4 * d["clip_norms"]).all(), (got, d["clip_norms"])


def test_n2_multitask_trainer_phases(dev, tmp_path, monkeypatch):
    """N2: the reference's five-phase MultiTaskTrainer surface on the HIP modules — each phase moves exactly the parameters
    its optimizer owns (frozen sub-graphs stay bit-identical), metrics have the reference's keys, run() walks all phases."""
    from torch.utils.data import DataLoader, TensorDataset
    from util import MINI_BERT, MINI_RESNET
    from multimodal_sentiment_aanalysis_amd.dataLoader import MultiTaskTrainer
    g = torch.Generator().manual_seed(7)
    n = 16
    ds = TensorDataset(torch.randn(n, 3, 64, 64, generator=g), torch.randint(0, MINI_BERT["vocab"], (n, 16), generator=g),
                       torch.ones(n, 16), torch.randint(0, 3, (n,), generator=g), torch.randint(0, 3, (n,), generator=g))
    loader = DataLoader(ds, batch_size=8)
    torch.manual_seed(0)
    model = mm.MultimodalTransformerModel(bert_config=MINI_BERT, resnet_config=MINI_RESNET, multitask=True, dropout=0.0)
    tr = MultiTaskTrainer(model, loader, loader, device=dev, test_person=3)

    def snap():
        return {k: v.detach().clone() for k, v in model.named_parameters()}

    def moved(before):
        return {k for k, v in model.named_parameters() if not torch.equal(v.detach(), before[k])}

    b = snap()
    m1 = tr.train_epoch_phase_eye(1)
    ch = moved(b)
    assert ch and all(k.startswith("encoder.text_net.") for k in ch), sorted(ch)[:5]
    assert set(m1) == {"loss", "a_loss", "v_loss", "c_loss", "a_acc", "v_acc"} and m1["a_loss"] == 0 and m1["c_loss"] > 0

    b = snap()
    tr.train_epoch_phase_pps(1)
    ch = moved(b)
    assert ch and all(k.startswith("encoder.image_net.") for k in ch), sorted(ch)[:5]

    b = snap()
    m2 = tr.train_epoch_phase2(1)
    ch = moved(b)
    assert any(k.startswith("arousal_head.") for k in ch) and any(k.startswith("encoder.") for k in ch)
    assert not any(k.startswith("valence_head.") or k in ("temperature", "contrastive_weight") for k in ch), sorted(ch)[:5]
    assert m2["a_loss"] > 0 and m2["c_loss"] == 0

    b = snap()
    tr.train_epoch_phase3(1)
    ch = moved(b)
    assert ch and all(k.startswith("valence_head.") for k in ch), sorted(ch)[:5]

    ev = tr.evaluate()
    assert all(v == v for v in ev.values()) and ev["loss"] > 0 and 0.0 <= ev["a_acc"] <= 1.0

    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(tr, "visualize_progress", lambda: None)
    tr.run(1, 0, 0, 1, 1)
    assert len(tr.metrics["train"]["loss"]) == 4 + 3 and any(f.endswith(".pth") for f in os.listdir(tmp_path))


def test_n4_device_prefetcher(dev):
    """N4: the double-buffered host->device pipeline yields exactly the loader's batches (tuple and (dict, labels) forms),
    on the device, in order."""
    from torch.utils.data import DataLoader, TensorDataset
    from multimodal_sentiment_aanalysis_amd.dataLoader import DevicePrefetcher, MultimodalDataLoader
    g = torch.Generator().manual_seed(3)
    ds = TensorDataset(torch.randn(20, 3, 8, 8, generator=g), torch.randint(0, 9, (20, 4), generator=g), torch.arange(20))
    loader = DataLoader(ds, batch_size=6, pin_memory=True)
    got = list(DevicePrefetcher(loader, dev))
    ref = list(loader)
    assert len(got) == len(ref) == 4 and len(DevicePrefetcher(loader, dev)) == 4
    for a, b in zip(got, ref):
        for x, y in zip(a, b):
            assert x.device.type == "cuda" and torch.equal(x.cpu(), y)
    dl = MultimodalDataLoader(file_path=None, batch_size=2, n=12, image_size=16, seq_len=8, subjects=3).dict_loader(1, False)
    pairs = list(zip(DevicePrefetcher(dl, dev), dl))  # the test split is not shuffled
    assert len(pairs) == 2
    for (d, y), (dr, yr) in pairs:
        assert y.device.type == "cuda" and torch.equal(y.cpu(), yr)
        assert all(torch.equal(d[k].cpu(), dr[k]) for k in dr)


# ------------------------------------------------------------------------------------------------ end-to-end C0 (full size)
def _c0_metrics(dev, precision, fname):
    """One training-mode forward + CE + backward of the full-size model (BERT-base + ResNet-50 + fusion head) on the C0
    batch (16 pairs, seed 1234, weights regenerated from the seed) against tests/golden/c0_*.npz (CPU oracle, fp32;
    generator: tests/golden/make_c0_golden.py). Also returns the distances of the ORACLE's own bf16-storage policy from
    its fp32 self (stored in the fixture): what any bf16-storage implementation of this graph deviates by."""
    from util import synth_batch
    d = np.load(os.path.join(G, fname))
    torch.manual_seed(int(d["seed"]))
    model = mm.MultimodalTransformerModel(dropout=0.0)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith("bn3.weight"):
                v.mul_(float(d["gamma3"]))
    image, ids, mask, labels = synth_batch(16, 128, 224, 224, 30522, seed=int(d["seed"]))
    assert np.array_equal(labels.numpy(), d["labels"])
    materialize(model, dev, precision)
    model.train()
    feats = {}
    inner = model.encoder.features

    def tapped(*a, **k):  # the two encoder features of THIS forward (a second forward would update the BN buffers again)
        i_f, t_f = inner(*a, **k)
        feats["image"], feats["text"] = i_f.detach().clone(), t_f.detach().clone()
        return i_f, t_f

    model.encoder.features = tapped
    logits, _aux = model(image.to(dev), ids.to(dev), mask.to(dev), labels.to(dev))
    loss = mm.CrossEntropyLoss()(logits, labels.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    relmax = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())  # noqa: E731
    m = {"dlogits": float(np.abs(logits.detach().cpu().numpy() - d["logits"]).max()),
         "dloss": abs(float(loss) - float(d["loss"]))}
    pol = {"dlogits": float(np.abs(d["logits_bf16_policy"] - d["logits"]).max()),
           "dloss": abs(float(d["loss_bf16_policy"]) - float(d["loss"])),
           "text_feat": relmax(d["text_feat_bf16_policy"], d["text_feat"]),
           "image_feat": relmax(d["image_feat_bf16_policy"], d["image_feat"])}
    m["text_feat"] = relmax(feats["text"].cpu().numpy(), d["text_feat"])
    m["image_feat"] = relmax(feats["image"].cpu().numpy(), d["image_feat"])
    params = dict(model.named_parameters())
    names, norms, strides = list(d["grad_names"]), d["grad_norms"], d["grad_strides"]
    gtot = float(d["grad_total_norm"])
    worst_norm, worst_l2, tot = ("", 0.0), ("", 0.0), 0.0
    by_group = {}
    for n, ref_norm, stride in zip(names, norms, strides):
        g = params[str(n)].grad
        assert g is not None, f"no gradient for {n}"
        gc = g.detach().float().cpu()
        if gc.dim() == 4:
            gc = gc.contiguous()  # logical OIHW order, as the oracle's gradient
        tot += float((gc.double() ** 2).sum())
        floor = 1e-4 * gtot
        e_norm = abs(float(gc.double().norm()) - ref_norm) / max(ref_norm, floor)
        smp = gc.reshape(-1)[::int(stride)].double().numpy()
        ref = d["gs." + str(n)].astype(np.float64)
        e_l2 = float(np.linalg.norm(smp - ref) / max(np.linalg.norm(ref), floor * (ref.size / gc.numel()) ** 0.5, 1e-30))
        if e_norm > worst_norm[1]:
            worst_norm = (str(n), e_norm)
        if e_l2 > worst_l2[1]:
            worst_l2 = (str(n), e_l2)
        grp = "image" if "image_net" in str(n) else "text" if "text_net" in str(n) else "head"
        by_group[grp] = max(by_group.get(grp, 0.0), e_l2)
    m.update(grad_total_norm=abs(tot ** 0.5 - gtot) / gtot, worst_norm=worst_norm, worst_l2=worst_l2, by_group=by_group)
    for k in ("encoder.image_net.resnet.bn1.running_var", "encoder.image_net.resnet.layer4.2.bn3.running_var"):
        got = model.state_dict()[k].detach().cpu().numpy()
        m["bn:" + k.split("resnet.")[1]] = relmax(got, d["bn." + k])
    del model
    torch.cuda.empty_cache()
    return m, pol


@pytest.mark.parametrize("fname", ["c0_full_size.npz", "c0_damped.npz"])
def test_c0_full_size_fp32(dev, fname):
    """SURVEY.md §8(c) "End-to-end C0" in the exact (fp32-storage) mode: the north-star tolerances, 1e-3 on the logits and
    1e-4 on the loss, at FULL size (BERT-base + ResNet-50, 16 pairs), and every parameter gradient. Two weight sets: the
    default initialisers, and the same with damped residual branches (make_c0_golden.damp_residual_branches)."""
    m, _ = _c0_metrics(dev, "fp32", fname)
    print(f"C0 fp32 {fname}:", m)
    assert m["dlogits"] < 1e-3 and m["dloss"] < 1e-4, m
    assert m["text_feat"] < 1e-4 and m["image_feat"] < 1e-3, m
    assert m["grad_total_norm"] < 1e-3, m
    # per-tensor gradients: the default-init ResNet-50 amplifies even fp32 summation-order noise ~160x on the way back
    # (measured 2.1e-2 worst on a BN bias of the image encoder, 3e-4 on text / head tensors); damped: 7.9e-3 / 3e-4
    lim = 2e-2 if "damped" in fname else 6e-2
    assert m["worst_norm"][1] < lim and m["worst_l2"][1] < lim, m
    assert m["by_group"]["text"] < 2e-3 and m["by_group"]["head"] < 2e-3, m
    assert m["bn:bn1.running_var"] < 1e-4 and m["bn:layer4.2.bn3.running_var"] < 1e-3, m


@pytest.mark.parametrize("fname", ["c0_full_size.npz", "c0_damped.npz"])
def test_c0_full_size_bf16(dev, fname):
    """The same goldens on the benchmarked path (bf16 storage, MFMA kernels). No bf16-storage implementation can meet
    1e-3 / 1e-4 here: the ORACLE ITSELF, run under the bf16 storage policy, is 0.59 (default init) / 0.095 (damped) from
    its own fp32 logits at this size — a random-init ResNet-50 with batch-statistics BatchNorm amplifies the 2^-9 storage
    rounding ~1.37x per bottleneck. What is asserted: the device's distance from the fp32 golden is of the
    size of the bf16-policy oracle's own distance (x2 + a small floor) for logits, loss and both encoder features — i.e.
    the deviation is bf16 storage, not the kernels. The kernels themselves are pinned tightly by the teacher-forced
    backward test and the per-layer forward test (tests/test_engines_gpu.py) and by the fp32 mode above."""
    m, pol = _c0_metrics(dev, "bf16", fname)
    print(f"C0 bf16 {fname}: device {m}\n   bf16-policy oracle vs fp32 oracle: {pol}")
    for k, floor in (("dlogits", 2e-2), ("text_feat", 1e-2), ("image_feat", 1e-2)):
        assert m[k] < 2.0 * pol[k] + floor, (k, m[k], pol[k])
    # the loss is 1-Lipschitz in the logits (mean CE): bounded by the logit distance, whatever sign pattern the draw has
    assert m["dloss"] < max(2.0 * pol["dloss"], pol["dlogits"]) + 1e-2, (m["dloss"], pol)


def test_c4_fp8_b128_workload(dev):
    """BASELINE.json configs[4] at its own workload — BERT-base + ResNet-50 + fusion head, B = 128, S = 128, 224x224,
    precision="fp8" (e4m3 operands in the text encoder's forward Linears, bf16 everything else) — against the ORACLE under the
    same policy (tests/golden/make_c4_golden.py: fp32 / bf16-policy / fp8-policy forwards of this very batch and weights):

      * text feature (BERT is smooth; no BatchNorm chaos): the device is within the format's own error of the fp8-policy oracle
        — closer to it than the fp8 policy is to the bf16 policy — so the device computes THAT quantized function;
      * first loss / logits: the image half of this random-init graph is chaotic under bf16 storage, so
        they are bounded as in test_c0_full_size_bf16: by the fp8-policy oracle's own distance from fp32 (x2 + floor);
      * four optimizer steps on the fixed batch: every loss finite, the last below the first."""
    from multimodal_sentiment_aanalysis_amd.fused import FusedTrainStep
    from util import synth_batch
    d = load("c4_fp8_b128.npz")
    torch.manual_seed(int(d["seed"]))
    model = mm.MultimodalTransformerModel(dropout=0.0)
    step = FusedTrainStep(model, dev, precision="fp8")
    image, ids, mask, labels = synth_batch(128, 128, 224, 224, 30522, seed=int(d["seed"]), dev=dev)
    assert torch.equal(labels.cpu(), d["labels"])
    model.train()
    with torch.no_grad():
        i, t = model.encoder.features(image, ids, mask)
    # (that forward updated the BatchNorm running statistics once more than the oracle's single forward: irrelevant below)
    t = t.float().cpu()
    t8, t16, t32 = d["text_feat_fp8"], d["text_feat_bf16"], d["text_feat_fp32"]
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    price = rel(t8, t16)
    print(f"text feature: device vs fp8-policy oracle {rel(t, t8):.3e}; vs bf16-policy {rel(t, t16):.3e}; "
          f"fp8-policy vs bf16-policy (price of e4m3) {price:.3e}; fp8-policy vs fp32 {rel(t8, t32):.3e}")
    # Two implementations of the SAME quantized function do not agree to bf16 resolution at this depth: e4m3 rounding is a step
    # function, so wherever their inputs differ by a bf16 ulp (2^-9) about 2^-9 / 2^-4 = 3 % of the elements land on the other
    # side of a rounding boundary and move by a whole e4m3 step — a random walk over the 48 quantized GEMMs of BERT-base (the
    # 2-layer engine test, test_bert_engine_fp8, agrees to a third of the format's error). What identifies the function is that
    # the device's quantization error is the ORACLE's quantization error: e_dev = device - bf16 policy and e_pol = fp8 policy -
    # bf16 policy point the same way (cosine 0.7 measured; an unrelated quantizer of the same strength would give ~0) and have
    # the same size, and the device is nearer to the fp8 policy than the fp8 policy is to bf16.
    e_dev, e_pol = t - t16, t8 - t16
    cos = (e_dev * e_pol).sum().item() / (e_dev.norm() * e_pol.norm()).item()
    print(f"quantization error: |e_dev| / |e_pol| = {(e_dev.norm() / e_pol.norm()).item():.3f}, cosine {cos:.3f}")
    assert cos > 0.5 and 0.7 < (e_dev.norm() / e_pol.norm()).item() < 1.4, (cos, e_dev.norm(), e_pol.norm())
    assert rel(t, t8) < 0.9 * price, "the device's text feature is not the fp8-policy function"
    losses = []
    logits0 = None
    for k in range(4):
        loss, logits = step.step(image, ids, mask, labels)
        if k == 0:
            logits0 = logits.detach().float().cpu()
        losses.append(loss.item())
    print("losses", losses, "oracle fp32 / bf16 / fp8:", float(d["loss_fp32"]), float(d["loss_bf16"]), float(d["loss_fp8"]))
    assert all(l == l and abs(l) < 1e4 for l in losses), losses
    assert losses[-1] < losses[0], losses
    ref32 = d["logits_fp32"]
    dl_dev = (logits0 - ref32).abs().max().item()
    dl_pol = (d["logits_fp8"] - ref32).abs().max().item()
    assert dl_dev < 2.0 * dl_pol + 2e-2, (dl_dev, dl_pol)
    dloss_pol = abs(float(d["loss_fp8"]) - float(d["loss_fp32"]))
    assert abs(losses[0] - float(d["loss_fp32"])) < max(2.0 * dloss_pol, dl_pol) + 1e-2, (losses[0], float(d["loss_fp32"]), dloss_pol)
    del step, model
    torch.cuda.empty_cache()


def test_n1_nt_xent_variant(dev):
    """The label-free NT-Xent of the reference's ME-MHACL script (ME-MHACL/train.py:47-66) = the two-view kernel with every
    sample its own class: against the golden made by CALLING the reference's own function (tests/golden/n1_nt_xent.npz,
    make_golden.py::gen_nt_xent) and, on more shapes, against the oracle's restatement and its autograd."""
    from multimodal_sentiment_aanalysis_amd.engine import nt_xent_loss
    from oracle import fusion as OF
    d = load("n1_nt_xent.npz")
    for tag in "abcd":
        x = d[f"ntxent.{tag}.z1"].to(dev).requires_grad_(True)
        y = d[f"ntxent.{tag}.z2"].to(dev).requires_grad_(True)
        loss = nt_xent_loss(x, y, float(d[f"ntxent.{tag}.T"]))
        loss.backward()
        ref = d[f"ntxent.{tag}.loss"].item()
        assert abs(loss.item() - ref) < 2e-5 * max(1.0, abs(ref)), (tag, loss.item(), ref)
        scale = max(d[f"ntxent.{tag}.dz1"].abs().max().item(), d[f"ntxent.{tag}.dz2"].abs().max().item())
        assert (x.grad.cpu() - d[f"ntxent.{tag}.dz1"]).abs().max().item() < 1e-4 * scale, tag
        assert (y.grad.cpu() - d[f"ntxent.{tag}.dz2"]).abs().max().item() < 1e-4 * scale, tag
    for B, D, T in ((16, 128, 0.5), (64, 128, 0.1), (5, 32, 0.5)):
        g = torch.Generator().manual_seed(B)
        z1, z2 = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
        a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
        ref = OF.nt_xent(a, b, T)
        ref.backward()
        x, y = z1.to(dev).requires_grad_(True), z2.to(dev).requires_grad_(True)
        loss = nt_xent_loss(x, y, T)
        loss.backward()
        assert abs(loss.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item())), (B, loss.item(), ref.item())
        scale = max(a.grad.abs().max().item(), b.grad.abs().max().item())
        assert (x.grad.cpu() - a.grad).abs().max().item() < 1e-4 * scale
        assert (y.grad.cpu() - b.grad).abs().max().item() < 1e-4 * scale
