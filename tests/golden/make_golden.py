"""Generates the golden fixtures under tests/golden/ (run in the BUILD container only; needs /root/reference and,
for the BERT fixtures, the `transformers` package):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

Fixtures hold plain arrays only (inputs, weights, expected outputs / gradients): no pickled modules, no reference
source. The reference classes are imported from /root/reference/MML_ZYC and CALLED to produce the expected values;
their encoder slots (eeg_net / eye_net / pps_net) are replaced by nn.Identity so that the real fusion code paths
(MultimodalModel.py:287-313 and :388-404) run on given 256-d feature vectors.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/MML_ZYC")
sys.dont_write_bytecode = True

import MultimodalModel as REF  # noqa: E402  (the reference's file)


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB, {len(out)} arrays")


def sd_arrays(module, prefix="w."):
    return {prefix + k: v for k, v in module.state_dict().items()}


def grads(module, prefix="g."):
    return {prefix + n: p.grad for n, p in module.named_parameters() if p.grad is not None}


def no_dropout(module):
    for m in module.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0


def rnd(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


# ------------------------------------------------------------------------------------------------ A1
def gen_cross_modal():
    torch.manual_seed(0)
    m = REF.CrossModalTransformer()
    with torch.no_grad():  # make zero-initialised biases / unit norms non-trivial
        m.multihead_attn.in_proj_bias.normal_(0, 0.1)
        m.multihead_attn.out_proj.bias.normal_(0, 0.1)
        m.norm.weight.add_(0.1 * torch.randn(256))
        m.norm.bias.add_(0.1 * torch.randn(256))
    for tag, Lk in (("l1", 1), ("l4", 4)):
        B = 16
        q = rnd(B, 256, seed=1).requires_grad_(True)
        k = (rnd(B, 256, seed=2) if Lk == 1 else rnd(B, Lk, 256, seed=2)).requires_grad_(True)
        v = (rnd(B, 256, seed=3) if Lk == 1 else rnd(B, Lk, 256, seed=3)).requires_grad_(True)
        w = rnd(B, 256, seed=4)
        m.zero_grad()
        out = m(q, k, v)
        (out * w).sum().backward()
        npz(f"a1_cross_modal_{tag}.npz", q=q, k=k, v=v, wgt=w, out=out, dq=q.grad, dk=k.grad, dv=v.grad,
            **sd_arrays(m), **grads(m))


# ------------------------------------------------------------------------------------------------ A2
def gen_mm_fusion():
    torch.manual_seed(1)
    m = REF.MultiModalEncoder()
    m.eeg_net, m.eye_net, m.pps_net = nn.Identity(), nn.Identity(), nn.Identity()
    with torch.no_grad():
        m.multihead_attn.in_proj_bias.normal_(0, 0.1)
        m.fusion_mlp[2].weight.add_(0.1 * torch.randn(256))
        m.fusion_mlp[2].bias.add_(0.1 * torch.randn(256))
    B = 16
    feats = [(3.0 * rnd(B, 256, seed=10 + i)).requires_grad_(True) for i in range(3)]
    w = rnd(B, 256, seed=20)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    arrays = {"w." + k: v for k, v in sd0.items()}
    for mode in ("train", "eval"):
        m.load_state_dict(sd0)
        m.train(mode == "train")
        m.zero_grad()
        for f in feats:
            f.grad = None
        out = m(*feats)
        (out * w).sum().backward()
        arrays.update({f"{mode}.out": out, **{f"{mode}.df{i}": feats[i].grad for i in range(3)}})
        arrays.update({f"{mode}.g.{n}": p.grad for n, p in m.named_parameters() if p.grad is not None})
        if mode == "train":
            arrays.update({"train.post." + k: v.clone() for k, v in m.state_dict().items() if "running" in k})
    # mean-pool twin (ME-MHACL/model.py:69-73) through the same reference attention module
    m.load_state_dict(sd0)
    m.eval()
    seq = torch.stack([torch.nn.functional.normalize(f.detach(), dim=-1) for f in feats], dim=0)
    attn, _ = m.multihead_attn(seq, seq, seq)
    arrays["eval.mean_pool_out"] = m.fusion_mlp(attn.mean(dim=0))
    npz("a2_mm_fusion.npz", wgt=w, **{f"f{i}": feats[i] for i in range(3)}, **arrays)


# ------------------------------------------------------------------------------------------------ A1 + A4 composed
def ref_head_model(seed):
    torch.manual_seed(seed)
    m = REF.MultimodalTransformerModel()
    m.eeg_net, m.eye_net, m.pps_net = nn.Identity(), nn.Identity(), nn.Identity()
    no_dropout(m)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, (nn.BatchNorm1d, nn.LayerNorm)):
                mod.weight.add_(0.1 * torch.randn_like(mod.weight))
                mod.bias.add_(0.1 * torch.randn_like(mod.bias))
    return m


def gen_fusion_head():
    m = ref_head_model(2)
    B = 16
    feats = [rnd(B, 256, seed=30 + i).requires_grad_(True) for i in range(3)]
    wa, wv = rnd(B, 3, seed=40), rnd(B, 3, seed=41)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    arrays = {"w." + k: v for k, v in sd0.items()}
    for mode in ("train", "eval"):
        m.load_state_dict(sd0)
        m.train(mode == "train")
        m.zero_grad()
        for f in feats:
            f.grad = None
        arousal, valence = m(*feats)  # labels=None -> (arousal, valence), MultimodalModel.py:319-320
        ((arousal * wa).sum() + (valence * wv).sum()).backward()
        arrays.update({f"{mode}.arousal": arousal, f"{mode}.valence": valence,
                       **{f"{mode}.df{i}": feats[i].grad for i in range(3)}})
        arrays.update({f"{mode}.g.{n}": p.grad for n, p in m.named_parameters() if p.grad is not None})
        if mode == "train":
            arrays.update({"train.post." + k: v.clone() for k, v in m.state_dict().items() if "running" in k})
    npz("a4_fusion_head.npz", wa=wa, wv=wv, **{f"f{i}": feats[i] for i in range(3)}, **arrays)


# ------------------------------------------------------------------------------------------------ A5, A6
def gen_small_heads():
    torch.manual_seed(3)
    c = REF.Classifier()
    p = REF.ProjectionHead()
    no_dropout(c)
    no_dropout(p)
    B = 16
    x = rnd(B, 256, seed=50).requires_grad_(True)
    wa, wv, wz = rnd(B, 3, seed=51), rnd(B, 3, seed=52), rnd(B, 128, seed=53)
    c.train()
    a, v = c(x)
    ((a * wa).sum() + (v * wv).sum()).backward()
    arr = dict(x=x, wa=wa, wv=wv, wz=wz, cls_a=a, cls_v=v, cls_dx=x.grad.clone())
    arr.update({"cls.w." + k: t for k, t in c.state_dict().items()})
    arr.update({"cls.g." + n: t.grad for n, t in c.named_parameters()})
    x.grad = None
    sd0 = {k: t.clone() for k, t in p.state_dict().items()}
    arr.update({"proj.w." + k: t for k, t in sd0.items()})
    p.train()
    z = p(x)
    (z * wz).sum().backward()
    arr.update(proj_z=z, proj_dx=x.grad.clone())
    arr.update({"proj.g." + n: t.grad for n, t in p.named_parameters()})
    arr.update({"proj.post." + k: t.clone() for k, t in p.state_dict().items() if "running" in k})
    # A6: CrossEntropyLoss (Trainer.py:17,68)
    logits = rnd(B, 3, seed=60).requires_grad_(True)
    labels = torch.randint(0, 3, (B,), generator=torch.Generator().manual_seed(61))
    loss = nn.CrossEntropyLoss()(logits, labels)
    loss.backward()
    arr.update(ce_logits=logits, ce_labels=labels, ce_loss=loss, ce_dlogits=logits.grad)
    npz("a5_a6_heads_ce.npz", **arr)


# ------------------------------------------------------------------------------------------------ A7
def gen_train_step():
    """Two Trainer.train_epoch bodies (Trainer.py:59-81): zero_grad, forward, CE, backward, clip_grad_norm_(1.0),
    AdamW(lr 1e-4, weight_decay 0.01).step(), on the reference fusion head fed with fixed feature vectors
    (the older single-head contract: only the arousal logits enter the loss)."""
    m = ref_head_model(4)
    m.train()
    B = 16
    feats = [rnd(B, 256, seed=70 + i) for i in range(3)]
    labels = torch.randint(0, 3, (B,), generator=torch.Generator().manual_seed(73))
    arrays = {"w0." + k: v.clone() for k, v in m.state_dict().items()}
    params = [p for n, p in m.named_parameters()]
    opt = torch.optim.AdamW(params, lr=0.0001, weight_decay=0.01)
    crit = nn.CrossEntropyLoss()
    for step in (1, 2):
        opt.zero_grad()
        arousal, _ = m(*feats)
        loss = crit(arousal, labels)
        loss.backward()
        total = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        arrays[f"loss{step}"] = loss.detach()
        arrays[f"norm{step}"] = total
        arrays.update({f"w{step}." + k: v.clone() for k, v in m.state_dict().items()})
    npz("a7_train_step.npz", labels=labels, **{f"f{i}": feats[i] for i in range(3)}, **arrays)


# ------------------------------------------------------------------------------------------------ E1 (not in reference)
def gen_bert():
    from transformers import BertConfig, BertModel
    import multimodal_sentiment_aanalysis_amd as mm
    from multimodal_sentiment_aanalysis_amd.engine import BERT_BASE, BertTextNet
    mini = dict(hidden=128, layers=2, heads=2, intermediate=512, vocab=1000, max_pos=64, type_vocab=2, ln_eps=1e-12)

    def hf_model(cfg):
        c = BertConfig(vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"],
                       num_attention_heads=cfg["heads"], intermediate_size=cfg["intermediate"],
                       max_position_embeddings=cfg["max_pos"], type_vocab_size=cfg["type_vocab"],
                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=cfg["ln_eps"])
        return BertModel(c).eval()

    # mini config with stored weights (HF initialisation, masked and unmasked)
    torch.manual_seed(5)
    hf = hf_model(mini)
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if n.endswith("bias") or "LayerNorm" in n:
                p.add_(0.05 * torch.randn_like(p))
    ids = torch.randint(0, mini["vocab"], (4, 32), generator=torch.Generator().manual_seed(6))
    mask = torch.ones(4, 32, dtype=torch.long)
    mask[1, 20:] = 0
    mask[3, 7:] = 0
    with torch.no_grad():
        o1 = hf(input_ids=ids)
        o2 = hf(input_ids=ids, attention_mask=mask)
    npz("e1_bert_mini.npz", ids=ids, mask=mask, hidden_nomask=o1.last_hidden_state, pooled_nomask=o1.pooler_output,
        hidden_mask=o2.last_hidden_state, pooled_mask=o2.pooler_output,
        **{"w." + k: v for k, v in hf.state_dict().items() if "position_ids" not in k})

    # BERT-base: weights regenerated from a seed by the product's initialiser on both sides; only outputs stored
    torch.manual_seed(1234)
    net = BertTextNet(BERT_BASE)
    sd = {k[len("bert."):]: v for k, v in net.state_dict().items() if k.startswith("bert.")}
    hfb = hf_model(BERT_BASE)
    missing = hfb.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if "position_ids" not in k], missing
    ids = torch.randint(0, BERT_BASE["vocab"], (16, 128), generator=torch.Generator().manual_seed(7))
    ids[:, 0] = 101
    with torch.no_grad():
        ob = hfb(input_ids=ids)
    npz("e1_bert_base_seed1234.npz", ids=ids, pooled=ob.pooler_output, hidden_cls=ob.last_hidden_state[:, 0],
        hidden_row5=ob.last_hidden_state[5])


# ------------------------------------------------------------------------------------------------ E2 (not in reference)
def tv_to_hf_resnet(name):
    """torchvision-style ResNet key (the oracle's / the product's names, no prefix) -> transformers.ResNetModel key."""
    sub = {"weight": "weight", "bias": "bias", "running_mean": "running_mean", "running_var": "running_var",
           "num_batches_tracked": "num_batches_tracked"}
    parts = name.split(".")
    if parts[0] == "conv1":
        return "embedder.embedder.convolution.weight"
    if parts[0] == "bn1":
        return "embedder.embedder.normalization." + parts[1]
    s, b = int(parts[0][len("layer"):]) - 1, int(parts[1])
    base = f"encoder.stages.{s}.layers.{b}."
    if parts[2] == "downsample":
        return base + ("shortcut.convolution.weight" if parts[3] == "0" else "shortcut.normalization." + parts[4])
    k = int(parts[2][-1]) - 1
    return base + f"layer.{k}." + ("convolution.weight" if parts[2].startswith("conv") else "normalization." + parts[3])


def gen_resnet():
    """E2 pin: transformers.ResNetModel (config-only construction, local code, no fetch) is ResNet v1.5 with
    downsample_in_bottleneck=False (stride on the 3x3) — the same public architecture oracle/resnet.py restates. Two fixtures:
    a mini net with STORED weights (train-mode pooled output + per-stage maps, every parameter gradient of a fixed scalar,
    BN running statistics after that forward, eval-mode output with those statistics), and full ResNet-50 with weights
    REGENERATED from a seed by the product's initialiser on both sides (outputs + strided gradient samples only).
    transformers.ResNetModel is run in FLOAT64 (weights and inputs are exact fp32 values), so the fixture is the exact function:
    a random-init BatchNorm ResNet amplifies fp32 summation-order noise to ~1e-2 in the gradients, which
    would otherwise be the floor of every comparison against it."""
    from transformers import ResNetConfig, ResNetModel
    from multimodal_sentiment_aanalysis_amd.engine import RESNET50, ResNetImageNet
    from oracle.resnet import resnet_param_shapes

    def hf_model(cfg):
        c = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[4 * w for w in cfg["widths"]],
                         depths=list(cfg["blocks"]), layer_type="bottleneck", hidden_act="relu",
                         downsample_in_first_stage=False, downsample_in_bottleneck=False)
        return ResNetModel(c)

    def load_tv(hf, sd_tv):
        ocfg = dict(blocks=tuple(hf.config.depths), widths=tuple(h // 4 for h in hf.config.hidden_sizes), expansion=4)
        mapped = {tv_to_hf_resnet(n): sd_tv[n] for n, _, _ in resnet_param_shapes(ocfg)}
        res = hf.load_state_dict(mapped, strict=True)
        assert not res.missing_keys and not res.unexpected_keys, res
        return ocfg

    def hf_to_tv(hf, ocfg):
        sd = hf.state_dict()
        return {n: sd[tv_to_hf_resnet(n)].clone() for n, _, _ in resnet_param_shapes(ocfg)}

    # ---- mini net, stored weights ---------------------------------------------------------------------------------
    mini = dict(blocks=(1, 2, 1, 1), widths=(64, 64, 128, 128))
    torch.manual_seed(21)
    hf = hf_model(mini)
    ocfg = dict(blocks=mini["blocks"], widths=mini["widths"], expansion=4)
    with torch.no_grad():  # non-trivial BN affine parameters and running statistics
        for n, p in hf.named_parameters():
            if "normalization" in n:
                p.add_(0.1 * torch.randn_like(p))
        for n, b in hf.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(0.1 * torch.randn_like(b))
            if n.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand_like(b))
    w0 = hf_to_tv(hf, ocfg)
    image = rnd(4, 3, 96, 96, seed=22)
    wgt = rnd(4, 512, seed=23)
    hf.double().train()
    out = hf(pixel_values=image.double(), output_hidden_states=True)
    pooled = out.pooler_output.flatten(1)
    (pooled * wgt.double()).sum().backward()
    g = {n: dict(hf.named_parameters())[tv_to_hf_resnet(n)].grad.clone()
         for n, _, buf in resnet_param_shapes(ocfg) if not buf}
    w1 = hf_to_tv(hf, ocfg)  # running statistics after one train-mode forward
    hf.eval()
    with torch.no_grad():
        pooled_eval = hf(pixel_values=image.double()).pooler_output.flatten(1)
    npz("e2_resnet_mini.npz", image=image, wgt=wgt, pooled_train=pooled, pooled_eval=pooled_eval,
        **{f"stage{i}": h.float() for i, h in enumerate(out.hidden_states)},
        **{"w." + k: v for k, v in w0.items()}, **{"g." + k: v.float() for k, v in g.items()},
        **{"w1." + k: v for k, v in w1.items() if "running" in k or "num_batches" in k})

    # ---- ResNet-50, weights regenerated from a seed by the product's initialiser ------------------------------------
    torch.manual_seed(1234)
    net = ResNetImageNet(RESNET50)
    sd = {k[len("resnet."):]: v.detach().clone().contiguous() for k, v in net.state_dict().items() if k.startswith("resnet.")}
    hfb = hf_model(RESNET50)
    ocfg = load_tv(hfb, sd)
    assert sum(p.numel() for p in hfb.parameters()) == 23508032
    image = rnd(4, 3, 224, 224, seed=24)
    wgt = rnd(4, 2048, seed=25)
    hfb.double().train()
    out = hfb(pixel_values=image.double())
    pooled = out.pooler_output.flatten(1)
    (pooled * wgt.double()).sum().backward()
    arrays = {}
    hp = dict(hfb.named_parameters())
    for n, _, buf in resnet_param_shapes(ocfg):
        if buf:
            continue
        gr = hp[tv_to_hf_resnet(n)].grad.reshape(-1)
        stride = max(1, gr.numel() // 512)
        arrays["gn." + n] = gr.norm()
        arrays["gs." + n] = gr[::stride][:512].clone()
    hfb.eval()
    with torch.no_grad():
        pooled_eval = hfb(pixel_values=image.double()).pooler_output.flatten(1)
    npz("e2_resnet50_seed1234.npz", image_seed=torch.tensor(24), wgt=wgt, pooled_train=pooled, pooled_eval=pooled_eval,
        last_map_b0=out.last_hidden_state[0, ::64], **arrays)


# ------------------------------------------------------------------------------------------------ N1
def gen_contrastive():
    """N1: the reference's two contrastive losses, CALLED: MultimodalTransformerModel.compute_contrastive_loss
    (MultimodalModel.py:232-260, learnable temperature; the forward calls it with feat1 is feat2, :272-284) and
    train.contrastive_loss (train.py:16-40, two views, T = 0.1). Stored: inputs, loss, input gradients, dT."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_train", "/root/reference/MML_ZYC/train.py")
    ref_train = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_train)
    m = ref_head_model(4)
    out = {}
    for tag, B, D, T, same in (("a", 16, 256, 0.01, True), ("b", 16, 256, 0.5, False), ("c", 64, 256, 0.07, True),
                               ("d", 5, 128, 0.2, False)):
        with torch.no_grad():
            m.temperature.fill_(T)
        f1 = rnd(B, D, seed=60 + B).requires_grad_(True)
        f2 = f1 if same else rnd(B, D, seed=61 + B).requires_grad_(True)
        labels = torch.randint(0, 3, (B,), generator=torch.Generator().manual_seed(70 + B))
        if tag == "d":
            labels[:] = torch.tensor([0, 1, 2, 1, 1])  # rows 0 and 2 have no positive
        m.temperature.grad = None
        loss = m.compute_contrastive_loss(f1, f2, labels)
        loss.backward()
        out.update({f"infonce.{tag}.f1": f1, f"infonce.{tag}.labels": labels, f"infonce.{tag}.T": torch.tensor(T),
                    f"infonce.{tag}.loss": loss, f"infonce.{tag}.df1": f1.grad, f"infonce.{tag}.dT": m.temperature.grad,
                    f"infonce.{tag}.same": torch.tensor(int(same))})
        if not same:
            out.update({f"infonce.{tag}.f2": f2, f"infonce.{tag}.df2": f2.grad})
    for tag, B, D in (("a", 16, 128), ("b", 64, 128), ("c", 6, 32)):
        z1 = rnd(B, D, seed=80 + B).requires_grad_(True)
        z2 = rnd(B, D, seed=81 + B).requires_grad_(True)
        labels = torch.randint(0, 3, (B,), generator=torch.Generator().manual_seed(90 + B))
        loss = ref_train.contrastive_loss(z1, z2, labels)
        loss.backward()
        out.update({f"supcon.{tag}.z1": z1, f"supcon.{tag}.z2": z2, f"supcon.{tag}.labels": labels,
                    f"supcon.{tag}.loss": loss, f"supcon.{tag}.dz1": z1.grad, f"supcon.{tag}.dz2": z2.grad})
    npz("n1_contrastive.npz", **out)


def gen_nt_xent():
    """N1, label-free variant: the NT-Xent loss of the reference's ME-MHACL script (MML_ZYC/ME-MHACL/train.py:47-66), CALLED. The
    file is a script whose top level loads a dataset and trains, so it cannot be imported; its `contrastive_loss` is a top-level
    `def` that uses only torch and torch.nn.functional: the function's own node is lifted out of the parsed file (ast) and
    compiled alone, in this container, at generation time — nothing else of the script runs, and no source text is stored.
    Stored: inputs, temperature, loss, input gradients."""
    import ast
    import torch.nn.functional as F
    path = "/root/reference/MML_ZYC/ME-MHACL/train.py"
    tree = ast.parse(open(path).read(), filename=path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "contrastive_loss"]
    assert len(fn) == 1, "ME-MHACL/train.py: expected one top-level contrastive_loss"
    ns = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    ref_nt_xent = ns["contrastive_loss"]
    out = {}
    for tag, B, D, T in (("a", 16, 128, 0.5), ("b", 64, 128, 0.1), ("c", 5, 32, 0.07), ("d", 32, 256, 1.0)):
        z1 = rnd(B, D, seed=100 + B).requires_grad_(True)
        z2 = rnd(B, D, seed=101 + B).requires_grad_(True)
        loss = ref_nt_xent(z1, z2, T)
        loss.backward()
        out.update({f"ntxent.{tag}.z1": z1, f"ntxent.{tag}.z2": z2, f"ntxent.{tag}.T": torch.tensor(T),
                    f"ntxent.{tag}.loss": loss, f"ntxent.{tag}.dz1": z1.grad, f"ntxent.{tag}.dz2": z2.grad})
    npz("n1_nt_xent.npz", **out)


# ------------------------------------------------------------------------------------------------ N2
def gen_multitask_phases():
    """The reference's own MultiTaskTrainer (dataLoader/MultiTaskTrainer.py) CALLED on the identity-encoder fusion head: one
    epoch of phase 2 (two batches: arousal CE; everything but the valence head trains) and then one epoch of phase 3 (two
    batches: valence CE; cross-attention, weighting and fusion modules are trainable but only the valence head is handed to
    the optimizer, so their gradients are never zeroed, keep accumulating, and enter every clip norm), plus evaluate().
    Stored: inputs, labels, the state_dict before / after each epoch, the epoch metrics — plain arrays."""
    import importlib.util
    os.environ.setdefault("MPLBACKEND", "Agg")
    spec = importlib.util.spec_from_file_location("ref_mtt", "/root/reference/MML_ZYC/dataLoader/MultiTaskTrainer.py")
    ref_mtt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_mtt)
    from torch.utils.data import DataLoader, TensorDataset
    m = ref_head_model(6)
    n, bs = 32, 16
    feats = [rnd(n, 256, seed=170 + i) for i in range(3)]
    arousal = torch.randint(0, 3, (n,), generator=torch.Generator().manual_seed(173))
    valence = torch.randint(0, 3, (n,), generator=torch.Generator().manual_seed(174))
    ds = TensorDataset(feats[0], feats[1], feats[2], arousal, valence)
    train, test = DataLoader(ds, batch_size=bs, shuffle=False), DataLoader(ds, batch_size=bs, shuffle=False)
    arrays = {"w0." + k: v.clone() for k, v in m.state_dict().items()}
    tr = ref_mtt.MultiTaskTrainer(m, train, test, device="cpu")
    # record what the reference's own `torch.nn.utils.clip_grad_norm_(self.model.parameters(), 1.0)` calls return (the total
    # norm over every parameter that holds a gradient — in phase 3 that includes the STALE gradient the frozen arousal head
    # kept from phase 2): the real function runs, wrapped only to log its return value
    real_clip, norms = torch.nn.utils.clip_grad_norm_, []

    def logging_clip(*a, **k):
        total = real_clip(*a, **k)
        norms.append(float(total))
        return total

    torch.nn.utils.clip_grad_norm_ = logging_clip
    try:
        r2 = tr.train_epoch_phase2(1)
        arrays.update({"w_p2." + k: v.clone() for k, v in m.state_dict().items()})
        e2 = tr.evaluate()
        m.train()
        r3 = tr.train_epoch_phase3(1)
    finally:
        torch.nn.utils.clip_grad_norm_ = real_clip
    arrays["clip_norms"] = torch.tensor(norms, dtype=torch.float64)  # 2 batches of phase 2, then 2 of phase 3
    print("clip norms per batch:", norms)
    arrays.update({"w_p3." + k: v.clone() for k, v in m.state_dict().items()})
    e3 = tr.evaluate()
    for tag, r in (("train_p2", r2), ("eval_p2", e2), ("train_p3", r3), ("eval_p3", e3)):
        for k, v in r.items():
            arrays[f"{tag}.{k}"] = torch.tensor(float(v), dtype=torch.float64)
    # which tensors each phase's optimizer owned / which were trainable (for the test's bookkeeping checks)
    moved2 = [k for k in m.state_dict() if not torch.equal(arrays["w0." + k], arrays["w_p2." + k]) and "running" not in k and "num_batches" not in k]
    moved3 = [k for k in m.state_dict() if not torch.equal(arrays["w_p2." + k], arrays["w_p3." + k]) and "running" not in k and "num_batches" not in k]
    print("phase 2 moved", len(moved2), "tensors; phase 3 moved", len(moved3), ":", sorted({k.split('.')[0] for k in moved3}))
    npz("n2_multitask_phases.npz", arousal=arousal, valence=valence, **{f"f{i}": feats[i] for i in range(3)}, **arrays)


if __name__ == "__main__":
    which = sys.argv[1:] or ["a1", "a2", "a4", "a5", "a7", "e1", "e2", "n1", "n1x", "n2"]
    if "n2" in which:
        gen_multitask_phases()
    if "n1" in which:
        gen_contrastive()
    if "n1x" in which:
        gen_nt_xent()
    if "a1" in which:
        gen_cross_modal()
    if "a2" in which:
        gen_mm_fusion()
    if "a4" in which:
        gen_fusion_head()
    if "a5" in which:
        gen_small_heads()
    if "a7" in which:
        gen_train_step()
    if "e1" in which:
        gen_bert()
    if "e2" in which:
        gen_resnet()
