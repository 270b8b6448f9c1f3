"""Whole text+image model and the reference train step, on the CPU (TEST INFRASTRUCTURE ONLY).

Composition (SURVEY.md §8a "recommended composition"): BERT pooled -> Linear(768,256) = text feature,
ResNet pooled -> Linear(2048,256) = image feature (encoder slot, MultimodalModel.py:264-266), then every
reference fusion module with its own shapes: ME-MHACL 8-head fusion over the 2-token modality sequence
(MultimodalModel.py:388-404), bidirectional CrossModalTransformer (:287-297), dynamic weighting + fusion MLP +
arousal head (:298-313), CE (Trainer.py:68), clip + AdamW (Trainer.py:79-81).
"""
import math

import torch

from . import fusion as F_
from .bert import bert_forward
from .policy import FP32
from .resnet import resnet_forward


def encoder_features(sd, image, token_ids, attention_mask, cfg, training, pol=FP32, prefix="encoder."):
    """text/image features [B,256] (fp32): the two encoder-slot outputs."""
    ids = token_ids.long()
    _, pooled = bert_forward(sd, prefix + "text_net.bert.", ids, attention_mask, cfg["bert"], pol)
    t = pooled @ pol.qw(sd[prefix + "text_net.proj.weight"]).t() + sd[prefix + "text_net.proj.bias"]
    feat = resnet_forward(sd, prefix + "image_net.resnet.", image.float(), cfg["resnet"], training, pol)
    i = feat @ pol.qw(sd[prefix + "image_net.proj.weight"]).t() + sd[prefix + "image_net.proj.bias"]
    return t, i


def multimodal_encoder_forward(sd, image, token_ids, attention_mask, cfg, training, pol=FP32, prefix=""):
    """MultiModalEncoder.forward (MultimodalModel.py:383-406) with (image, text) modalities."""
    t, i = encoder_features(sd, image, token_ids, attention_mask, cfg, training, pol, prefix)
    return F_.mm_fusion(sd, prefix.rstrip("."), [t, i], training, cfg.get("mm_heads", 8), cfg.get("pool", "max"))


def model_forward(sd, image, token_ids, attention_mask, cfg, training, pol=FP32):
    """MultimodalTransformerModel.forward (text+image): returns logits [B,3] and a dict of intermediates."""
    t, i = encoder_features(sd, image, token_ids, attention_mask, cfg, training, pol)
    mm = F_.mm_fusion(sd, "encoder", [t, i], training, cfg.get("mm_heads", 8), cfg.get("pool", "max"))
    i_enh = F_.cross_modal_transformer(sd, "cross_attn_t2i", t, i, i, cfg.get("cross_heads", 4))
    t_enh = F_.cross_modal_transformer(sd, "cross_attn_i2t", i, t, t, cfg.get("cross_heads", 4))
    logits, fused = F_.weighted_fusion_logits(sd, mm, t, i, i_enh, t_enh, training)
    return logits, dict(text=t, image=i, mm=mm, i_enh=i_enh, t_enh=t_enh, fused=fused)


def clip_grad_norm(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_(params, 1.0): Trainer.py:80 (global L2 norm, coef clamped to 1)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adamw_step(params, grads, state, lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.AdamW defaults as used at Trainer.py:19-21 (decoupled decay on every parameter)."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    b1, b2 = betas
    for n, p in params.items():
        g = grads[n]
        m = state.setdefault("m." + n, torch.zeros_like(p))
        v = state.setdefault("v." + n, torch.zeros_like(p))
        p.mul_(1 - lr * weight_decay)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = v.sqrt() / math.sqrt(1 - b2 ** t) + eps
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))


def train_step(sd, param_names, image, token_ids, attention_mask, labels, cfg, opt_state, pol=FP32, lr=1e-4,
               weight_decay=0.01, max_norm=1.0):
    """One Trainer.train_epoch body (Trainer.py:59-81): zero_grad, forward, CE, backward, clip, AdamW.
    Mutates sd (parameters, BN buffers) in place. Returns (loss, logits, grads before clipping, total_norm)."""
    params = {n: sd[n].detach().requires_grad_(True) for n in param_names}
    work = dict(sd)
    work.update(params)
    logits, _ = model_forward(work, image, token_ids, attention_mask, cfg, True, pol)
    loss = F_.cross_entropy(logits, labels)
    grads_t = torch.autograd.grad(loss, [params[n] for n in param_names], allow_unused=True)
    # parameters that received no gradient are skipped entirely (no decay either), as torch.optim.AdamW does for grad None
    grads = {n: g for n, g in zip(param_names, grads_t) if g is not None}
    raw = {n: g.clone() for n, g in grads.items()}
    total = clip_grad_norm(list(grads.values()), max_norm)
    with torch.no_grad():
        adamw_step({n: sd[n] for n in grads}, grads, opt_state, lr, weight_decay)
    return loss.detach(), logits.detach(), raw, total
