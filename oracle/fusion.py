"""Fusion head of the reference, restated with elementary tensor ops (TEST INFRASTRUCTURE ONLY).

All functions take a flat `sd` (name -> tensor, the product model's state_dict layout) and a `prefix`.
Citations are to /root/reference/MML_ZYC/.
"""
import math

import torch


# ------------------------------------------------------------------------------------------------ building blocks
def linear(sd, p, x):
    """nn.Linear: y = x W^T + b."""
    return x @ sd[p + ".weight"].t() + sd[p + ".bias"]


def gelu(x):
    """nn.GELU() default (exact erf form): MultimodalModel.py:173,182,187,195."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def layer_norm(sd, p, x, eps=1e-5):
    """nn.LayerNorm(256): MultimodalModel.py:122,149 (biased variance over the last dim)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * sd[p + ".weight"] + sd[p + ".bias"]


def batch_norm_1d(sd, p, x, training, eps=1e-5, momentum=0.1):
    """nn.BatchNorm1d on [B, C]: MultimodalModel.py:181,186,194,380,419,423 (eps 1e-5, momentum 0.1).
    Training: batch mean / biased variance normalise; running stats take the UNBIASED variance. Updates the
    running buffers in `sd` in place, like the module does."""
    if training:
        n = x.shape[0]
        mean = x.mean(0)
        var = ((x - mean) ** 2).mean(0)
        with torch.no_grad():
            sd[p + ".running_mean"].mul_(1 - momentum).add_(momentum * mean.detach())
            unbiased = var.detach() * (n / max(n - 1, 1))
            sd[p + ".running_var"].mul_(1 - momentum).add_(momentum * unbiased)
            if p + ".num_batches_tracked" in sd:
                sd[p + ".num_batches_tracked"].add_(1)
    else:
        mean, var = sd[p + ".running_mean"], sd[p + ".running_var"]
    return (x - mean) / torch.sqrt(var + eps) * sd[p + ".weight"] + sd[p + ".bias"]


def l2_normalize(x, eps=1e-12):
    """F.normalize(x, dim=-1): MultimodalModel.py:388-390 (x / max(||x||_2, eps))."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def multihead_attention(sd, p, query, key, value, num_heads):
    """torch.nn.MultiheadAttention forward (packed in-projection, no masks, dropout 0), batch-first layout
    [B, L, E]. Row blocks of in_proj_weight are Wq | Wk | Wv; q is scaled by 1/sqrt(head_dim) before QK^T;
    softmax over keys; heads concatenated; out_proj. Relied on at MultimodalModel.py:112-116,139-143
    (4 heads) and :374-375,397 (8 heads); torch/nn/functional.py::multi_head_attention_forward."""
    E = query.shape[-1]
    W, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = query @ W[:E].t() + b[:E]
    k = key @ W[E:2 * E].t() + b[E:2 * E]
    v = value @ W[2 * E:].t() + b[2 * E:]
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    hd = E // num_heads
    q = q.view(B, Lq, num_heads, hd).transpose(1, 2) * (1.0 / math.sqrt(hd))
    k = k.view(B, Lk, num_heads, hd).transpose(1, 2)
    v = v.view(B, Lk, num_heads, hd).transpose(1, 2)
    attn = torch.softmax(q @ k.transpose(-1, -2), dim=-1)  # [B, h, Lq, Lk]
    ctx = (attn @ v).transpose(1, 2).reshape(B, Lq, E)
    out = ctx @ sd[p + ".out_proj.weight"].t() + sd[p + ".out_proj.bias"]
    return out, attn.mean(1)  # head-averaged weights (need_weights=True default; discarded by the reference)


# ------------------------------------------------------------------------------------------------ A1
def cross_modal_transformer(sd, p, query, key, value, num_heads=4):
    """CrossModalTransformer.forward, MultimodalModel.py:124-149: 2-D inputs are unsqueezed to length-1
    sequences (:132-137); attn = MHA(q,k,v) (:139-143); g = sigmoid(W_g [q; attn] + b) (:147);
    out = LayerNorm(g*q + (1-g)*attn) (:148-149). Sequence inputs [B, L, E] keep Lk > 1 for key/value but the
    squeeze(1) at :144,:147 requires Lq == 1."""
    q3 = query.unsqueeze(1) if query.dim() == 2 else query
    k3 = key.unsqueeze(1) if key.dim() == 2 else key
    v3 = value.unsqueeze(1) if value.dim() == 2 else value
    attn, _ = multihead_attention(sd, p + ".multihead_attn", q3, k3, v3, num_heads)
    attn = attn.squeeze(1)
    q2 = q3.squeeze(1)
    gate = torch.sigmoid(linear(sd, p + ".gate.0", torch.cat([q2, attn], dim=1)))
    return layer_norm(sd, p + ".norm", gate * q2 + (1 - gate) * attn)


# ------------------------------------------------------------------------------------------------ A2
def mm_fusion(sd, p, feats, training, num_heads=8, pool="max"):
    """Fusion part of MultiModalEncoder.forward, MultimodalModel.py:388-404: L2-normalise each modality
    vector (:388-390), stack to a length-M sequence (:393-396), 8-head self-attention (:397), max over the
    modality axis (:401; ME-MHACL/model.py:73 uses mean), then Linear -> ReLU -> BatchNorm1d (:377-381, :404).
    `feats` is a list of M tensors [B, 256]."""
    seq = torch.stack([l2_normalize(f) for f in feats], dim=1)  # [B, M, E]
    attn, _ = multihead_attention(sd, p + ".multihead_attn", seq, seq, seq, num_heads)
    fused = attn.max(dim=1)[0] if pool == "max" else attn.mean(dim=1)
    h = torch.relu(linear(sd, p + ".fusion_mlp.0", fused))
    return batch_norm_1d(sd, p + ".fusion_mlp.2", h, training)


# ------------------------------------------------------------------------------------------------ A4
def dynamic_weights(sd, p, f1, f2, f3):
    """attention_weights, MultimodalModel.py:171-176, 299-301: softmax(W2 GELU(W1 [f1;f2;f3]))."""
    h = gelu(linear(sd, p + ".0", torch.cat([f1, f2, f3], dim=1)))
    return torch.softmax(linear(sd, p + ".2", h), dim=1)


def fusion_mlp(sd, p, x, training):
    """fusion, MultimodalModel.py:179-189: Linear(768,256) BN GELU Dropout Linear(256,128) BN GELU Dropout
    (dropout is the identity here: parity runs use p = 0 / eval)."""
    h = gelu(batch_norm_1d(sd, p + ".1", linear(sd, p + ".0", x), training))
    return gelu(batch_norm_1d(sd, p + ".5", linear(sd, p + ".4", h), training))


def arousal_head(sd, p, x, training):
    """arousal_head, MultimodalModel.py:192-199: Linear(128,128) BN GELU Dropout Linear(128,3)."""
    h = gelu(batch_norm_1d(sd, p + ".1", linear(sd, p + ".0", x), training))
    return linear(sd, p + ".4", h)


def valence_head(sd, p, x, training):
    """valence_head, MultimodalModel.py:200-225: 128->256->256->128->64->3 with BN+GELU+Dropout between."""
    h = x
    for lin, bn in ((0, 1), (4, 5), (8, 9), (12, 13)):
        h = gelu(batch_norm_1d(sd, f"{p}.{bn}", linear(sd, f"{p}.{lin}", h), training))
    return linear(sd, p + ".16", h)


def weighted_fusion_logits(sd, anchor, raw2, raw3, enh2, enh3, training, prefix=""):
    """MultimodalTransformerModel.forward :298-313: weights from the three raw feature vectors, scale
    [anchor, enhanced2, enhanced3] per sample, concatenate, fusion MLP, arousal head."""
    w = dynamic_weights(sd, prefix + "attention_weights", anchor, raw2, raw3)
    fused = torch.cat([anchor * w[:, 0:1], enh2 * w[:, 1:2], enh3 * w[:, 2:3]], dim=1)
    fused = fusion_mlp(sd, prefix + "fusion", fused, training)
    return arousal_head(sd, prefix + "arousal_head", fused, training), fused


# ------------------------------------------------------------------------------------------------ A5
def classifier(sd, p, x):
    """Classifier.forward, MultimodalModel.py:447-451: shared Linear(256,128)+ReLU(+Dropout) then two Linear(128,3)."""
    h = torch.relu(linear(sd, p + ".shared.0", x))
    return linear(sd, p + ".fc_arousal", h), linear(sd, p + ".fc_valence", h)


def projection_head(sd, p, x, training):
    """ProjectionHead.forward, MultimodalModel.py:416-429: Linear ReLU BN (Dropout) Linear ReLU BN (Dropout) Linear."""
    h = batch_norm_1d(sd, p + ".net.2", torch.relu(linear(sd, p + ".net.0", x)), training)
    h = batch_norm_1d(sd, p + ".net.6", torch.relu(linear(sd, p + ".net.4", h)), training)
    return linear(sd, p + ".net.8", h)


# ------------------------------------------------------------------------------------------------ A6
def cross_entropy(logits, labels):
    """nn.CrossEntropyLoss() (mean reduction, int64 targets): Trainer.py:17,68; Tester.py:20,57."""
    z = logits - logits.max(dim=1, keepdim=True)[0]
    logp = z - torch.log(torch.exp(z).sum(dim=1, keepdim=True))
    return -(logp.gather(1, labels.view(-1, 1))).mean()


def cross_entropy_grad(logits, labels):
    """d(mean CE)/d(logits) = (softmax - onehot) / B — what the fused CE kernel emits."""
    p = torch.softmax(logits, dim=1)
    p[torch.arange(logits.shape[0]), labels] -= 1.0
    return p / logits.shape[0]


# ------------------------------------------------------------------------------------------------ N1
def supervised_infonce(feat1, feat2, labels, temperature):
    """compute_contrastive_loss, MultimodalModel.py:232-260."""
    f1, f2 = l2_normalize(feat1), l2_normalize(feat2)
    sim = f1 @ f2.t() / temperature
    pos = (labels.view(-1, 1) == labels.view(1, -1)).float()
    pos = pos - torch.diag(torch.diag(pos))
    sim = sim - sim.max(dim=1, keepdim=True)[0]
    e = torch.exp(sim)
    return (-torch.log(((e * pos).sum(1) + 1e-12) / (e.sum(1) + 1e-12))).mean()


def supcon_two_view(z1, z2, labels, temperature=0.1):
    """contrastive_loss, train.py:16-40: two views stacked to [2B, D], S = z z^T / T, the diagonal excluded from the
    positives and from the soft-max denominator (no max shift; +1e-8 inside the log and in the positive count)."""
    z = torch.cat([l2_normalize(z1), l2_normalize(z2)], dim=0)
    sim = z @ z.t() / temperature
    lab = torch.cat([labels.view(-1), labels.view(-1)], dim=0)
    eye = torch.eye(z.shape[0], dtype=torch.bool)
    mask = (lab.view(-1, 1) == lab.view(1, -1)).float().masked_fill(eye, 0.0)
    denom = torch.exp(sim).masked_fill(eye, 0.0).sum(dim=1, keepdim=True)
    log_prob = sim - torch.log(denom + 1e-8)
    return (-(mask * log_prob).sum(dim=1) / (mask.sum(dim=1) + 1e-8)).mean()


def nt_xent(z1, z2, temperature):
    """NT-Xent of the reference's ME-MHACL script (MML_ZYC/ME-MHACL/train.py:47-66): z = normalize(cat(z1, z2)); sim = z z^T with
    the diagonal filled with -9e15, divided by T; loss = CE(sim, partner index). (That file is a script — it cannot be imported
    without running its data loading; the golden tests/golden/n1_nt_xent.npz is made by calling its `contrastive_loss` alone,
    lifted out of the parsed file with ast at generation time — tests/test_oracle_golden.py pins this restatement on it.)"""
    import torch.nn.functional as F
    B = z1.shape[0]
    z = F.normalize(torch.cat([z1, z2], 0), dim=1)
    sim = z @ z.t()
    sim = sim.masked_fill(torch.eye(2 * B, dtype=torch.bool), -9e15) / temperature
    targets = torch.cat([torch.arange(B, 2 * B), torch.arange(0, B)])
    return F.cross_entropy(sim, targets)
