"""BERT encoder restated with elementary tensor ops (TEST INFRASTRUCTURE ONLY).

NOT IN THE REFERENCE: /root/reference contains no BERT. This restates the public BERT-base
architecture as implemented by the third-party package transformers==5.15.0
(`transformers/models/bert/modeling_bert.py`: BertEmbeddings, BertSelfAttention, BertSelfOutput,
BertIntermediate, BertOutput, BertPooler) — post-LN encoder, exact-erf GELU, LayerNorm eps 1e-12, additive
attention mask, absolute position embeddings, token_type_ids = 0, tanh pooler on the first token. It fills the
reference's encoder slot (MultimodalModel.py:264-266). Pinned against that package by tests/golden/make_golden.py.
Dropout probabilities are 0 (the product runs the encoders without dropout).
"""
import math

import torch

from .fusion import gelu
from .policy import FP32


def _ln(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def bert_forward(sd, p, input_ids, attention_mask, cfg, pol=FP32):
    """Returns (last_hidden [B,S,H], pooled [B,H]). `p` is the name prefix of the BertModel parameters (HF names).
    `pol.q` marks where the bf16 HIP path stores a rounded tensor."""
    q, qw, lin = pol.q, pol.qw, pol.linear  # lin: the four Linears per layer that configs[4] runs on fp8 operands
    B, S = input_ids.shape
    H, A, eps = cfg["hidden"], cfg["heads"], cfg["ln_eps"]
    hd = H // A
    W = lambda n: qw(sd[p + n])  # working copy of a matrix parameter in the storage type (its gradient stays fp32)
    emb = (W("embeddings.word_embeddings.weight")[input_ids]
           + W("embeddings.position_embeddings.weight")[torch.arange(S)].unsqueeze(0)
           + W("embeddings.token_type_embeddings.weight")[0].view(1, 1, H))
    x = q(_ln(q(emb), sd[p + "embeddings.LayerNorm.weight"], sd[p + "embeddings.LayerNorm.bias"], eps))
    x = x.reshape(B * S, H)
    keep = None if attention_mask is None else (attention_mask != 0)
    for l in range(cfg["layers"]):
        lp = f"encoder.layer.{l}."
        Wqkv = torch.cat([W(lp + "attention.self.query.weight"), W(lp + "attention.self.key.weight"),
                          W(lp + "attention.self.value.weight")], 0)
        bqkv = torch.cat([sd[p + lp + "attention.self.query.bias"], sd[p + lp + "attention.self.key.bias"],
                          sd[p + lp + "attention.self.value.bias"]], 0)
        qkv = q(lin(x, Wqkv) + bqkv).view(B, S, 3, A, hd)
        qh, kh, vh = qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2)
        scores = qh @ kh.transpose(-1, -2) * (1.0 / math.sqrt(hd))
        if keep is not None:
            scores = scores.masked_fill(~keep[:, None, None, :], -3.0e38)
        probs = q(torch.softmax(scores, dim=-1))
        ctx = q((probs @ vh).transpose(1, 2).reshape(B * S, H))
        s1 = q(lin(ctx, W(lp + "attention.output.dense.weight")) + sd[p + lp + "attention.output.dense.bias"] + x)
        h1 = q(_ln(s1, sd[p + lp + "attention.output.LayerNorm.weight"], sd[p + lp + "attention.output.LayerNorm.bias"], eps))
        pre = lin(h1, W(lp + "intermediate.dense.weight")) + sd[p + lp + "intermediate.dense.bias"]
        act = q(gelu(pre))
        s2 = q(lin(act, W(lp + "output.dense.weight")) + sd[p + lp + "output.dense.bias"] + h1)
        x = q(_ln(s2, sd[p + lp + "output.LayerNorm.weight"], sd[p + lp + "output.LayerNorm.bias"], eps))
    hidden = x.view(B, S, H)
    pooled = q(torch.tanh(hidden[:, 0] @ W("pooler.dense.weight").t() + sd[p + "pooler.dense.bias"]))
    return hidden, pooled


BERT_BASE = dict(hidden=768, layers=12, heads=12, intermediate=3072, vocab=30522, max_pos=512, type_vocab=2,
                 ln_eps=1e-12)
BERT_LARGE = dict(hidden=1024, layers=24, heads=16, intermediate=4096, vocab=30522, max_pos=512, type_vocab=2,
                  ln_eps=1e-12)


def bert_param_shapes(cfg):
    """(name, shape) in HF BertModel order (no prefix)."""
    H, I = cfg["hidden"], cfg["intermediate"]
    out = [("embeddings.word_embeddings.weight", (cfg["vocab"], H)),
           ("embeddings.position_embeddings.weight", (cfg["max_pos"], H)),
           ("embeddings.token_type_embeddings.weight", (cfg["type_vocab"], H)),
           ("embeddings.LayerNorm.weight", (H,)), ("embeddings.LayerNorm.bias", (H,))]
    for l in range(cfg["layers"]):
        lp = f"encoder.layer.{l}."
        out += [(lp + "attention.self.query.weight", (H, H)), (lp + "attention.self.query.bias", (H,)),
                (lp + "attention.self.key.weight", (H, H)), (lp + "attention.self.key.bias", (H,)),
                (lp + "attention.self.value.weight", (H, H)), (lp + "attention.self.value.bias", (H,)),
                (lp + "attention.output.dense.weight", (H, H)), (lp + "attention.output.dense.bias", (H,)),
                (lp + "attention.output.LayerNorm.weight", (H,)), (lp + "attention.output.LayerNorm.bias", (H,)),
                (lp + "intermediate.dense.weight", (I, H)), (lp + "intermediate.dense.bias", (I,)),
                (lp + "output.dense.weight", (H, I)), (lp + "output.dense.bias", (H,)),
                (lp + "output.LayerNorm.weight", (H,)), (lp + "output.LayerNorm.bias", (H,))]
    out += [("pooler.dense.weight", (H, H)), ("pooler.dense.bias", (H,))]
    return out
