"""Drop-in model classes for the text+image hot path (same file / class names as the reference's
MML_ZYC/MultimodalModel.py so `from MultimodalModel import MultiModalEncoder, ProjectionHead, Classifier`
and `MultimodalTransformerModel()` keep working; main.py:40,66, train.py:5,156-158).

Every class is an nn.Module with the reference's constructor signature, `state_dict()` keys and forward
contract, but its arithmetic runs in libmmsa_hip.so (engine.py). The three positional inputs `(x1, x2, x3)`
that the reference loops pass (`eeg, eye, pps` at Trainer.py:53-60) carry `(image, token_ids, attention_mask)`.
"""
import torch
import torch.nn as nn

from .engine import (BERT_BASE, RESNET50, BertTextNet, HeadEngine, ResNetImageNet, materialize)  # noqa: F401
from ._lib import HEAD_CLASSIFIER, HEAD_CROSS_MODAL, HEAD_MM_FUSION, HEAD_PROJECTION, HEAD_WEIGHTED


class CrossModalTransformer(HeadEngine):
    """MultimodalModel.py:108-149: 4-head nn.MultiheadAttention (batch_first) + sigmoid gate + LayerNorm.
    state_dict keys: multihead_attn.{in_proj_weight,in_proj_bias,out_proj.weight,out_proj.bias}, gate.0.*, norm.*"""

    kind = HEAD_CROSS_MODAL

    def __init__(self, embed_dim=256, num_heads=4):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self._tokens = 1
        self._init_head()

    def _base_cfg(self):
        c = super()._base_cfg()
        c.update(embed=self.embed_dim, heads=self.num_heads, tokens=self._tokens)
        return c

    def _out_dims(self):
        return [self.embed_dim]

    def forward(self, query, key, value):
        # 2-D inputs are length-1 sequences (MultimodalModel.py:132-137); the query must be a single token (:144)
        if query.dim() == 3:
            if query.shape[1] != 1:
                raise ValueError("CrossModalTransformer: the reference squeezes the query to one token (:144)")
            query = query[:, 0]
        key = key.unsqueeze(1) if key.dim() == 2 else key
        value = value.unsqueeze(1) if value.dim() == 2 else value
        self._tokens = key.shape[1]
        return self._run(query, key, value)[0]


class MultiModalEncoder(HeadEngine):
    """MultimodalModel.py:357-406 with (image, text) modalities: two encoder slots -> L2-normalise -> 8-head
    self-attention over the modality tokens -> max-pool -> Linear/ReLU/BatchNorm.
    state_dict keys: image_net.*, text_net.*, multihead_attn.*, fusion_mlp.{0,2}.* (reference: eeg_net/eye_net/pps_net)."""

    kind = HEAD_MM_FUSION

    def __init__(self, bert_config=None, resnet_config=None, pool="max"):
        super().__init__()
        self.feat_dim, self.num_heads = 256, 8
        self.pool = pool  # "max": MultimodalModel.py:401; "mean": ME-MHACL/model.py:73
        self.image_net = ResNetImageNet(resnet_config, self.feat_dim)
        self.text_net = BertTextNet(bert_config, self.feat_dim)
        self._init_head()

    def _base_cfg(self):
        c = super()._base_cfg()
        c.update(embed=self.feat_dim, heads=self.num_heads, tokens=2, pool_mode=0 if self.pool == "max" else 1)
        return c

    def _out_dims(self):
        return [self.feat_dim]

    def features(self, image, token_ids, attention_mask=None):
        i = self.image_net(image)  # may run on its own stream (EngineModule.use_side_stream): joined below
        t = self.text_net(token_ids, attention_mask)
        self.image_net.join()
        return i, t

    def fuse(self, image_feat, text_feat):
        return self._run(text_feat, image_feat)[0]

    def forward(self, image, token_ids, attention_mask=None, labels=None):
        self._prepare(image.device)  # one flat buffer for this module and both encoders
        i, t = self.features(image, token_ids, attention_mask)
        return self.fuse(i, t)


class ProjectionHead(HeadEngine):
    """MultimodalModel.py:409-429 (SimCLR-style MLP). state_dict keys net.{0,2,4,6,8}.*"""

    kind = HEAD_PROJECTION

    def __init__(self, in_dim=256, hidden_dim=256, out_dim=128):
        super().__init__()
        self.in_dim, self.hidden_dim, self.out_dim = in_dim, hidden_dim, out_dim
        self.dropout_p = 0.5
        self._init_head()

    def _base_cfg(self):
        c = super()._base_cfg()
        c.update(embed=self.in_dim, hidden=self.hidden_dim, out_dim=self.out_dim, dropout_p=self.dropout_p)
        return c

    def _out_dims(self):
        return [self.out_dim]

    def forward(self, x):
        return self._run(x)[0]


class Classifier(HeadEngine):
    """MultimodalModel.py:432-451: shared Linear+ReLU+Dropout(0.5), two 3-way heads. keys shared.0.*, fc_arousal.*, fc_valence.*"""

    kind = HEAD_CLASSIFIER

    def __init__(self, in_dim=256, hidden_dim=128):
        super().__init__()
        self.in_dim, self.hidden_dim = in_dim, hidden_dim
        self.dropout_p = 0.5
        self._init_head()

    def _base_cfg(self):
        c = super()._base_cfg()
        c.update(embed=self.in_dim, hidden=self.hidden_dim, num_classes=3, dropout_p=self.dropout_p)
        return c

    def _out_dims(self):
        return [3, 3]

    def forward(self, x):
        out_a, out_v = self._run(x)
        return out_a, out_v


class MultimodalTransformerModel(HeadEngine):
    """MultimodalModel.py:152-322 for (image, text): encoder slots + ME-MHACL fusion token (`encoder`), bidirectional
    CrossModalTransformer (:287-297), dynamic weighting + fusion MLP + arousal head (:298-313).

    Forward contracts:
      * Trainer / Tester (older single-head contract, Trainer.py:60, Tester.py:53):
            logits, aux_loss = model(image, token_ids, attention_mask, labels)   /   logits = model(image, token_ids, mask)
      * multitask=True (MultiTaskTrainer.py:199,369): (arousal, valence[, c1, c2, c3]) with the valence head enabled.
    state_dict keys of the weighted head match the reference: attention_weights.*, fusion.*, arousal_head.*, valence_head.*
    """

    kind = HEAD_WEIGHTED

    def __init__(self, num_classes=3, temperature=0.01, bert_config=None, resnet_config=None, multitask=False,
                 dropout=0.3):
        super().__init__()
        self.num_classes, self.multitask, self.dropout_p = num_classes, multitask, dropout
        self.encoder = MultiModalEncoder(bert_config, resnet_config)
        self.cross_attn_t2i = CrossModalTransformer()
        self.cross_attn_i2t = CrossModalTransformer()
        self._init_head()
        # learnable scalars of the reference (MultimodalModel.py:228,230), used by compute_contrastive_loss (N1)
        self.contrastive_weight = nn.Parameter(torch.ones(1))
        self.temperature = nn.Parameter(torch.tensor(float(temperature)))

    def _base_cfg(self):
        c = super()._base_cfg()
        c.update(embed=256, num_classes=self.num_classes, valence=int(self.multitask), dropout_p=self.dropout_p)
        return c

    def _out_dims(self):
        return [self.num_classes, 128] + ([self.num_classes] if self.multitask else [])

    # the reference's attribute names (MultiTaskTrainer.py:59,79,99,120-127 reach into the model by name): slot 1 is the
    # ME-MHACL fusion token (the whole encoder), slot 2 the text encoder, slot 3 the image encoder
    @property
    def eeg_net(self):
        return self.encoder

    @property
    def eye_net(self):
        return self.encoder.text_net

    @property
    def pps_net(self):
        return self.encoder.image_net

    @property
    def cross_attn_e2p(self):
        return self.cross_attn_t2i

    @property
    def cross_attn_p2e(self):
        return self.cross_attn_i2t

    def compute_contrastive_loss(self, feat1, feat2, labels):
        """MultimodalModel.py:232-260 (supervised InfoNCE, learnable temperature) as one fused HIP launch (N1)."""
        from .engine import supervised_infonce
        return supervised_infonce(feat1, feat2, labels, self.temperature)

    def forward(self, image, token_ids, attention_mask=None, labels=None):
        self._prepare(image.device)  # one flat buffer for the whole model
        i, t = self.encoder.features(image, token_ids, attention_mask)
        # (the ME-MHACL fusion token and the two cross-modal transformers are independent chains; running the transformers on side
        #  streams beside the fusion chain was measured in rounds 3-4: 16.80 against 16.85 ms per step — no gain, removed)
        mm = self.encoder.fuse(i, t)
        i_enh = self.cross_attn_t2i(query=t, key=i, value=i)
        t_enh = self.cross_attn_i2t(query=i, key=t, value=t)
        outs = self._run(mm, t, i, i_enh, t_enh)
        logits = outs[0]
        if self.multitask:
            if labels is None:
                return logits, outs[2]
            # MultimodalModel.py:270-284,315-317: one InfoNCE term per modality feature against the arousal labels, each
            # scaled by the learnable contrastive_weight (the three slots are the fusion token, text and image features)
            arousal = labels[0] if isinstance(labels, (tuple, list)) else labels
            c = [self.contrastive_weight * self.compute_contrastive_loss(f, f, arousal) for f in (mm, t, i)]
            return logits, outs[2], c[0], c[1], c[2]
        if labels is None:
            return logits
        # aux loss of the Trainer contract (Trainer.py:60,86: `.item()` is called on it): no contrastive term here, a constant zero
        # kept on the device instead of a torch.zeros launch per step
        z = getattr(self, "_zero_aux", None)
        if z is None or z.device != logits.device:
            z = self._zero_aux = torch.zeros(1, device=logits.device)
        return logits, z
