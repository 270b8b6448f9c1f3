"""Generates tests/golden/c0_full_size.npz: the END-TO-END C0 golden of SURVEY.md §8(c) — BASELINE.json configs[0]:
BERT-base + ResNet-50 + the reference's fusion head on 16 synthetic 224x224 + 128-token pairs, 3-class CE — computed by
the CPU oracle in fp32 (run in the build container; ~1 min of CPU):

    python tests/golden/make_c0_golden.py

Weights are NOT stored (133 M parameters): both sides regenerate them from `torch.manual_seed(1234)` with the product's
own CPU initialisers (`MultimodalTransformerModel(dropout=0.0)`), the batch from `util.synth_batch(seed=1234)`. Stored:
logits [16,3], loss, the two encoder features [16,256], the global gradient norm, and for EVERY parameter tensor its
gradient's L2 norm plus a strided sample of <= 1024 of its elements (sample stride recorded) — plain arrays only.
The fusion-head part of this graph is the arithmetic pinned on the reference's own modules (make_golden.py); the two
encoders are not in the reference (oracle/__init__.py: BERT pinned on transformers, ResNet "parity unpinned").
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import multimodal_sentiment_aanalysis_amd as mm  # noqa: E402  (CPU: parameter tables + initialisers only)
from oracle import fusion as OF  # noqa: E402
from oracle import model as OM  # noqa: E402
from oracle.bert import BERT_BASE  # noqa: E402
from oracle.policy import FP32  # noqa: E402
from oracle.resnet import RESNET50  # noqa: E402
from util import synth_batch  # noqa: E402

SEED, B, S, HW = 1234, 16, 128, 224
MAX_SAMPLE = 1024


def sample(g):
    flat = g.reshape(-1)
    stride = max(1, -(-flat.numel() // MAX_SAMPLE))
    return flat[::stride].clone(), stride


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    torch.manual_seed(SEED)
    model = mm.MultimodalTransformerModel(dropout=0.0)
    sd = {k: v.detach().clone().contiguous() for k, v in model.state_dict().items()}
    names = [n for n, _ in model.named_parameters() if n not in ("contrastive_weight", "temperature")]
    image, ids, mask, labels = synth_batch(B, S, HW, HW, 30522, seed=SEED)
    cfg = dict(bert=BERT_BASE, resnet=RESNET50)
    params = {n: sd[n].detach().requires_grad_(True) for n in names}
    work = dict(sd)
    work.update(params)
    t0 = time.time()
    logits, mid = OM.model_forward(work, image, ids, mask, cfg, True, FP32)
    loss = OF.cross_entropy(logits, labels)
    gs = torch.autograd.grad(loss, [params[n] for n in names], allow_unused=True)
    print(f"oracle fwd+bwd: {time.time() - t0:.1f} s, loss {loss.item():.6f}")
    out = {"logits": logits.detach().numpy(), "loss": np.float32(loss.item()), "labels": labels.numpy(),
           "text_feat": mid["text"].detach().numpy(), "image_feat": mid["image"].detach().numpy(),
           "seed": np.int64(SEED)}
    total = 0.0
    gnames, gnorms, gstrides = [], [], []
    for n, g in zip(names, gs):
        if g is None:
            continue
        total += float((g.double() ** 2).sum())
        smp, stride = sample(g)
        gnames.append(n)
        gnorms.append(float(g.double().norm()))
        gstrides.append(stride)
        out["gs." + n] = smp.numpy()
    out["grad_names"] = np.array(gnames)
    out["grad_norms"] = np.array(gnorms, dtype=np.float64)
    out["grad_strides"] = np.array(gstrides, dtype=np.int64)
    out["grad_total_norm"] = np.float64(total ** 0.5)
    # post-forward BN running statistics of a few layers (momentum 0.1, unbiased variance)
    for k in ("encoder.image_net.resnet.bn1.running_mean", "encoder.image_net.resnet.bn1.running_var",
              "encoder.image_net.resnet.layer4.2.bn3.running_mean", "encoder.image_net.resnet.layer4.2.bn3.running_var"):
        out["bn." + k] = work[k].numpy()
    path = os.path.join(HERE, "c0_full_size.npz")
    np.savez_compressed(path, **out)
    print(f"c0_full_size.npz: {os.path.getsize(path) / 1e6:.2f} MB, {len(gnames)} gradient tensors, "
          f"|g| = {total ** 0.5:.6f}, logits[0] = {logits[0].tolist()}")


if __name__ == "__main__":
    main()
