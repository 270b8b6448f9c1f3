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
from oracle.policy import BF16G, FP32  # noqa: E402
from oracle.resnet import RESNET50  # noqa: E402
from util import synth_batch  # noqa: E402

SEED, B, S, HW = 1234, 16, 128, 224
MAX_SAMPLE = 1024


def sample(g):
    flat = g.reshape(-1)
    stride = max(1, -(-flat.numel() // MAX_SAMPLE))
    return flat[::stride].clone(), stride


def damp_residual_branches(sd_or_model, gamma3):
    """Scale the last BatchNorm weight of every bottleneck (bn3) by gamma3 (in place). torchvision offers the extreme form as
    `zero_init_residual`; trained ResNets sit in this damped regime. At the default init (gamma3 = 1) a random ResNet-50 with
    batch-statistics BatchNorm AMPLIFIES any perturbation ~1.37x per bottleneck (x160 over the net: mean-field "gradient
    explosion" of BN nets), so bf16 storage rounding (2^-9) reaches ~50 % of the last block's activations in ANY
    implementation — the oracle's own bf16 policy included. Both regimes get a golden."""
    items = sd_or_model.items() if isinstance(sd_or_model, dict) else sd_or_model.state_dict().items()
    with torch.no_grad():
        for k, v in items:
            if k.endswith("bn3.weight"):
                v.mul_(gamma3)


def main(gamma3=1.0, fname="c0_full_size.npz"):
    torch.set_num_threads(os.cpu_count() or 8)
    torch.manual_seed(SEED)
    model = mm.MultimodalTransformerModel(dropout=0.0)
    damp_residual_branches(model, gamma3)
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
    # the same forward under the bf16 storage policy: how far ANY bf16-storage implementation of this graph is from fp32
    with torch.no_grad():
        lb, mb = OM.model_forward({k: v.clone() for k, v in sd.items()}, image, ids, mask, cfg, True, BF16G)
    bf = {"logits_bf16_policy": lb.numpy(), "loss_bf16_policy": np.float32(OF.cross_entropy(lb, labels).item()),
          "text_feat_bf16_policy": mb["text"].numpy(), "image_feat_bf16_policy": mb["image"].numpy()}
    print("bf16-policy oracle vs fp32 oracle: |dlogits| %.3e, |dloss| %.3e, text feat %.3e, image feat %.3e" % (
        (lb - logits.detach()).abs().max().item(), abs(float(bf["loss_bf16_policy"]) - loss.item()),
        ((mb["text"] - mid["text"].detach()).abs().max() / mid["text"].detach().abs().max()).item(),
        ((mb["image"] - mid["image"].detach()).abs().max() / mid["image"].detach().abs().max()).item()))
    out = {"gamma3": np.float32(gamma3), **bf, "logits": logits.detach().numpy(), "loss": np.float32(loss.item()), "labels": labels.numpy(),
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
    path = os.path.join(HERE, fname)
    np.savez_compressed(path, **out)
    print(f"{fname}: {os.path.getsize(path) / 1e6:.2f} MB, {len(gnames)} gradient tensors, "
          f"|g| = {total ** 0.5:.6f}, logits[0] = {logits[0].tolist()}")


if __name__ == "__main__":
    main(1.0, "c0_full_size.npz")
    main(0.25, "c0_damped.npz")
