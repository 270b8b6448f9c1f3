"""Generates tests/golden/c4_fp8_b128.npz: BASELINE.json configs[4] at its OWN workload — BERT-base + ResNet-50 + fusion head,
B = 128, S = 128, 224x224, training-mode forward — under three precision policies of the CPU oracle (run in the build container;
a few minutes of CPU):

    python tests/golden/make_c4_golden.py

  fp32        the reference arithmetic;
  bf16        bf16 storage at the points where the HIP path stores bf16 (oracle/policy.py BF16G);
  fp8         bf16 storage + e4m3 operands (per-token activation scales, per-tensor weight scales, current scaling) in the text encoder's four forward Linears per layer
              (BF16G_FP8: csrc/gemm_fp8.hip restated) — what precision="fp8" computes.

Weights are regenerated from seed 1234 by the product's initialisers on both sides (as make_c0_golden.py), the batch from
util.synth_batch(seed=1234). Stored: loss, logits, text / image features per policy — plain arrays. The GPU test
(tests/test_golden_gpu.py::test_c4_fp8_b128_workload) bounds the device's fp8 step by the ORACLE's own fp8-vs-fp32 distance."""
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
from oracle.policy import BF16G, BF16G_FP8, FP32  # noqa: E402
from oracle.resnet import RESNET50  # noqa: E402
from util import synth_batch  # noqa: E402

SEED, B, S, HW = 1234, 128, 128, 224


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    torch.manual_seed(SEED)
    model = mm.MultimodalTransformerModel(dropout=0.0)
    sd = {k: v.detach().clone().contiguous() for k, v in model.state_dict().items()}
    image, ids, mask, labels = synth_batch(B, S, HW, HW, 30522, seed=SEED)
    cfg = dict(bert=BERT_BASE, resnet=RESNET50)
    out = {"seed": np.int64(SEED), "labels": labels.numpy()}
    for tag, pol in (("fp32", FP32), ("bf16", BF16G), ("fp8", BF16G_FP8)):
        t0 = time.time()
        with torch.no_grad():
            logits, mid = OM.model_forward({k: v.clone() for k, v in sd.items()}, image, ids, mask, cfg, True, pol)
        loss = OF.cross_entropy(logits, labels).item()
        print(f"{tag}: loss {loss:.6f} ({time.time() - t0:.0f} s)")
        out[f"loss_{tag}"] = np.float32(loss)
        out[f"logits_{tag}"] = logits.numpy()
        out[f"text_feat_{tag}"] = mid["text"].numpy()
        out[f"image_feat_{tag}"] = mid["image"].numpy()
    path = os.path.join(HERE, "c4_fp8_b128.npz")
    np.savez_compressed(path, **out)
    t32, t16, t8 = (torch.from_numpy(out[f"text_feat_{k}"]) for k in ("fp32", "bf16", "fp8"))
    print(f"c4_fp8_b128.npz: {os.path.getsize(path) / 1e6:.2f} MB; text feature rel L2: bf16 vs fp32 {(t16 - t32).norm() / t32.norm():.3e}, "
          f"fp8 vs fp32 {(t8 - t32).norm() / t32.norm():.3e}, fp8 vs bf16 {(t8 - t16).norm() / t16.norm():.3e}")


if __name__ == "__main__":
    main()
