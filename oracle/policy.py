"""Precision policy of the oracle (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).

`Policy("fp32")` is the reference arithmetic (the reference runs fp32 on CPU: Trainer.py:53-54 `.float()`).
`Policy("bf16")` rounds to bfloat16 (round-to-nearest-even, straight-through gradient) at exactly the points
where the HIP path stores a bf16 tensor, so the two compute the same function up to accumulation order.
All arithmetic stays fp32 in both modes.
"""
import torch


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class Policy:
    def __init__(self, storage="fp32"):
        assert storage in ("fp32", "bf16")
        self.storage = storage

    def q(self, x):
        """Round to the activation/weight storage type (identity for fp32)."""
        if self.storage == "fp32":
            return x
        return _RoundBF16.apply(x)


FP32 = Policy("fp32")
BF16 = Policy("bf16")
