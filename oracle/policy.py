"""Precision policy of the oracle (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).

`Policy("fp32")` is the reference arithmetic (the reference runs fp32 on CPU: Trainer.py:53-54 `.float()`).
`Policy("bf16")` rounds to bfloat16 (round-to-nearest-even) at exactly the points where the HIP path stores a bf16
tensor, so the two compute the same function up to accumulation order. All arithmetic stays fp32 in both modes.

Three refinements over round 1:
  * `round_grads=True` (BF16G): the BACKWARD of an activation rounding point rounds the gradient too — the HIP backward
    stores every activation gradient (BatchNorm dz, data gradients, LayerNorm dx, ...) as bf16 at the same tensors.
    Weight gradients are fp32 on both sides: weights go through `qw` (forward rounding of the working copy only).
  * `forced={name: tensor}`: TEACHER FORCING. A named rounding point returns the given tensor (the value the HIP forward
    actually stored, read back from its workspace) instead of rounding its own input, while gradients still flow to
    the input. The oracle's backward is then evaluated at the HIP path's own forward values: with the forward fixed the
    backward is a LINEAR map, so the comparison is not amplified by the chaos of a deep ReLU/BatchNorm net under bf16
    rounding (two correct bf16 forwards differ by 1-ulp flips that decorrelate within a few layers), and a wrong
    backward kernel cannot hide behind a loose tolerance.
  * `trace={}`: collects every named activation (for tests that look at intermediate tensors).
"""
import torch


def _rne(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _rne(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBF16Both(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _rne(x)

    @staticmethod
    def backward(ctx, g):
        return _rne(g)


class _Force(torch.autograd.Function):
    """forward: the forced value; backward: (optionally bf16-rounded) gradient to the computed input."""

    @staticmethod
    def forward(ctx, x, forced, round_grads):
        ctx.round_grads = round_grads
        assert forced.shape == x.shape, (forced.shape, x.shape)
        return forced.to(torch.float32).clone()

    @staticmethod
    def backward(ctx, g):
        return (_rne(g) if ctx.round_grads else g), None, None


class Policy:
    def __init__(self, storage="fp32", round_grads=False, forced=None, trace=None):
        assert storage in ("fp32", "bf16")
        self.storage, self.round_grads, self.forced, self.trace = storage, round_grads, forced, trace
        self.local_err = {}  # forcing: relative L2 distance of each forced tensor from the value computed from its forced inputs

    def q(self, x, name=None):
        """Round an ACTIVATION to the storage type (identity for fp32). `name` identifies the tensor for forcing/tracing."""
        if self.forced is not None and name is not None and name in self.forced:
            with torch.no_grad():  # a per-layer check of the forward: this layer's output given the stored inputs
                f = self.forced[name].to(torch.float32)
                self.local_err[name] = ((x.detach() - f).norm() / f.norm().clamp_min(1e-30)).item()
            x = _Force.apply(x, self.forced[name], self.round_grads and self.storage == "bf16")
        elif self.storage == "bf16":
            x = (_RoundBF16Both if self.round_grads else _RoundBF16).apply(x)
        if self.trace is not None and name is not None:
            self.trace[name] = x
        return x

    def qw(self, w):
        """The working copy of a WEIGHT in the storage type: forward rounding only (weight gradients are fp32)."""
        return w if self.storage == "fp32" else _RoundBF16.apply(w)


FP32 = Policy("fp32")
BF16 = Policy("bf16")
BF16G = Policy("bf16", round_grads=True)
