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
  * `fp8="token"` (BF16G_FP8, BASELINE.json configs[4]): the text encoder's four forward Linears per layer take e4m3 operands —
    the activation is scaled per TOKEN (row) by 448 / row amax, the weight per TENSOR by 448 / amax (amax over the bf16-stored
    values: "current" scaling), both rounded to OCP e4m3 (torch.float8_e4m3fn = round-to-nearest-even, what v_cvt_pk_fp8_f32
    does), multiplied exactly and rescaled: csrc/gemm_fp8.hip (fp8_quant_rows_kernel, fp8_quant_batch_kernel) restated. The
    backward is the bf16 one (straight-through: gradients use the UNQUANTIZED bf16 operands, as the device's backward does).
    `fp8="tensor"`: both operands per tensor (the stand-alone mmsa_fp8_quantize + mmsa_gemm_fp8 pair); `fp8="row"`: one scale
    per activation row and per weight row (output feature).
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


FP8_MAX = 448.0


def e4m3_quantize(x, per_row):  # per_row: one scale per row of the last dimension (a token / an output feature)
    """(q, scale): q = e4m3(x * 448 / amax) as fp32 values, scale = amax / 448 (amax over the tensor, or over each row);
    the expressions of csrc/gemm_fp8.hip::fp8_quant_kernel (inv = 448 / amax in fp32, clamp, RNE conversion)."""
    a = x.abs().amax(dim=-1, keepdim=True) if per_row else x.abs().amax()
    a = a.clamp_min(1e-20).to(torch.float32)
    inv = torch.tensor(FP8_MAX, dtype=torch.float32) / a
    f = (x.to(torch.float32) * inv).clamp(-FP8_MAX, FP8_MAX)
    return f.to(torch.float8_e4m3fn).to(torch.float32), a * torch.tensor(1.0 / FP8_MAX, dtype=torch.float32)


class _Fp8Linear(torch.autograd.Function):
    """forward: (q_x q_w^T) * scale_x * scale_w; backward: straight-through with the unquantized operands."""

    @staticmethod
    def forward(ctx, x, w, mode):
        qx, sx = e4m3_quantize(x, mode in ("row", "token"))
        qw, sw = e4m3_quantize(w, mode == "row")
        ctx.save_for_backward(x, w)
        y = qx @ qw.t()
        return y * sx * (sw.t() if mode == "row" else sw)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return g @ w, g.t() @ x, None


class Policy:
    def __init__(self, storage="fp32", round_grads=False, forced=None, trace=None, fp8=None):
        assert storage in ("fp32", "bf16") and fp8 in (None, "tensor", "row", "token")
        assert fp8 is None or storage == "bf16"
        self.storage, self.round_grads, self.forced, self.trace, self.fp8 = storage, round_grads, forced, trace, fp8
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

    def linear(self, x, w):
        """x [M,K] (a stored activation) times w [N,K]^T (a working-copy weight, already through `qw`) for the Linears that the
        fp8 configuration quantizes; plain product otherwise."""
        if self.fp8 is None:
            return x @ w.t()
        return _Fp8Linear.apply(x, w, self.fp8)

    def qw(self, w):
        """The working copy of a WEIGHT in the storage type: forward rounding only (weight gradients are fp32)."""
        return w if self.storage == "fp32" else _RoundBF16.apply(w)


FP32 = Policy("fp32")
BF16 = Policy("bf16")
BF16G = Policy("bf16", round_grads=True)
BF16G_FP8 = Policy("bf16", round_grads=True, fp8="token")  # what mmsa_bert_fwd does in fp8 mode: per-token x, per-tensor W
