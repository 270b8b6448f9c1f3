"""GPU parity tests of the streaming 1x1-convolution GEMM (csrc/gemm_stream.hip), through the C ABI (mmsa_gemm).

The kernel takes the HBM-bound launches of the image encoder's first two stages (K = 64 .. 256, N = 64 .. 512, plain store or
+ residual): every eligible (N, K) pair in both operand layouts, with and without the side operand, at row counts that give a
wave zero, one, two and many tiles (the two-register-set stream has three exits), against a float32 matmul of the same
bf16-rounded operands AND against the persistent kernel (MMSA_DISABLE=stream1x1), which accumulates K in the same order with the
same instruction: the two must agree BIT FOR BIT. (The statistics epilogue is pinned through the ResNet engine:
test_conv_epilogue_statistics_match_the_statistics_pass and the teacher-forced test run on it.)
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from multimodal_sentiment_aanalysis_amd import kernels as K
from multimodal_sentiment_aanalysis_amd._lib import GEMM_BF16_MFMA

BF = torch.bfloat16


def rnd(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(BF).to(dev)


class disabled:
    def __init__(self, what):
        self.what = what

    def __enter__(self):
        self.old = os.environ.get("MMSA_DISABLE")
        os.environ["MMSA_DISABLE"] = self.what

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("MMSA_DISABLE", None)
        else:
            os.environ["MMSA_DISABLE"] = self.old


NK = [(64, 64), (64, 128), (64, 256), (128, 64), (128, 128), (128, 256), (256, 64), (256, 128), (512, 64), (512, 128)]


@pytest.mark.parametrize("with_add", [False, True])
@pytest.mark.parametrize("b_kmajor", [0, 1])
@pytest.mark.parametrize("N,Kd", NK)
def test_stream_matches_reference_and_persistent_kernel(dev, N, Kd, b_kmajor, with_add):
    for M in (4096, 16 * 2048 + 16 * 5, 50176):  # one tile per row-wave at most / ragged second round / a long stream
        A = rnd((M, Kd), dev, 1)
        B = rnd((Kd, N) if b_kmajor else (N, Kd), dev, 2, scale=0.25)
        add = rnd((M, N), dev, 3) if with_add else None
        ref = A.float() @ (B.float() if b_kmajor else B.float().T)
        if with_add:
            ref = ref + add.float()
        outs = []
        for off in (False, True):
            C = torch.full((M, N), float("nan"), dtype=BF, device=dev)
            kw = dict(b_kmajor=b_kmajor, add=add, ldadd=N, impl=GEMM_BF16_MFMA)
            if off:
                with disabled("stream1x1"):
                    K.gemm(A, B, C, M, N, Kd, Kd, N if b_kmajor else Kd, N, **kw)
            else:
                K.gemm(A, B, C, M, N, Kd, Kd, N if b_kmajor else Kd, N, **kw)
            outs.append(C)
        err = (outs[0].float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        assert err < 1.2e-2, f"M={M} N={N} K={Kd} km={b_kmajor} add={with_add}: rel-to-max err {err:.3e}"
        assert torch.equal(outs[0], outs[1]), f"M={M} N={N} K={Kd} km={b_kmajor} add={with_add}: differs from the persistent kernel"


def test_stream_respects_leading_dimensions(dev):
    """Views: A, C and the side operand as column slices of wider buffers (the engines pass row strides, not shapes)."""
    M, N, Kd = 8192, 256, 64
    Abig, Cbig, Sbig = rnd((M, Kd + 64), dev, 4), torch.zeros((M, N + 128), dtype=BF, device=dev), rnd((M, N + 32), dev, 5)
    B = rnd((N, Kd), dev, 6, scale=0.25)
    A, C, S = Abig[:, 32:32 + Kd], Cbig[:, 64:64 + N], Sbig[:, 16:16 + N]
    K.gemm(A, B, C, M, N, Kd, Kd + 64, Kd, N + 128, add=S, ldadd=N + 32, impl=GEMM_BF16_MFMA)
    ref = A.float() @ B.float().T + S.float()
    err = (C.float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1.2e-2, err
    assert Cbig[:, :64].abs().max().item() == 0 and Cbig[:, 64 + N:].abs().max().item() == 0, "stores stay inside the view"
