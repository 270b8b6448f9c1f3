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


@pytest.mark.parametrize("B,H,W", [(32, 56, 56), (8, 40, 24), (16, 33, 17)])
def test_stream_3x3_stage1_matches_conv2d_and_persistent_kernel(dev, B, H, W):
    """The 3 x 3 stream (64 -> 64 channels, stride 1, pad 1: conv2 of the stage-1 bottlenecks, forward and data gradient)
    against F.conv2d + autograd and, bit for bit, against the persistent kernel's implicit GEMM (same tap-major K order). Odd
    image sizes make 16-pixel tiles straddle image rows and images (every lane decomposes its own pixel; padding taps read
    zeros through the buffer descriptor)."""
    import torch.nn.functional as F
    if (B * H * W) % 16 or B * H * W < 4096:
        pytest.skip("not a stream shape")
    Cin = Cout = 64
    x = rnd((B, H, W, Cin), dev, 1)
    w = rnd((Cout, 3, 3, Cin), dev, 2, 0.1)
    dy = rnd((B, H, W, Cout), dev, 3)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.float().permute(0, 3, 1, 2)
    yr = F.conv2d(xr, wr, stride=1, padding=1)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    M, Kd = B * H * W, 9 * Cin
    res = {}
    for tag, off in (("stream", None), ("persistent", "stream3x3")):
        ctx = disabled(off) if off else disabled("")
        with ctx:
            y = torch.full((M, Cout), float("nan"), dtype=BF, device=dev)
            g = K.conv_geom(H, W, H, W, 3, 3, 1, 1, -1, 1, Cin, Cin)
            K.gemm(x, w, y, M, Cout, Kd, Cin, Kd, Cout, gather=1, geom=g, impl=GEMM_BF16_MFMA)
            dx = torch.full((M, Cin), float("nan"), dtype=BF, device=dev)
            g = K.conv_geom(H, W, H, W, 3, 3, 1, -1, 1, 1, Cout, Cout)
            K.gemm(dy, w, dx, M, Cin, 9 * Cout, Cout, Kd, Cin, b_kmajor=1, gather=1, geom=g, b_tap_stride=Cin, impl=GEMM_BF16_MFMA)
        res[tag] = (y, dx)
    y, dx = res["stream"]
    e1 = (y.view(B, H, W, Cout).float() - yr.detach().permute(0, 2, 3, 1)).abs().max().item() / yr.abs().max().item()
    e2 = (dx.view(B, H, W, Cin).float() - xr.grad.permute(0, 2, 3, 1)).abs().max().item() / xr.grad.abs().max().item()
    assert e1 < 1.2e-2 and e2 < 1.2e-2, (e1, e2)
    assert torch.equal(y, res["persistent"][0]), "forward differs from the persistent kernel"
    assert torch.equal(dx, res["persistent"][1]), "data gradient differs from the persistent kernel"
