"""GPU parity tests of the persistent 3-stage MFMA GEMM (csrc/gemm_mfma2.hip), through the C ABI.

The kernel's hazards are ordering hazards (counted vmcnt waits, one barrier per K step, prefetch across tile
boundaries), so the cases are chosen to walk every schedule shape: one K step per tile (K = 64), many tiles per
workgroup (> 256 items), ragged M / N edges, every tile shape (MMSA_G2_NJ = 2 / 3 / 4 for the 256-row tile, 2:1 / 2:2 for the 128-row tile), all four operand layouts,
split-K slabs, plain and memory-reading epilogues, and the BERT-base shapes at full size. The reference is a
float32 matmul of the same bf16-rounded operands (torch, on the device); tolerance = bf16 output rounding.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from multimodal_sentiment_aanalysis_amd import kernels as K
from multimodal_sentiment_aanalysis_amd._lib import ACT_GELU, GEMM_BF16_MFMA

BF = torch.bfloat16


def rnd(shape, dev, seed, scale=1.0, dtype=BF):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


def close(got, ref, what, tol=1.2e-2):
    got = got.float()
    ref = ref.float()
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item() / scale
    assert err < tol, f"{what}: rel-to-max err {err:.3e} >= {tol}"


class force_nj:
    def __init__(self, nj):
        self.nj = nj

    def __enter__(self):
        if self.nj:
            os.environ["MMSA_G2_NJ"] = str(self.nj)

    def __exit__(self, *a):
        os.environ.pop("MMSA_G2_NJ", None)


SHAPES = [  # M, N, K
    (256, 128, 64),      # one tile, one step
    (8192, 64, 64),      # K = 64 stream: one step per tile, 32 tiles
    (100000, 64, 64),    # > 256 items per launch, ragged last row tile
    (50000, 256, 128),   # two steps per tile, many tiles, ragged
    (8192, 768, 768),    # BERT out-proj
    (8192, 2304, 768),   # BERT QKV
    (1000, 200, 192),    # ragged M and N (N % 8 == 0), three steps
    (777, 136, 320),     # ragged, N not a multiple of 16
]


@pytest.mark.parametrize("nj", [0, 2, 3, 4, "2:1", "2:2"])
@pytest.mark.parametrize("M,N,Kd", SHAPES)
def test_g2_nt(dev, M, N, Kd, nj):
    A = rnd((M, Kd), dev, 1)
    B = rnd((N, Kd), dev, 2)
    C = torch.full((M, N), float("nan"), dtype=BF, device=dev)
    with force_nj(nj):
        K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, impl=GEMM_BF16_MFMA)
    close(C, A.float() @ B.float().T, f"NT nj={nj}")


@pytest.mark.parametrize("nj", [0, 2, 3, 4, "2:1", "2:2"])
@pytest.mark.parametrize("M,N,Kd", [(8192, 768, 3072), (50000, 64, 256), (1000, 200, 192), (8192, 3072, 768)])
def test_g2_nn(dev, M, N, Kd, nj):
    A = rnd((M, Kd), dev, 1)
    B = rnd((Kd, N), dev, 2)
    C = torch.full((M, N), float("nan"), dtype=BF, device=dev)
    with force_nj(nj):
        K.gemm(A, B, C, M, N, Kd, Kd, N, N, b_kmajor=1, impl=GEMM_BF16_MFMA)
    close(C, A.float() @ B.float(), f"NN nj={nj}")


@pytest.mark.parametrize("nj", [0, 2, 3, 4, "2:1", "2:2"])
@pytest.mark.parametrize("M,N,Kd,split", [(768, 3072, 8192, 8), (64, 576, 200704, 64), (256, 64, 50176, 32),
                                          (136, 264, 1024, 1), (2304, 768, 8192, 4), (64, 64, 6400, 16)])
def test_g2_tn_f32(dev, M, N, Kd, split, nj):
    A = rnd((Kd, M), dev, 1, 0.5)
    B = rnd((Kd, N), dev, 2, 0.5)
    C0 = rnd((M, N), dev, 3, dtype=torch.float32)
    C = C0.clone()
    with force_nj(nj):
        K.gemm(A, B, C, M, N, Kd, M, N, N, a_kmajor=1, b_kmajor=1, out_f32=1, accumulate=1, split_k=split,
               impl=GEMM_BF16_MFMA)
    ref = C0.double() + A.double().T @ B.double()
    close(C.double(), ref, f"TN nj={nj}", tol=2e-5)
    # overwrite mode, bf16-free fp32 store
    C2 = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    with force_nj(nj):
        K.gemm(A, B, C2, M, N, Kd, M, N, N, a_kmajor=1, b_kmajor=1, out_f32=1, split_k=split, impl=GEMM_BF16_MFMA)
    close(C2.double(), A.double().T @ B.double(), f"TN overwrite nj={nj}", tol=2e-5)


@pytest.mark.parametrize("nj", [0, 2, 3, 4, "2:1", "2:2"])
def test_g2_tn_kcontig_b(dev, nj):
    M, N, Kd = 520, 264, 192
    A = rnd((Kd, M), dev, 1)
    B = rnd((N, Kd), dev, 2)
    C = torch.full((M, N), float("nan"), dtype=BF, device=dev)
    with force_nj(nj):
        K.gemm(A, B, C, M, N, Kd, M, Kd, N, a_kmajor=1, b_kmajor=0, impl=GEMM_BF16_MFMA)
    close(C, A.float().T @ B.float().T, f"TN/kc nj={nj}")


@pytest.mark.parametrize("nj", [0, 3, "2:2"])
@pytest.mark.parametrize("M,N,Kd", [(8192, 3072, 768), (700, 200, 128)])
def test_g2_epilogues(dev, M, N, Kd, nj):
    A = rnd((M, Kd), dev, 1, 0.3)
    B = rnd((N, Kd), dev, 2, 0.3)
    bias = rnd((N,), dev, 3, dtype=torch.float32)
    add = rnd((M, N), dev, 4)
    pre = rnd((M, N), dev, 5)
    acc = A.float() @ B.float().T + bias
    C = torch.full((M, N), float("nan"), dtype=BF, device=dev)
    C2 = torch.full((M, N), float("nan"), dtype=BF, device=dev)
    with force_nj(nj):
        K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias, C2=C2, ldc2=N, act=ACT_GELU, impl=GEMM_BF16_MFMA)
        close(C2, acc, "pre-activation")
        close(C, F.gelu(acc), "gelu")
        K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias, add=add, ldadd=N, impl=GEMM_BF16_MFMA)
        close(C, acc + add.float(), "bias+residual")
        x = pre.float()
        gp = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
        K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, mul=pre, ldmul=N, add=add, ldadd=N, impl=GEMM_BF16_MFMA)
        close(C, (A.float() @ B.float().T) * gp + add.float(), "gelu-grad")
        # a plain-store launch right after memory-reading epilogues, then repeated launches must agree bitwise
        K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, impl=GEMM_BF16_MFMA)
        first = C.clone()
        for _ in range(3):
            K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, impl=GEMM_BF16_MFMA)
            assert torch.equal(C, first), "repeat launches differ (ordering hazard)"
    close(first, A.float() @ B.float().T, "plain")


CONVS = [  # B, H, W, Cin, Cout, k, stride, pad — large enough for many tiles / several items per workgroup
    (32, 56, 56, 64, 64, 3, 1, 1),      # ResNet stage 1 (M = 100352): one tap per K step
    (16, 56, 56, 128, 128, 3, 2, 1),    # strided 3x3 (stage 2 first block)
    (16, 28, 28, 256, 512, 1, 2, 0),    # strided 1x1 downsample
    (8, 14, 14, 256, 256, 3, 1, 1),     # K = 2304
    (4, 7, 7, 512, 512, 3, 1, 1),       # few tiles, deep K -> split-K with a gather
]


@pytest.mark.parametrize("nj", [0, 2, 3, "2:1", "2:2"])
@pytest.mark.parametrize("cfg", CONVS)
def test_g2_conv(dev, cfg, nj):
    B, H, W, Cin, Cout, k, s, p = cfg
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = rnd((B, H, W, Cin), dev, 1)
    w = rnd((Cout, k, k, Cin), dev, 2, 0.1)
    dy = rnd((B, OH, OW, Cout), dev, 3)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.float().permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.conv2d(xr, wr, stride=s, padding=p)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    Kd = k * k * Cin
    with force_nj(nj):
        y = torch.full((B * OH * OW, Cout), float("nan"), dtype=BF, device=dev)
        g = K.conv_geom(H, W, OH, OW, k, k, s, 1, -p, 1, Cin, Cin)
        K.gemm(x, w, y, B * OH * OW, Cout, Kd, Cin, Kd, Cout, gather=1, geom=g, impl=GEMM_BF16_MFMA)
        close(y.view(B, OH, OW, Cout), yr.detach().permute(0, 2, 3, 1), "conv fwd")
        dx = torch.full((B * H * W, Cin), float("nan"), dtype=BF, device=dev)
        g = K.conv_geom(OH, OW, H, W, k, k, 1, -1, p, s, Cout, Cout)
        K.gemm(dy, w, dx, B * H * W, Cin, k * k * Cout, Cout, Kd, Cin, b_kmajor=1, gather=1, geom=g, b_tap_stride=Cin,
               impl=GEMM_BF16_MFMA)
        close(dx.view(B, H, W, Cin), xr.grad.permute(0, 2, 3, 1), "conv dgrad")
        dw = torch.full((Cout, Kd), float("nan"), dtype=torch.float32, device=dev)
        g = K.conv_geom(H, W, OH, OW, k, k, s, 1, -p, 1, Cin, Cin)
        K.gemm(dy, x, dw, Cout, Kd, B * OH * OW, Cout, Cin, Kd, a_kmajor=1, b_kmajor=1, gather=2, geom=g, out_f32=1,
               split_k=16, impl=GEMM_BF16_MFMA)
        close(dw.view(Cout, k, k, Cin), wr.grad.permute(0, 2, 3, 1), "conv wgrad", tol=2e-3)


def test_g2_group(dev):
    """The four weight gradients of a BERT-base layer (K = 8192 rows) as one grouped launch, and a ragged small group."""
    for Kd, shapes in [(8192, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]), (640, [(136, 264), (520, 72), (64, 64)])]:
        jobs, refs = [], []
        for i, (M, N) in enumerate(shapes):
            A = rnd((Kd, M), dev, 10 + i, 0.5)
            B = rnd((Kd, N), dev, 20 + i, 0.5)
            C = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
            jobs.append((A, B, C))
            refs.append(A.double().T @ B.double())
        rc = K.gemm_group(jobs)
        assert rc == 0, rc
        for (A, B, C), ref in zip(jobs, refs):
            close(C.double(), ref, f"group K={Kd} {tuple(C.shape)}", tol=2e-5)


def test_g2_group_split_plain(dev):
    """The 1x1 weight gradients of a ResNet-50 stage as ONE launch with a common K split (mmsa_gemm_group_split): stage 3 at
    B = 64 (six 1024x256 and five 256x1024 outputs, K = 12544 pixels), a ragged small group, and the accumulating form."""
    for Kd, shapes, acc in [(12544, [(1024, 256)] * 6 + [(256, 1024)] * 5, 0), (640, [(136, 264), (520, 72), (64, 64)], 0),
                            (3136, [(2048, 512), (512, 2048), (2048, 512)], 1)]:
        jobs, refs = [], []
        for i, (M, N) in enumerate(shapes):
            A = rnd((Kd, M), dev, 10 + i, 0.5)
            B = rnd((Kd, N), dev, 40 + i, 0.5)
            if acc:
                C = rnd((M, N), dev, 70 + i, 3.0).float()
                refs.append(C.double() + A.double().T @ B.double())
            else:
                C = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
                refs.append(A.double().T @ B.double())
            jobs.append((A, B, C))
        rc = K.gemm_group(jobs, ws_bytes=96 << 20, accumulate=acc)
        assert rc == 0, rc
        for (A, B, C), ref in zip(jobs, refs):
            close(C.double(), ref, f"group split K={Kd} {tuple(C.shape)}", tol=2e-5)
        if not acc:  # bitwise reproducible: the slabs are summed in slab order
            again = [(A, B, torch.empty_like(C)) for A, B, C in jobs]
            assert K.gemm_group(again, ws_bytes=96 << 20) == 0
            for (_, _, C0), (_, _, C1) in zip(jobs, again):
                assert torch.equal(C0, C1)


@pytest.mark.parametrize("cfg", [(16, 14, 14, 64, 64, 3, 1, 1, 5), (4, 28, 28, 128, 128, 3, 1, 1, 3), (64, 7, 7, 64, 128, 3, 1, 1, 2)])
def test_g2_group_split_conv(dev, cfg):
    """Same-geometry 3x3 weight gradients (implicit GEMM over the activation, gather 2) as one grouped launch with a K split."""
    B, H, W, Cin, Cout, k, s, p, n = cfg
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    Kd = k * k * Cin
    g = K.conv_geom(H, W, OH, OW, k, k, s, 1, -p, 1, Cin, Cin)
    jobs, refs = [], []
    for i in range(n):
        x = rnd((B, H, W, Cin), dev, 100 + i)
        dy = rnd((B, OH, OW, Cout), dev, 200 + i)
        wr = torch.zeros((Cout, Cin, k, k), device=dev, requires_grad=True)
        F.conv2d(x.float().permute(0, 3, 1, 2), wr, stride=s, padding=p).backward(dy.float().permute(0, 3, 1, 2))
        refs.append(wr.grad.permute(0, 2, 3, 1).reshape(Cout, Kd))
        jobs.append((dy.view(B * OH * OW, Cout), x, torch.full((Cout, Kd), float("nan"), dtype=torch.float32, device=dev)))
    rc = K.gemm_group(jobs, ws_bytes=64 << 20, geom=g)
    assert rc == 0, rc
    for (_, _, C), ref in zip(jobs, refs):
        close(C, ref, f"grouped conv wgrad {cfg}", tol=2e-3)


def test_g2_group_with_bias_problems(dev):
    """A BERT-base layer's group as the engine launches it: four weight gradients plus two bias gradients expressed as
    dY^T x ones[K][8] (every column of the [N][8] result is the column sum of dY), six problems in one launch."""
    Kd = 8192
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    jobs, refs = [], []
    dys = []
    for i, (M, N) in enumerate(shapes):
        A = rnd((Kd, M), dev, 10 + i, 0.5)
        B = rnd((Kd, N), dev, 20 + i, 0.5)
        C = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
        jobs.append((A, B, C))
        refs.append(A.double().T @ B.double())
        dys.append(A)
    ones = torch.ones((Kd, 8), dtype=BF, device=dev)
    for A in (dys[1], dys[3]):
        C = torch.full((A.shape[1], 8), float("nan"), dtype=torch.float32, device=dev)
        jobs.append((A, ones, C))
        refs.append(A.double().sum(0)[:, None].expand(-1, 8))
    rc = K.gemm_group(jobs)
    assert rc == 0, rc
    for (A, B, C), ref in zip(jobs, refs):
        close(C.double(), ref, f"group+bias {tuple(C.shape)}", tol=2e-5)
