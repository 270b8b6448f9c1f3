"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes).

References are computed on the CPU in float64 from the same (already storage-rounded) inputs, so the only
differences are accumulation order and the final rounding to the storage type.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from multimodal_sentiment_aanalysis_amd import kernels as K
from multimodal_sentiment_aanalysis_amd._lib import (ACT_GELU, ACT_NONE, ACT_RELU, ACT_TANH, GEMM_BF16_MFMA,
                                                     GEMM_BF16_SIMT, GEMM_F32_MFMA, GEMM_F32_SIMT, GEMM_F32_VALU)

IMPLS = [(GEMM_F32_SIMT, torch.float32), (GEMM_BF16_MFMA, torch.bfloat16), (GEMM_BF16_SIMT, torch.bfloat16),
         (GEMM_F32_MFMA, torch.float32)]


class disabled:
    """MMSA_DISABLE=<names> for the calls inside (the library reads the variable at every call)."""

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


def rnd(shape, dtype, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


def assert_close(got, ref, dtype, what, k=1):
    got = got.detach().cpu().double()
    ref = ref.double()
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item() / scale
    tol = 1.2e-2 if dtype == torch.bfloat16 else 2e-5
    assert err < tol, f"{what}: rel-to-max err {err:.3e} >= {tol}"


@pytest.mark.parametrize("impl,dtype", IMPLS)
@pytest.mark.parametrize("M,N,Kd", [(256, 256, 128), (300, 136, 72), (128, 64, 64), (1024, 768, 768), (77, 260, 200)])
def test_gemm_nt(dev, impl, dtype, M, N, Kd):
    A = rnd((M, Kd), dtype, dev, 1)
    B = rnd((N, Kd), dtype, dev, 2)
    C = torch.empty(M, N, dtype=dtype, device=dev)
    K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, impl=impl)
    assert_close(C, A.cpu().double() @ B.cpu().double().T, dtype, "NT")


@pytest.mark.parametrize("impl,dtype", IMPLS)
@pytest.mark.parametrize("M,N,Kd", [(256, 256, 128), (300, 136, 72), (512, 768, 3072)])
def test_gemm_nn(dev, impl, dtype, M, N, Kd):
    A = rnd((M, Kd), dtype, dev, 1)
    B = rnd((Kd, N), dtype, dev, 2)  # stored [K][N]
    C = torch.empty(M, N, dtype=dtype, device=dev)
    K.gemm(A, B, C, M, N, Kd, Kd, N, N, b_kmajor=1, impl=impl)
    assert_close(C, A.cpu().double() @ B.cpu().double(), dtype, "NN")


@pytest.mark.parametrize("impl,dtype", IMPLS)
@pytest.mark.parametrize("M,N,Kd,split", [(256, 256, 128, 1), (136, 264, 200, 1), (768, 768, 4096, 4), (64, 576, 1000, 3)])
def test_gemm_tn_f32out(dev, impl, dtype, M, N, Kd, split):
    A = rnd((Kd, M), dtype, dev, 1)  # stored [K][M]
    B = rnd((Kd, N), dtype, dev, 2)  # stored [K][N]
    C0 = rnd((M, N), torch.float32, dev, 3)
    C = C0.clone()
    K.gemm(A, B, C, M, N, Kd, M, N, N, a_kmajor=1, b_kmajor=1, out_f32=1, accumulate=1, split_k=split, impl=impl)
    ref = C0.cpu().double() + A.cpu().double().T @ B.cpu().double()
    assert_close(C, ref, torch.float32, "TN accumulate", k=Kd)


@pytest.mark.parametrize("impl,dtype", IMPLS)
def test_gemm_tn_kcontig_b(dev, impl, dtype):
    M, N, Kd = 128, 256, 192
    A = rnd((Kd, M), dtype, dev, 1)
    B = rnd((N, Kd), dtype, dev, 2)
    C = torch.empty(M, N, dtype=dtype, device=dev)
    K.gemm(A, B, C, M, N, Kd, M, Kd, N, a_kmajor=1, b_kmajor=0, impl=impl)
    assert_close(C, A.cpu().double().T @ B.cpu().double().T, dtype, "TN/kc")


@pytest.mark.parametrize("impl,dtype", IMPLS)
def test_gemm_epilogue(dev, impl, dtype):
    M, N, Kd = 200, 192, 128
    A = rnd((M, Kd), dtype, dev, 1, 0.3)
    B = rnd((N, Kd), dtype, dev, 2, 0.3)
    bias = rnd((N,), torch.float32, dev, 3)
    add = rnd((M, N), dtype, dev, 4)
    pre = rnd((M, N), dtype, dev, 5)
    acc = A.cpu().double() @ B.cpu().double().T + bias.cpu().double()
    # bias + GELU with pre-activation side output
    C = torch.empty(M, N, dtype=dtype, device=dev)
    C2 = torch.empty(M, N, dtype=dtype, device=dev)
    K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias, C2=C2, ldc2=N, act=ACT_GELU, impl=impl)
    assert_close(C2, acc, dtype, "pre-activation")
    assert_close(C, F.gelu(acc), dtype, "gelu")
    # bias + residual
    K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias, add=add, ldadd=N, impl=impl)
    assert_close(C, acc + add.cpu().double(), dtype, "bias+residual")
    # gelu' multiply (FFN backward) + add
    x = pre.cpu().double()
    gp = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, mul=pre, ldmul=N, add=add, ldadd=N, impl=impl)
    assert_close(C, (A.cpu().double() @ B.cpu().double().T) * gp + add.cpu().double(), dtype, "gelu-grad")
    # tanh / relu, fp32 output
    Cf = torch.empty(M, N, dtype=torch.float32, device=dev)
    K.gemm(A, B, Cf, M, N, Kd, Kd, Kd, N, bias=bias, act=ACT_TANH, out_f32=1, impl=impl)
    assert_close(Cf, torch.tanh(acc), torch.float32 if dtype == torch.float32 else torch.bfloat16, "tanh f32")
    K.gemm(A, B, Cf, M, N, Kd, Kd, Kd, N, bias=bias, act=ACT_RELU, out_f32=1, impl=impl)
    assert_close(Cf, torch.relu(acc), torch.float32 if dtype == torch.float32 else torch.bfloat16, "relu f32")


def test_gemm_f32_n3(dev):
    M, N, Kd = 16, 3, 128
    A = rnd((M, Kd), torch.float32, dev, 1)
    B = rnd((N, Kd), torch.float32, dev, 2)
    bias = rnd((N,), torch.float32, dev, 3)
    C = torch.empty(M, N, dtype=torch.float32, device=dev)
    K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias, impl=GEMM_F32_SIMT)
    assert_close(C, A.cpu().double() @ B.cpu().double().T + bias.cpu().double(), torch.float32, "N=3")


@pytest.mark.parametrize("dtype,impl", [(torch.float32, GEMM_F32_SIMT), (torch.bfloat16, GEMM_BF16_MFMA)])
@pytest.mark.parametrize("M,N,Kd,lay", [(64, 256, 256, "NT"), (64, 256, 768, "NT"), (128, 768, 256, "NT"), (64, 3, 128, "NT"),
                                        (64, 768, 768, "NT"), (50, 72, 200, "NT"), (64, 256, 2048, "NT"), (64, 768, 256, "NN"),
                                        (128, 256, 768, "NN"), (64, 128, 8, "NN"), (256, 768, 64, "TN"), (768, 256, 128, "TN"),
                                        (8, 128, 64, "TN"), (128, 256, 192, "TK")])
def test_gemm_batch_rows(dev, dtype, impl, M, N, Kd, lay):
    """Batch-row problems (fusion head, pooler, projections) take the one-launch 16x16-per-wave MFMA kernel
    (gemm_f32_tiny.hip) through the ordinary entry point: every operand layout, K tails inside a 64-chunk, partial tiles,
    bias + accumulate epilogues; checked against the float64 product at storage resolution."""
    f32 = torch.float32
    if lay == "NT":
        A, B = rnd((M, Kd), dtype, dev, 1), rnd((N, Kd), dtype, dev, 2)
        kw, ref = dict(lda=Kd, ldb=Kd), A.cpu().double() @ B.cpu().double().T
    elif lay == "NN":
        A, B = rnd((M, Kd), dtype, dev, 1), rnd((Kd, N), dtype, dev, 2)
        kw, ref = dict(lda=Kd, ldb=N, b_kmajor=1), A.cpu().double() @ B.cpu().double()
    elif lay == "TN":
        A, B = rnd((Kd, M), dtype, dev, 1), rnd((Kd, N), dtype, dev, 2)
        kw, ref = dict(lda=M, ldb=N, a_kmajor=1, b_kmajor=1), A.cpu().double().T @ B.cpu().double()
    else:
        A, B = rnd((Kd, M), dtype, dev, 1), rnd((N, Kd), dtype, dev, 2)
        kw, ref = dict(lda=M, ldb=Kd, a_kmajor=1), A.cpu().double().T @ B.cpu().double().T
    bias = rnd((N,), f32, dev, 3)
    if dtype == torch.bfloat16 and N % 4:
        pytest.skip("bf16 launchers take N % 4 == 0")
    C = torch.empty(M, N, dtype=dtype, device=dev)
    K.gemm(A, B, C, M, N, Kd, kw["lda"], kw["ldb"], N, a_kmajor=kw.get("a_kmajor", 0), b_kmajor=kw.get("b_kmajor", 0),
           bias=bias, impl=impl)
    assert_close(C, ref + bias.cpu().double(), dtype, f"{lay} bias", k=Kd)
    C0 = rnd((M, N), f32, dev, 4)
    Cf = C0.clone()
    K.gemm(A, B, Cf, M, N, Kd, kw["lda"], kw["ldb"], N, a_kmajor=kw.get("a_kmajor", 0), b_kmajor=kw.get("b_kmajor", 0),
           out_f32=1, accumulate=1, impl=impl)
    assert_close(Cf, C0.cpu().double() + ref, f32 if dtype == f32 else torch.bfloat16, f"{lay} accumulate", k=Kd)


@pytest.mark.parametrize("case", ["nt", "nt_k16", "nt_split", "nn", "tn_split", "tn_kc", "edge", "epilogue"])
def test_gemm_f32_mfma_matches_valu_bitwise(dev, case):
    """The exact-fp32 matrix-core kernel (v_mfma_f32_32x32x2_f32, gemm_f32_mfma.hip) against the VALU-fma kernel: the MFMA
    is a k-ordered fmaf chain and both kernels use the same K order and split-K partition, so the results must be EQUAL
    BIT FOR BIT — every fast loader (k-contiguous, m/n-contiguous), the generic edge loaders, slabs + reducer, epilogues."""
    f32 = torch.float32
    if case == "nt":
        M, N, Kd = 1024, 768, 768
        A, B = rnd((M, Kd), f32, dev, 1), rnd((N, Kd), f32, dev, 2)
        run = lambda C, impl: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, impl=impl)  # noqa: E731
    elif case == "nt_k16":  # K a multiple of 16 only
        M, N, Kd = 256, 384, 208
        A, B = rnd((M, Kd), f32, dev, 1), rnd((N, Kd), f32, dev, 2)
        run = lambda C, impl: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, impl=impl)  # noqa: E731
    elif case == "nt_split":  # split-K slabs, rows / columns past the tile edge
        M, N, Kd = 200, 264, 2048
        A, B = rnd((M, Kd), f32, dev, 1), rnd((N, Kd), f32, dev, 2)
        run = lambda C, impl: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, out_f32=1, split_k=4, impl=impl)  # noqa: E731
    elif case == "nn":
        M, N, Kd = 512, 768, 3072
        A, B = rnd((M, Kd), f32, dev, 1), rnd((Kd, N), f32, dev, 2)
        run = lambda C, impl: K.gemm(A, B, C, M, N, Kd, Kd, N, N, b_kmajor=1, impl=impl)  # noqa: E731
    elif case == "tn_split":
        M, N, Kd = 768, 768, 4096
        A, B = rnd((Kd, M), f32, dev, 1), rnd((Kd, N), f32, dev, 2)
        run = lambda C, impl: K.gemm(A, B, C, M, N, Kd, M, N, N, a_kmajor=1, b_kmajor=1, out_f32=1, split_k=4, impl=impl)  # noqa: E731
    elif case == "tn_kc":
        M, N, Kd = 128, 256, 192
        A, B = rnd((Kd, M), f32, dev, 1), rnd((N, Kd), f32, dev, 2)
        run = lambda C, impl: K.gemm(A, B, C, M, N, Kd, M, Kd, N, a_kmajor=1, b_kmajor=0, impl=impl)  # noqa: E731
    elif case == "edge":  # nothing aligned: generic loaders, scalar epilogue
        M, N, Kd = 77, 261, 203
        A, B = rnd((M, Kd), f32, dev, 1), rnd((N, Kd), f32, dev, 2)
        run = lambda C, impl: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, impl=impl)  # noqa: E731
    else:
        M, N, Kd = 200, 192, 128
        A, B = rnd((M, Kd), f32, dev, 1, 0.3), rnd((N, Kd), f32, dev, 2, 0.3)
        bias, add = rnd((N,), f32, dev, 3), rnd((M, N), f32, dev, 4)
        run = lambda C, impl: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias, act=ACT_GELU, add=add, ldadd=N, impl=impl)  # noqa: E731
    C1 = torch.zeros(M, N, dtype=f32, device=dev)
    C2 = torch.zeros(M, N, dtype=f32, device=dev)
    run(C1, GEMM_F32_MFMA)
    run(C2, GEMM_F32_VALU)
    assert torch.isfinite(C1).all() and C1.abs().max() > 0
    assert torch.equal(C1, C2), f"{case}: max |diff| {(C1 - C2).abs().max().item():.3e}"


@pytest.mark.parametrize("M,N,Kd", [(256, 256, 128), (1024, 768, 768), (300, 136, 256), (8192, 3072, 768), (77, 260, 384)])
def test_gemm_fp8(dev, M, N, Kd):
    """BASELINE.json configs[4]: the fp8 (OCP e4m3) quantizer and NT GEMM (gemm_fp8.hip). (1) the quantizer's bytes are torch's
    own float8_e4m3fn rounding of x / scale with scale = amax / 448; (2) the GEMM equals the float64 product of the DEQUANTIZED
    operands (what the MFMA multiplies is exactly those values; fp32 accumulation, one bf16 output rounding) — this also pins
    the operand lane maps of v_mfma_f32_16x16x32_fp8_fp8 and the k-permuted fragment reads; (3) bias + GELU + residual epilogue."""
    A = rnd((M, Kd), torch.bfloat16, dev, 1)
    B = rnd((N, Kd), torch.bfloat16, dev, 2, 0.05)
    Aq, sa = K.fp8_quantize(A)
    Bq, sb = K.fp8_quantize(B)
    for x, q, s_ in ((A, Aq, sa), (B, Bq, sb)):
        amax = x.float().abs().max()
        assert abs(s_.item() - amax.item() / 448.0) <= 1e-6 * amax.item()
        want = (x.float() / s_).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
        frac = (want != q).float().mean().item()
        assert frac < 1e-3, f"quantizer differs from torch's e4m3 rounding on {frac:.2%} of the elements"
    Ad = Aq.view(torch.float8_e4m3fn).float().cpu().double() * sa.item()
    Bd = Bq.view(torch.float8_e4m3fn).float().cpu().double() * sb.item()
    ref = Ad @ Bd.T
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    assert K.gemm_fp8(Aq, sa, Bq, sb, C) == 0
    assert_close(C, ref, torch.bfloat16, "fp8 NT")
    # the fp8 result vs the bf16 operands' exact product: e4m3 keeps 3 mantissa bits (about 4-6 % of the output's rms)
    exact = A.cpu().double() @ B.cpu().double().T
    rel = ((C.cpu().double() - exact).norm() / exact.norm()).item()
    assert rel < 0.08, rel
    bias = rnd((N,), torch.float32, dev, 3)
    add = rnd((M, N), torch.bfloat16, dev, 4)
    assert K.gemm_fp8(Aq, sa, Bq, sb, C, bias=bias, act=ACT_GELU, add=add) == 0
    assert_close(C, F.gelu(ref + bias.cpu().double()) + add.cpu().double(), torch.bfloat16, "fp8 bias+gelu+residual")
    Cf = torch.empty(M, N, dtype=torch.float32, device=dev)
    assert K.gemm_fp8(Aq, sa, Bq, sb, Cf) == 0
    assert_close(Cf, ref, torch.float32 if False else torch.bfloat16, "fp8 NT fp32 out")
    # fp32 output: no output rounding left — what remains (measured 2-3e-5 of the largest entry, an order above fp32 summation
    # noise) is the fp8 MFMA's internal alignment of the 32 products of one instruction before they reach the fp32 accumulator
    assert ((Cf.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 1e-4


@pytest.mark.parametrize("M,N,Kd", [(256, 256, 128), (1024, 768, 768), (300, 136, 256), (8192, 768, 3072), (77, 260, 384), (512, 1024, 4096)])
def test_gemm_fp8_per_token_scales(dev, M, N, Kd):
    """The forms the text encoder's forward uses: per-TOKEN activation scales in one pass (mmsa_fp8_quantize_rows), per-tensor
    weight scales for many tensors in two launches (mmsa_fp8_quantize_batch), and the NT GEMM whose A operand carries one scale
    per row (mmsa_gemm_fp8_rows). Bytes = torch's float8_e4m3fn rounding of x / scale; GEMM = float64 product of the dequantized
    operands. Rows of A are a strided view (the engine quantizes column slices of wider buffers)."""
    wide = rnd((M, Kd + 64), torch.bfloat16, dev, 1)
    wide[:, 7] *= 30.0  # rows with very different ranges: what per-token scales are for
    wide[3] *= 0.01
    A = wide[:, :Kd]
    Aq, sa = K.fp8_quantize_rows(A)
    amax = A.float().abs().amax(dim=1)
    assert torch.allclose(sa, amax / 448.0, rtol=1e-6, atol=0)
    inv = torch.tensor(448.0, device=dev) / amax  # the kernel's own expression (x * (448 / amax), clamp, round to nearest even)
    want = (A.float() * inv[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    frac = (want != Aq).float().mean().item()
    assert frac < 1e-4, f"row quantizer differs from torch's e4m3 rounding on {frac:.3%} of the elements"
    # three weights inside one flat buffer, quantized together
    flat = rnd((3 * N * Kd + 64,), torch.bfloat16, dev, 2, 0.05)
    offs = [0, N * Kd + 32, 2 * N * Kd + 64]
    flat[offs[1]:offs[1] + N * Kd] *= 4.0
    img, sb = K.fp8_quantize_batch(flat, offs, [N * Kd] * 3)
    for t, off in enumerate(offs):
        w = flat[off:off + N * Kd].float()
        assert abs(sb[t].item() - w.abs().max().item() / 448.0) <= 1e-6 * w.abs().max().item()
        want = (w * (torch.tensor(448.0, device=dev) / w.abs().max())).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
        frac = (want != img[off:off + N * Kd]).float().mean().item()
        assert frac < 1e-4, f"batch quantizer, tensor {t}: {frac:.3%} of the bytes differ"
    Bq = img[offs[1]:offs[1] + N * Kd].view(N, Kd)
    Ad = Aq.view(torch.float8_e4m3fn).float().cpu().double() * sa.cpu().double()[:, None]
    Bd = Bq.view(torch.float8_e4m3fn).float().cpu().double() * sb[1].item()
    ref = Ad @ Bd.T
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    assert K.gemm_fp8(Aq, sa, Bq, sb[1:2], C, row_scales=True) == 0
    assert_close(C, ref, torch.bfloat16, "fp8 NT, per-token scales")
    bias = rnd((N,), torch.float32, dev, 3)
    add = rnd((M, N), torch.bfloat16, dev, 4)
    assert K.gemm_fp8(Aq, sa, Bq, sb[1:2], C, bias=bias, act=ACT_GELU, add=add, row_scales=True) == 0
    assert_close(C, F.gelu(ref + bias.cpu().double()) + add.cpu().double(), torch.bfloat16, "fp8 per-token bias+gelu+residual")
    # per-token scales keep a small-range row's precision: row 3 (1 % of the others' range) against the exact bf16 product
    exact = A.cpu().double() @ (flat[offs[1]:offs[1] + N * Kd].view(N, Kd).cpu().double()).T
    rel3 = ((ref[3] - exact[3]).norm() / exact[3].norm()).item()
    assert rel3 < 0.08, rel3


CONVS = [  # B, H, W, Cin, Cout, k, stride, pad
    (2, 8, 8, 64, 64, 3, 1, 1),
    (2, 9, 7, 64, 128, 3, 2, 1),
    (3, 8, 8, 128, 64, 1, 1, 0),
    (2, 8, 8, 64, 128, 1, 2, 0),
    (2, 14, 14, 128, 128, 3, 1, 1),
    (2, 16, 16, 64, 64, 3, 2, 1),
    (2, 8, 8, 128, 128, 3, 2, 1),
    (2, 16, 16, 256, 256, 1, 2, 0),
    (2, 16, 16, 256, 64, 1, 1, 0),
]


@pytest.mark.parametrize("impl,dtype", IMPLS)
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_implicit_gemm(dev, impl, dtype, cfg):
    B, H, W, Cin, Cout, k, s, p = cfg
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = rnd((B, H, W, Cin), dtype, dev, 1)
    w = rnd((Cout, k, k, Cin), dtype, dev, 2, 0.1)
    dy = rnd((B, OH, OW, Cout), dtype, dev, 3)
    xr = x.cpu().double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.cpu().double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.conv2d(xr, wr, stride=s, padding=p)
    yr.backward(dy.cpu().double().permute(0, 3, 1, 2))
    Kd = k * k * Cin
    # forward
    y = torch.empty(B * OH * OW, Cout, dtype=dtype, device=dev)
    g = K.conv_geom(H, W, OH, OW, k, k, s, 1, -p, 1, Cin, Cin)
    K.gemm(x, w, y, B * OH * OW, Cout, Kd, Cin, Kd, Cout, gather=1, geom=g, impl=impl)
    assert_close(y.view(B, OH, OW, Cout), yr.detach().permute(0, 2, 3, 1), dtype, "conv fwd")
    # data gradient
    dx = torch.empty(B * H * W, Cin, dtype=dtype, device=dev)
    g = K.conv_geom(OH, OW, H, W, k, k, 1, -1, p, s, Cout, Cout)
    K.gemm(dy, w, dx, B * H * W, Cin, k * k * Cout, Cout, Kd, Cin, b_kmajor=1, gather=1, geom=g, b_tap_stride=Cin,
           impl=impl)
    assert_close(dx.view(B, H, W, Cin), xr.grad.permute(0, 2, 3, 1), dtype, "conv dgrad")
    # weight gradient (fp32 out, split-K)
    dw = torch.zeros(Cout, Kd, dtype=torch.float32, device=dev)
    g = K.conv_geom(H, W, OH, OW, k, k, s, 1, -p, 1, Cin, Cin)
    K.gemm(dy, x, dw, Cout, Kd, B * OH * OW, Cout, Cin, Kd, a_kmajor=1, b_kmajor=1, gather=2, geom=g, out_f32=1,
           split_k=2, impl=impl)
    assert_close(dw.view(Cout, k, k, Cin), wr.grad.permute(0, 2, 3, 1), torch.float32, "conv wgrad")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,H", [(64, 256), (515, 768), (33, 1024), (8, 2048)])
def test_layernorm(dev, dtype, M, H):
    x = rnd((M, H), dtype, dev, 1, 2.0)
    gamma = rnd((H,), torch.float32, dev, 2)
    beta = rnd((H,), torch.float32, dev, 3)
    dy = rnd((M, H), dtype, dev, 4)
    xr = x.cpu().double().requires_grad_(True)
    gr = gamma.cpu().double().requires_grad_(True)
    br = beta.cpu().double().requires_grad_(True)
    yr = F.layer_norm(xr, (H,), gr, br, 1e-12)
    yr.backward(dy.cpu().double())
    y, mean, rstd = K.layernorm_fwd(x, gamma, beta, 1e-12)
    assert_close(y, yr.detach(), dtype, "ln fwd")
    dx, dg, db = K.layernorm_bwd(dy, x, mean, rstd, gamma)
    assert_close(dx, xr.grad, dtype, "ln dx")
    assert_close(dg, gr.grad, torch.float32, "ln dgamma")
    assert_close(db, br.grad, torch.float32, "ln dbeta")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum(dev, dtype):
    x = rnd((1000, 2304), dtype, dev, 1)
    out = rnd((2304,), torch.float32, dev, 2)
    ref = out.cpu().double() + x.cpu().double().sum(0)
    K.colsum(x, out, accumulate=1)
    assert_close(out, ref, torch.float32, "colsum")


def attn_ref(qkv, mask, dctx, B, S, heads):
    Hd = heads * 64
    t = qkv.cpu().double().view(B, S, 3, heads, 64).requires_grad_(True)
    q, k, v = t[:, :, 0].transpose(1, 2), t[:, :, 1].transpose(1, 2), t[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(-1, -2) / 8.0
    if mask is not None:
        s = s.masked_fill(mask.cpu()[:, None, None, :] == 0, -1e30)
    p = torch.softmax(s, -1)
    ctx = (p @ v).transpose(1, 2).reshape(B * S, Hd)
    ctx.backward(dctx.cpu().double())
    return ctx.detach(), t.grad.reshape(B * S, 3 * Hd)


@pytest.mark.parametrize("impl,dtype", IMPLS[:3])  # attention: fp32 SIMT, bf16 MFMA, bf16 SIMT (no GEMM-only impls)
@pytest.mark.parametrize("B,S,heads,masked", [(2, 128, 2, False), (3, 32, 4, True), (2, 64, 1, True), (1, 16, 2, False),
                                              (2, 48, 2, True), (1, 256, 1, False),
                                              # S > 128: the MFMA backward is the recompute variant (attn_bwd_mfma_rc_kernel)
                                              (2, 256, 2, True), (1, 192, 2, True), (3, 256, 4, False)])
def test_attention(dev, impl, dtype, B, S, heads, masked):
    Hd = heads * 64
    qkv = rnd((B * S, 3 * Hd), dtype, dev, 1)
    dctx = rnd((B * S, Hd), dtype, dev, 2)
    mask = None
    if masked:
        mask = torch.ones(B, S)
        for b in range(B):
            mask[b, S - 3 - 2 * b:] = 0
        mask = mask.to(dev)
    ctx_ref, dqkv_ref = attn_ref(qkv, mask, dctx, B, S, heads)
    ctx = K.attention_fwd(qkv, mask, B, S, heads, impl=impl)
    tol_dtype = dtype
    assert_close(ctx, ctx_ref, tol_dtype, "attn fwd")
    dqkv = K.attention_bwd(qkv, mask, dctx, B, S, heads, impl=impl)
    got = dqkv.cpu().double()
    scale = dqkv_ref.abs().max().item()
    err = (got - dqkv_ref).abs().max().item() / scale
    assert err < (3e-2 if dtype == torch.bfloat16 else 1e-4), f"attn bwd err {err}"


@pytest.mark.parametrize("B,S,heads", [(3, 128, 12), (2, 256, 4)])
def test_attention_forward_eight_wave_workgroups_match_four_wave(dev, B, S, heads):
    """S % 128 == 0: 8 waves per workgroup (K / V staged once per 128 queries) against the 4-wave launch (MMSA_DISABLE=attn_fwd8):
    the per-wave code is the same -> equal bits."""
    qkv = rnd((B * S, 3 * heads * 64), torch.bfloat16, dev, 1)
    mask = torch.ones(B, S, device=dev)
    mask[:, S - 5:] = 0
    a = K.attention_fwd(qkv, mask, B, S, heads)
    with disabled("attn_fwd8"):
        b = K.attention_fwd(qkv, mask, B, S, heads)
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,C,act,use_res", [(512, 64, ACT_RELU, False), (128, 64, ACT_RELU, True), (2048, 256, ACT_RELU, True),
                                             (32, 512, ACT_NONE, False), (8, 128, ACT_RELU, False), (16, 256, ACT_GELU, False),
                                             (64, 256, ACT_GELU, False), (200, 128, ACT_NONE, False), (256, 64, ACT_GELU, False),
                                             (1000, 192, ACT_RELU, True)])
def test_batchnorm(dev, dtype, M, C, act, use_res):
    x = rnd((M, C), dtype, dev, 1, 2.0)
    res = rnd((M, C), dtype, dev, 2) if use_res else None
    gamma = (1 + 0.1 * rnd((C,), torch.float32, dev, 3))
    beta = 0.1 * rnd((C,), torch.float32, dev, 4)
    dy = rnd((M, C), dtype, dev, 5)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    xr = x.cpu().double().requires_grad_(True)
    rr = res.cpu().double().requires_grad_(True) if use_res else None
    gr, br = gamma.cpu().double().requires_grad_(True), beta.cpu().double().requires_grad_(True)
    rmr, rvr = torch.zeros(C, dtype=torch.double), torch.ones(C, dtype=torch.double)
    z = F.batch_norm(xr, rmr, rvr, gr, br, True, 0.1, 1e-5)
    if use_res:
        z = z + rr
    yr = torch.relu(z) if act == ACT_RELU else (F.gelu(z) if act == ACT_GELU else z)
    yr.backward(dy.cpu().double())
    y, mean, invstd = K.bn_fwd(x, gamma, beta, rm, rv, res, act, True)
    assert_close(y, yr.detach(), dtype, "bn fwd")
    assert_close(rm, rmr, torch.float32, "running mean")
    assert_close(rv, rvr, torch.float32, "running var")
    dx, dres, dg, db = K.bn_bwd(dy, x, y, mean, invstd, gamma, beta, act, True, want_dres=use_res)
    assert_close(dx, xr.grad, dtype, "bn dx")
    assert_close(dg, gr.grad, torch.float32 if dtype == torch.float32 else torch.bfloat16, "bn dgamma")
    assert_close(db, br.grad, torch.float32 if dtype == torch.float32 else torch.bfloat16, "bn dbeta")
    if use_res:
        assert_close(dres, rr.grad, dtype, "bn dres")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C", [(2, 32, 32, 64), (1, 9, 7, 64), (3, 16, 16, 128)])
def test_maxpool(dev, dtype, B, H, W, C):
    x = torch.relu(rnd((B, H, W, C), dtype, dev, 1))  # many exact ties at 0, like the post-ReLU stem output
    xr = x.cpu().double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    OH, OW = yr.shape[2], yr.shape[3]
    dy = rnd((B, OH, OW, C), dtype, dev, 2)
    yr.backward(dy.cpu().double().permute(0, 3, 1, 2))
    y, idx = K.maxpool_fwd(x.view(-1, C), B, H, W, C)
    assert_close(y.view(B, OH, OW, C), yr.detach().permute(0, 2, 3, 1), dtype, "maxpool fwd")
    dx = K.maxpool_bwd(dy.view(-1, C), idx, B, H, W, C)
    # exact scatter in fp32; in bf16 the up-to-4 overlapping contributions are summed in fp32 and rounded once
    tol = 1e-6 if dtype == torch.float32 else 3e-2
    assert (dx.view(B, H, W, C).cpu().double() - xr.grad.permute(0, 2, 3, 1)).abs().max().item() < tol
