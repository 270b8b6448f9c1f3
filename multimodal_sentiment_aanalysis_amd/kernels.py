"""Thin tensor-level wrappers over the C ABI (one Python function per `mmsa_*` entry point).

They take torch tensors only to obtain device pointers, shapes and the current HIP stream; all arithmetic
happens in libmmsa_hip.so. Used by the autograd Functions in this package and by the GPU parity tests.
"""
import ctypes

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, GEMM_BF16_MFMA, GEMM_BF16_SIMT,
                   GEMM_F32_SIMT, ConvGeom, GemmDesc, check, dtype_code, ptr, stream_ptr)  # noqa: F401

_ws_cache = {}


def workspace(nbytes, device, tag="ws"):
    """Grow-only scratch buffer per (device, tag); never shrinks, so steady-state steps allocate nothing."""
    key = (str(device), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def default_impl(t):
    return GEMM_BF16_MFMA if t.dtype == torch.bfloat16 else GEMM_F32_SIMT


def conv_geom(SH, SW, GH, GW, KH, KW, mul, kmul, off, div, cper, src_pix_stride):
    g = ConvGeom()
    g.SH, g.SW, g.GH, g.GW, g.KH, g.KW = SH, SW, GH, GW, KH, KW
    g.mul, g.kmul, g.off, g.div, g.cper, g.src_pix_stride = mul, kmul, off, div, cper, src_pix_stride
    return g


def gemm(A, B, C, M, N, K, lda, ldb, ldc, a_kmajor=0, b_kmajor=0, gather=0, geom=None, b_tap_stride=0, bias=None,
         C2=None, ldc2=0, act=ACT_NONE, mul=None, ldmul=0, add=None, ldadd=0, out_f32=0, accumulate=0, split_k=1,
         impl=None):
    """C[M,N] = epilogue(opA @ opB); see include/mmsa.h (mmsa_gemm). A/B/C are tensors (any view; pointers are used)."""
    L = _lib.load()
    d = GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.a_kmajor, d.b_kmajor, d.gather = a_kmajor, b_kmajor, gather
    d.b_tap_stride = b_tap_stride
    if geom is not None:
        d.geom = geom
    d.bias = bias.data_ptr() if bias is not None else None
    d.C2 = C2.data_ptr() if C2 is not None else None
    d.ldc2 = ldc2
    d.act = act
    d.mul = mul.data_ptr() if mul is not None else None
    d.ldmul = ldmul
    d.add = add.data_ptr() if add is not None else None
    d.ldadd = ldadd
    d.out_f32, d.accumulate, d.split_k = out_f32, accumulate, split_k
    if split_k > 1:
        ws = workspace(L.mmsa_gemm_ws_bytes(M, N, split_k), A.device, "splitk")
        d.ws = ws.data_ptr()
    if impl is None:
        impl = default_impl(A)
    check(L.mmsa_gemm(ctypes.byref(d), impl, stream_ptr()), "mmsa_gemm")
    return C


def fp8_quantize(x):
    """x: contiguous bf16 tensor (numel % 8 == 0) -> (e4m3 bytes as uint8, same shape; scale: 1-element fp32 tensor = amax / 448)."""
    L = _lib.load()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    scale = torch.empty(1, dtype=torch.float32, device=x.device)
    amax = torch.empty(L.mmsa_fp8_quantize_ws_bytes(), dtype=torch.uint8, device=x.device)
    check(L.mmsa_fp8_quantize(ptr(x), x.numel(), ptr(out), ptr(scale), ptr(amax), stream_ptr()), "mmsa_fp8_quantize")
    return out, scale


def fp8_quantize_rows(x):
    """x: [M, K] bf16 (row-contiguous view) -> (e4m3 bytes [M, K] uint8, per-row scales [M] fp32 = row amax / 448): one pass."""
    L = _lib.load()
    M, Kd = x.shape
    out = torch.empty((M, Kd), dtype=torch.uint8, device=x.device)
    scales = torch.empty(M, dtype=torch.float32, device=x.device)
    check(L.mmsa_fp8_quantize_rows(ptr(x), x.stride(0), M, Kd, ptr(out), ptr(scales), stream_ptr()), "mmsa_fp8_quantize_rows")
    return out, scales


def fp8_quantize_batch(base, offsets, numels):
    """Per-tensor quantization of the tensors base[off : off + n] (a flat bf16 buffer) in two launches: (e4m3 image of `base` with
    those ranges written, scales [len(offsets)])."""
    L = _lib.load()
    n = len(offsets)
    out = torch.zeros(base.numel(), dtype=torch.uint8, device=base.device)
    scales = torch.empty(n, dtype=torch.float32, device=base.device)
    ws = torch.empty(L.mmsa_fp8_quantize_batch_ws_bytes(n), dtype=torch.uint8, device=base.device)
    off = (ctypes.c_int64 * n)(*offsets)
    num = (ctypes.c_int64 * n)(*numels)
    check(L.mmsa_fp8_quantize_batch(ptr(base), off, num, n, ptr(out), ptr(scales), ptr(ws), stream_ptr()), "mmsa_fp8_quantize_batch")
    return out, scales


def gemm_fp8(Aq, sa, Bq, sb, C, bias=None, act=ACT_NONE, add=None, C2=None, row_scales=False):
    """C[M,N] = epilogue(sa * sb * Aq[M,K] Bq[N,K]^T) on e4m3 bytes (mmsa_gemm_fp8). Returns the status (3 = unsupported shape)."""
    L = _lib.load()
    M, Kd = Aq.shape
    N = Bq.shape[0]
    d = GemmDesc()
    d.A, d.B, d.C = Aq.data_ptr(), Bq.data_ptr(), C.data_ptr()
    d.M, d.N, d.K = M, N, Kd
    d.lda, d.ldb, d.ldc = Kd, Kd, N
    d.bias = bias.data_ptr() if bias is not None else None
    d.act = act
    d.add = add.data_ptr() if add is not None else None
    d.ldadd = N
    d.C2 = C2.data_ptr() if C2 is not None else None
    d.ldc2 = N
    d.out_f32 = int(C.dtype == torch.float32)
    d.split_k = 1
    if row_scales:  # sa: one scale per row of Aq (fp8_quantize_rows)
        return L.mmsa_gemm_fp8_rows(ctypes.byref(d), ptr(sa), ptr(sb), stream_ptr())
    return L.mmsa_gemm_fp8(ctypes.byref(d), ptr(sa), ptr(sb), stream_ptr())


def gemm_group(jobs, ws_bytes=0, geom=None, accumulate=0):
    """jobs: list of (A[K,M] k-major, B[K,N] k-major, C[M,N] fp32) -> one grouped weight-gradient launch (mmsa_gemm_group).
    ws_bytes > 0: with a K split planned inside a workspace of that size (mmsa_gemm_group_split); geom: every job is the weight
    gradient of a convolution with this geometry (gather 2: B is the [pixels][Cin] activation, C is [Cout][taps * Cin]).
    Returns the status code (3 = the library declined to group them)."""
    L = _lib.load()
    arr = (GemmDesc * len(jobs))()
    for d, (A, B, C) in zip(arr, jobs):
        Kd, M = A.shape
        N = C.shape[1]
        d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
        d.M, d.N, d.K = M, N, Kd
        d.lda, d.ldb, d.ldc = A.stride(0), B.stride(-2), C.stride(0)
        d.a_kmajor, d.b_kmajor, d.out_f32, d.split_k, d.accumulate = 1, 1, 1, 1, accumulate
        if geom is not None:
            d.gather = 2
            d.geom = geom
    if ws_bytes > 0:
        ws = workspace(ws_bytes, jobs[0][0].device, "group_splitk")
        return L.mmsa_gemm_group_split(arr, len(jobs), ws.data_ptr(), ws_bytes, stream_ptr())
    return L.mmsa_gemm_group(arr, len(jobs), stream_ptr())


def layernorm_fwd(x, gamma, beta, eps):
    L = _lib.load()
    M, H = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    check(L.mmsa_layernorm_fwd(dtype_code(x), ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), M, H,
                               eps, stream_ptr()), "mmsa_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma=None, dbeta=None, accumulate=0):
    L = _lib.load()
    M, H = x.shape
    dx = torch.empty_like(x)
    if dgamma is None:
        dgamma = torch.empty(H, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(H, dtype=torch.float32, device=x.device)
        accumulate = 0
    ws = workspace(L.mmsa_layernorm_bwd_ws_bytes(H), x.device, "ln")
    check(L.mmsa_layernorm_bwd(dtype_code(x), ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dx), ptr(dgamma),
                               ptr(dbeta), accumulate, ptr(ws), M, H, stream_ptr()), "mmsa_layernorm_bwd")
    return dx, dgamma, dbeta


def colsum(x, out=None, accumulate=0):
    L = _lib.load()
    M, N = x.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x.device)
        accumulate = 0
    ws = workspace(L.mmsa_colsum_ws_bytes(N), x.device, "colsum")
    check(L.mmsa_colsum(dtype_code(x), ptr(x), x.stride(0), ptr(out), accumulate, ptr(ws), M, N, stream_ptr()),
          "mmsa_colsum")
    return out


def attention_fwd(qkv, mask, B, S, heads, impl=None):
    L = _lib.load()
    if impl is None:
        impl = default_impl(qkv)
    ctx = torch.empty(B * S, heads * 64, dtype=qkv.dtype, device=qkv.device)
    check(L.mmsa_attention_fwd(impl, ptr(qkv), ptr(mask), ptr(ctx), B, S, heads, 64, stream_ptr()),
          "mmsa_attention_fwd")
    return ctx


def attention_bwd(qkv, mask, dctx, B, S, heads, impl=None):
    L = _lib.load()
    if impl is None:
        impl = default_impl(qkv)
    dqkv = torch.empty_like(qkv)
    ws = workspace(L.mmsa_attention_bwd_ws_bytes(B, S, heads), qkv.device, "attn")
    check(L.mmsa_attention_bwd(impl, ptr(qkv), ptr(mask), ptr(dctx), ptr(dqkv), ptr(ws), B, S, heads, 64,
                               stream_ptr()), "mmsa_attention_bwd")
    return dqkv


def bn_fwd(x, gamma, beta, running_mean, running_var, res=None, act=ACT_NONE, training=True, eps=1e-5, momentum=0.1):
    L = _lib.load()
    M, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = workspace(L.mmsa_bn_ws_bytes(C), x.device, "bn")
    check(L.mmsa_bn_fwd(dtype_code(x), ptr(x), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(mean),
                        ptr(invstd), ptr(res), ptr(y), ptr(ws), M, C, eps, momentum, act, int(training), stream_ptr()),
          "mmsa_bn_fwd")
    return y, mean, invstd


def bn_bwd(dy, x, y, mean, invstd, gamma, beta, act=ACT_NONE, training=True, want_dres=False):
    L = _lib.load()
    M, C = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = workspace(L.mmsa_bn_ws_bytes(C), x.device, "bn")
    check(L.mmsa_bn_bwd(dtype_code(x), ptr(dy), ptr(x), ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(dx),
                        ptr(dres), ptr(dgamma), ptr(dbeta), 0, ptr(ws), M, C, act, int(training), stream_ptr()),
          "mmsa_bn_bwd")
    return dx, dres, dgamma, dbeta


def maxpool_fwd(x, B, H, W, C):
    L = _lib.load()
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty(B * OH * OW, C, dtype=x.dtype, device=x.device)
    idx = torch.empty(B * OH * OW, C, dtype=torch.uint8, device=x.device)
    check(L.mmsa_maxpool_fwd(dtype_code(x), ptr(x), ptr(y), ptr(idx), B, H, W, C, stream_ptr()), "mmsa_maxpool_fwd")
    return y, idx


def maxpool_bwd(dy, idx, B, H, W, C):
    L = _lib.load()
    dx = torch.empty(B * H * W, C, dtype=dy.dtype, device=dy.device)
    check(L.mmsa_maxpool_bwd(dtype_code(dy), ptr(dy), ptr(idx), ptr(dx), B, H, W, C, stream_ptr()), "mmsa_maxpool_bwd")
    return dx
