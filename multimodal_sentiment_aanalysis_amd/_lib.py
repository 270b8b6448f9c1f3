"""ctypes binding of libmmsa_hip.so (C ABI: include/mmsa.h).

The product path has no CPU fallback: if the shared library is missing, or a call returns a non-zero status,
this module raises. PyTorch is used only for device memory, streams and autograd plumbing.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMSA_LIB: another build of the same library (timing ablations: tools/microbench/ablate_nomfma.sh); default = the in-tree one
LIB_PATH = os.environ.get("MMSA_LIB") or os.path.join(_HERE, "lib", "libmmsa_hip.so")

MMSA_F32, MMSA_BF16, MMSA_FP8 = 0, 1, 2
GEMM_F32_SIMT, GEMM_BF16_MFMA, GEMM_BF16_SIMT, GEMM_F32_MFMA, GEMM_F32_VALU = 0, 1, 2, 3, 4
ACT_NONE, ACT_GELU, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4

_STATUS = {1: "bad argument", 2: "kernel launch failure", 3: "unsupported shape"}


class MmsaError(RuntimeError):
    pass


class ConvGeom(ctypes.Structure):
    _fields_ = [
        ("SH", ctypes.c_int32), ("SW", ctypes.c_int32), ("GH", ctypes.c_int32), ("GW", ctypes.c_int32),
        ("KH", ctypes.c_int32), ("KW", ctypes.c_int32), ("mul", ctypes.c_int32), ("kmul", ctypes.c_int32),
        ("off", ctypes.c_int32), ("div", ctypes.c_int32), ("cper", ctypes.c_int32),
        ("src_pix_stride", ctypes.c_int64),
    ]


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("lda", ctypes.c_int64), ("ldb", ctypes.c_int64), ("ldc", ctypes.c_int64),
        ("a_kmajor", ctypes.c_int32), ("b_kmajor", ctypes.c_int32), ("gather", ctypes.c_int32),
        ("b_tap_stride", ctypes.c_int64),
        ("geom", ConvGeom),
        ("bias", ctypes.c_void_p), ("C2", ctypes.c_void_p), ("ldc2", ctypes.c_int64), ("act", ctypes.c_int32),
        ("mul", ctypes.c_void_p), ("ldmul", ctypes.c_int64), ("add", ctypes.c_void_p), ("ldadd", ctypes.c_int64),
        ("out_f32", ctypes.c_int32), ("accumulate", ctypes.c_int32), ("split_k", ctypes.c_int32),
        ("ws", ctypes.c_void_p),
    ]


class BertCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("batch", "seq", "hidden", "layers", "heads", "intermediate", "vocab", "max_pos", "type_vocab",
                 "out_dim", "dtype")] + [("ln_eps", ctypes.c_float)]


class ResnetCfg(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int32), ("height", ctypes.c_int32), ("width", ctypes.c_int32),
                ("blocks", ctypes.c_int32 * 4), ("widths", ctypes.c_int32 * 4),
                ("out_dim", ctypes.c_int32), ("dtype", ctypes.c_int32), ("training", ctypes.c_int32),
                ("bn_eps", ctypes.c_float), ("bn_momentum", ctypes.c_float)]


class HeadCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("batch", "embed", "tokens", "heads", "pool_mode", "num_classes", "valence", "hidden", "out_dim",
                 "training")] + [("bn_eps", ctypes.c_float), ("bn_momentum", ctypes.c_float),
                                 ("ln_eps", ctypes.c_float), ("dropout_p", ctypes.c_float),
                                 ("seed", ctypes.c_uint64)]


HEAD_CROSS_MODAL, HEAD_MM_FUSION, HEAD_WEIGHTED, HEAD_CLASSIFIER, HEAD_PROJECTION = 0, 1, 2, 3, 4

RANGE_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64)  # mmsa_range_cb (include/mmsa.h)

_lib = None


def load():
    """Load the HIP library; raises MmsaError when it has not been built (python __graft_entry__.py / make)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MmsaError(
            f"{LIB_PATH} is missing: build it with `make -C multimodal_sentiment_aanalysis_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU fallback for the product path.")
    _lib = ctypes.CDLL(LIB_PATH)
    _lib.mmsa_abi_version.restype = ctypes.c_int
    _declare(_lib)
    return _lib


def _declare(L):
    sz = ctypes.c_size_t
    i32, i64, vp, f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_float
    sig = {
        "mmsa_gemm_ws_bytes": (sz, [i32, i32, i32]),
        "mmsa_gemm": (ctypes.c_int, [ctypes.POINTER(GemmDesc), i32, vp]),
        "mmsa_gemm_group": (ctypes.c_int, [ctypes.POINTER(GemmDesc), i32, vp]),
        "mmsa_gemm_group_split": (ctypes.c_int, [ctypes.POINTER(GemmDesc), i32, vp, ctypes.c_size_t, vp]),
        "mmsa_fp8_quantize_ws_bytes": (sz, []),
        "mmsa_fp8_quantize": (ctypes.c_int, [vp, i64, vp, vp, vp, vp]),
        "mmsa_gemm_fp8": (ctypes.c_int, [ctypes.POINTER(GemmDesc), vp, vp, vp]),
        "mmsa_fp8_quantize_rows": (ctypes.c_int, [vp, i64, i32, i32, vp, vp, vp]),
        "mmsa_gemm_fp8_rows": (ctypes.c_int, [ctypes.POINTER(GemmDesc), vp, vp, vp]),
        "mmsa_fp8_quantize_batch_ws_bytes": (sz, [i32]),
        "mmsa_fp8_quantize_batch": (ctypes.c_int, [vp, vp, vp, i32, vp, vp, vp, vp]),
        "mmsa_layernorm_fwd": (ctypes.c_int, [i32, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]),
        "mmsa_layernorm_bwd_ws_bytes": (sz, [i32]),
        "mmsa_layernorm_bwd": (ctypes.c_int, [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, vp]),
        "mmsa_colsum_ws_bytes": (sz, [i32]),
        "mmsa_colsum": (ctypes.c_int, [i32, vp, i64, vp, i32, vp, i32, i32, vp]),
        "mmsa_attention_bwd_ws_bytes": (sz, [i32, i32, i32]),
        "mmsa_attention_fwd": (ctypes.c_int, [i32, vp, vp, vp, i32, i32, i32, i32, vp]),
        "mmsa_attention_bwd": (ctypes.c_int, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    }
    P = ctypes.POINTER
    i64p, i32p, pp = P(ctypes.c_int64), P(ctypes.c_int32), P(ctypes.c_void_p)
    sig.update({
        "mmsa_bert_param_count": (ctypes.c_int, [P(BertCfg)]),
        "mmsa_bert_param_total": (i64, [P(BertCfg)]),
        "mmsa_bert_param_info": (ctypes.c_int, [P(BertCfg), ctypes.c_int, ctypes.c_char_p, ctypes.c_int, i64p, i32p, i64p]),
        "mmsa_bert_ws_bytes": (sz, [P(BertCfg)]),
        "mmsa_bert_fwd": (ctypes.c_int, [P(BertCfg), vp, vp, vp, vp, vp, vp, vp]),
        "mmsa_bert_bwd": (ctypes.c_int, [P(BertCfg), vp, vp, vp, vp, vp, vp, vp, i32, vp]),
        "mmsa_bert_bwd_cb": (ctypes.c_int, [P(BertCfg), vp, vp, vp, vp, vp, vp, vp, i32, vp, RANGE_CB, vp, i32, vp]),
        "mmsa_resnet_bwd_cb": (ctypes.c_int, [P(ResnetCfg), vp, vp, vp, vp, vp, i32, vp, RANGE_CB, vp, vp]),
        "mmsa_resnet_bwd_cb2": (ctypes.c_int, [P(ResnetCfg), vp, vp, vp, vp, vp, i32, vp, vp, RANGE_CB, vp, vp]),
        "mmsa_resnet_param_count": (ctypes.c_int, [P(ResnetCfg), i32]),
        "mmsa_resnet_param_total": (i64, [P(ResnetCfg), i32]),
        "mmsa_resnet_param_info": (ctypes.c_int, [P(ResnetCfg), i32, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, i64p, i32p, i64p]),
        "mmsa_resnet_ws_bytes": (sz, [P(ResnetCfg)]),
        "mmsa_resnet_ws_offset": (i64, [P(ResnetCfg), i32, i32]),
        "mmsa_resnet_fwd": (ctypes.c_int, [P(ResnetCfg), vp, vp, vp, vp, vp, vp, vp]),
        "mmsa_resnet_bwd": (ctypes.c_int, [P(ResnetCfg), vp, vp, vp, vp, vp, i32, vp]),
        "mmsa_head_param_count": (ctypes.c_int, [i32, P(HeadCfg), i32]),
        "mmsa_head_param_total": (i64, [i32, P(HeadCfg), i32]),
        "mmsa_head_param_info": (ctypes.c_int, [i32, P(HeadCfg), i32, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, i64p, i32p, i64p]),
        "mmsa_head_ws_bytes": (sz, [i32, P(HeadCfg)]),
        "mmsa_head_fwd": (ctypes.c_int, [i32, P(HeadCfg), vp, vp, pp, pp, vp, vp]),
        "mmsa_head_bwd": (ctypes.c_int, [i32, P(HeadCfg), vp, pp, pp, pp, vp, i32, vp, vp]),
        "mmsa_ce_fwd_bwd": (ctypes.c_int, [vp, vp, vp, vp, vp, i32, i32, f32, vp]),
        "mmsa_contrastive_ws_bytes": (sz, [i32, i32]),
        "mmsa_infonce_fwd_bwd": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp, vp]),
        "mmsa_supcon_fwd_bwd": (ctypes.c_int, [vp, vp, vp, f32, vp, vp, vp, i32, i32, f32, vp, vp]),
        "mmsa_grad_norm_ws_bytes": (sz, []),
        "mmsa_grad_norm": (ctypes.c_int, [vp, i64, f32, f32, vp, vp, vp]),
        "mmsa_adamw_step": (ctypes.c_int, [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp, f32, vp]),
        "mmsa_grad_norm_guard": (ctypes.c_int, [vp, i64, f32, f32, vp, vp, vp, vp, f32, f32, vp]),
        "mmsa_grad_norm_ranges": (ctypes.c_int, [vp, i64p, i64p, i32, f32, f32, vp, vp, vp, vp, f32, f32, vp]),
        "mmsa_grad_sumsq_ranges": (ctypes.c_int, [vp, i64p, i64p, i32, vp, vp, vp]),
        "mmsa_grad_norm_from_sumsq": (ctypes.c_int, [vp, i32, f32, f32, vp, vp, vp, f32, f32, vp]),
        "mmsa_grad_scale_clip": (ctypes.c_int, [vp, i64, vp, vp]),
        "mmsa_adamw_step_dev": (ctypes.c_int, [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, vp, f32, vp]),
        "mmsa_cast_f32": (ctypes.c_int, [i32, vp, vp, i64, vp]),
        "mmsa_widen_bf16": (ctypes.c_int, [vp, vp, i64, vp]),
        "mmsa_mfma_clock_probe": (ctypes.c_int, [vp, i32, i32, i32, vp]),
        "mmsa_prof_begin": (ctypes.c_int, [i32]),
        "mmsa_prof_sample": (ctypes.c_int, [i32, i32]),
        "mmsa_prof_mode": (ctypes.c_int, [i32]),
        "mmsa_prof_end": (ctypes.c_int, [P(ctypes.c_double), P(ctypes.c_double), i64p]),
        "mmsa_prof_last_bytes": (ctypes.c_double, []),
        "mmsa_bn_ws_bytes": (sz, [i32]),
        "mmsa_bn_fwd": (ctypes.c_int, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, i32, i32, vp]),
        "mmsa_bn_bwd": (ctypes.c_int, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, vp]),
        "mmsa_maxpool_fwd": (ctypes.c_int, [i32, vp, vp, vp, i32, i32, i32, i32, vp]),
        "mmsa_maxpool_bwd": (ctypes.c_int, [i32, vp, vp, vp, i32, i32, i32, i32, vp]),
        "mmsa_avgpool_fwd": (ctypes.c_int, [i32, vp, vp, i32, i32, i32, vp]),
        "mmsa_avgpool_bwd": (ctypes.c_int, [i32, vp, vp, i32, i32, i32, vp]),
        "mmsa_stem_im2col": (ctypes.c_int, [i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    })
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args


def check(status, what):
    if status != 0:
        raise MmsaError(f"{what} failed: {_STATUS.get(status, status)}")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def dtype_code(t):
    if t.dtype == torch.float32:
        return MMSA_F32
    if t.dtype == torch.bfloat16:
        return MMSA_BF16
    raise MmsaError(f"unsupported storage dtype {t.dtype}")


def ptr_array(tensors):
    """void*[] of device pointers (None -> NULL)."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def param_table(count_fn, info_fn):
    """Read a parameter layout table from the library: [(name, offset, shape tuple)]."""
    out = []
    name = ctypes.create_string_buffer(256)
    off, nd = ctypes.c_int64(), ctypes.c_int32()
    shape = (ctypes.c_int64 * 4)()
    n = count_fn()
    if n < 0:
        raise MmsaError("invalid engine configuration")
    for i in range(n):
        check(info_fn(i, name, 256, ctypes.byref(off), ctypes.byref(nd), shape), "param_info")
        out.append((name.value.decode(), off.value, tuple(shape[j] for j in range(nd.value))))
    return out
