"""CPU: the C-ABI library builds/loads and exports every symbol include/mmsa.h declares; argument checks return status
codes without touching a GPU."""
import ctypes
import os
import re

from multimodal_sentiment_aanalysis_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mmsa.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmsa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert L.mmsa_abi_version() == 1


def test_argument_validation_without_gpu():
    L = _lib.load()
    assert L.mmsa_gemm(None, 1, None) == 1
    bad = _lib.BertCfg(batch=1, seq=16, hidden=100, layers=1, heads=1, intermediate=64, vocab=10, max_pos=16, type_vocab=2,
                       out_dim=256, dtype=0, ln_eps=1e-12)
    assert L.mmsa_bert_param_count(ctypes.byref(bad)) == -1
    assert L.mmsa_bert_ws_bytes(ctypes.byref(bad)) == 0
    assert L.mmsa_ce_fwd_bwd(None, None, None, None, None, 4, 3, 1.0, None) == 1


def test_layout_tables():
    L = _lib.load()
    c = _lib.BertCfg(batch=2, seq=16, hidden=768, layers=12, heads=12, intermediate=3072, vocab=30522, max_pos=512,
                     type_vocab=2, out_dim=256, dtype=1, ln_eps=1e-12)
    t = _lib.param_table(lambda: L.mmsa_bert_param_count(ctypes.byref(c)), lambda *a: L.mmsa_bert_param_info(ctypes.byref(c), *a))
    import math
    assert sum(math.prod(s) for n, _, s in t if n.startswith("bert.")) == 109482240  # BERT-base
    offs = [o for _, o, _ in t]
    assert offs == sorted(offs) and all(o % 64 == 0 for o in offs)
    # q | k | v weights are contiguous so the QKV projection is one GEMM
    q = next(x for x in t if x[0].endswith("layer.0.attention.self.query.weight"))
    k = next(x for x in t if x[0].endswith("layer.0.attention.self.key.weight"))
    assert k[1] == q[1] + 768 * 768
