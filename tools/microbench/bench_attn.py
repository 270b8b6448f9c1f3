"""BERT self-attention kernels at the text encoder's shape (B = 64, S = 128, 12 heads; --large: B = 32, S = 256, 16 heads): forward
and backward, default against MMSA_DISABLE=<names> in one process (the switch is read at every call), with the equality of results.

  python3 tools/microbench/bench_attn.py [names] [--large]      # names: comma list, default attn_fwd8
"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from multimodal_sentiment_aanalysis_amd import kernels as K


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    names = args[0] if args else "attn_fwd8"
    B, S, H = (32, 256, 16) if "--large" in sys.argv else (64, 128, 12)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B * S, 3 * H * 64, generator=g).to(dev, torch.bfloat16)
    dctx = torch.randn(B * S, H * 64, generator=g).to(dev, torch.bfloat16)
    mask = torch.ones(B, S, device=dev)
    mask[:, S - 7:] = 0
    res = {}
    for off in ("", names):
        os.environ["MMSA_DISABLE"] = off
        f = timed(lambda: K.attention_fwd(qkv, mask, B, S, H))
        b = timed(lambda: K.attention_bwd(qkv, mask, dctx, B, S, H))
        res[off] = (f, b, K.attention_fwd(qkv, mask, B, S, H), K.attention_bwd(qkv, mask, dctx, B, S, H))
    os.environ["MMSA_DISABLE"] = ""
    mb_f = (qkv.numel() + dctx.numel()) * 2 / 1e6
    mb_b = (2 * qkv.numel() + dctx.numel()) * 2 / 1e6
    for off, (f, b, _, _) in res.items():
        print(f"disable=[{off}] fwd {f:6.1f} us ({mb_f / f * 1e3:5.0f} GB/s)  bwd {b:6.1f} us ({mb_b / b * 1e3:5.0f} GB/s)")
    a, c = res[""], res[names]
    print("equal results: fwd", torch.equal(a[2], c[2]), " bwd", torch.equal(a[3], c[3]))


if __name__ == "__main__":
    main()
