"""BatchNorm kernels at the image encoder's shapes (B = 64): forward (statistics pass + apply) and backward (partial sums +
finalize + apply) per shape, default against MMSA_DISABLE=<names> in one process (the switch is read at every call).

  python3 tools/microbench/bench_bn.py [names]      # names: comma list for MMSA_DISABLE (default: none, both columns are the shipped kernels)
"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from multimodal_sentiment_aanalysis_amd import kernels as K

SHAPES = [(802816, 64), (200704, 64), (200704, 256), (200704, 128), (50176, 128), (50176, 512), (50176, 256), (12544, 256),
          (12544, 1024), (12544, 512), (3136, 512), (3136, 2048)]
COUNT = {(802816, 64): 1, (200704, 64): 6, (200704, 256): 4, (200704, 128): 1, (50176, 128): 7, (50176, 512): 5, (50176, 256): 1,
         (12544, 256): 11, (12544, 1024): 7, (12544, 512): 1, (3136, 512): 5, (3136, 2048): 4}


def timed(fn, iters=20):
    for _ in range(3):
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
    names = sys.argv[1] if len(sys.argv) > 1 else ""
    dev = torch.device("cuda:0")
    tot = {"fwd": [0.0, 0.0], "bwd": [0.0, 0.0]}
    print(f"{'M':>8} {'C':>5} {'n':>3} | fwd us (default / {names} off) | bwd us | bwd GB/s default")
    for M, C in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(1)
        x = torch.randn(M, C, generator=g).to(dev, torch.bfloat16)
        dy = torch.randn(M, C, generator=g).to(dev, torch.bfloat16)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        y, mean, invstd = K.bn_fwd(x, gamma, beta, rm, rv, None, K.ACT_RELU, True)
        row = []
        for off in ("", names):
            os.environ["MMSA_DISABLE"] = off
            f = timed(lambda: K.bn_fwd(x, gamma, beta, rm, rv, None, K.ACT_RELU, True))
            b = timed(lambda: K.bn_bwd(dy, x, y, mean, invstd, gamma, beta, K.ACT_RELU, True))
            row.append((f, b))
        os.environ["MMSA_DISABLE"] = ""
        n = COUNT[(M, C)]
        for i in range(2):
            tot["fwd"][i] += n * row[i][0]
            tot["bwd"][i] += n * row[i][1]
        gbs = (M * C * 2 * (3 + 2 + 1)) / row[0][1] / 1e3  # partial: dy, x, y; apply: dy, x, y -> dx
        print(f"{M:>8} {C:>5} {n:>3} | {row[0][0]:8.1f} {row[1][0]:8.1f} | {row[0][1]:8.1f} {row[1][1]:8.1f} | {gbs:7.0f}")
    print(f"weighted by layer count: fwd {tot['fwd'][0] / 1e3:.3f} / {tot['fwd'][1] / 1e3:.3f} ms, bwd {tot['bwd'][0] / 1e3:.3f} / {tot['bwd'][1] / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
