"""Tile-shape / K-split sweep for the low-occupancy and store-bound GEMM shapes of the ResNet-50 step."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from multimodal_sentiment_aanalysis_amd import kernels as K
dev = torch.device("cuda")
def rnd(*s): return torch.randn(*s, device=dev).to(torch.bfloat16)
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def setenv(k, v):
    if v is None: os.environ.pop(k, None)
    else: os.environ[k] = str(v)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "occ"):
    print("low-occupancy shapes: us per (tile cfg x split)")
    for M, N, Kd in [(3136, 512, 2048), (3136, 512, 4608), (3136, 2048, 512), (3136, 2048, 1024), (12544, 256, 2304), (12544, 256, 1024)]:
        A, B, Bk, C = rnd(M, Kd), rnd(N, Kd), rnd(Kd, N), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for lay in ("NT", "NN"):
            fn = (lambda: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, split_k=16)) if lay == "NT" else \
                 (lambda: K.gemm(A, Bk, C, M, N, Kd, Kd, N, N, b_kmajor=1, split_k=16))
            setenv("MMSA_G2_NJ", None); setenv("MMSA_G2_SPLIT", None)
            print(f"{lay} {M}x{N}x{Kd}: auto {bench(fn):7.1f}", flush=True)
            for cfg in ["4", "3", "2", "2:2", "2:1"]:
                setenv("MMSA_G2_NJ", cfg)
                row = []
                for sp in [1, 2, 3, 4, 6, 8]:
                    setenv("MMSA_G2_SPLIT", sp)
                    row.append(bench(fn))
                print(f"   cfg {cfg:4s} split 1,2,3,4,6,8: " + " ".join(f"{x:7.1f}" for x in row), flush=True)
    setenv("MMSA_G2_NJ", None); setenv("MMSA_G2_SPLIT", None)
if which in ("all", "add"):
    print("store-bound data-gradient shapes (NN): us plain / +add per tile cfg")
    for M, N, Kd in [(200704, 256, 64), (200704, 256, 128), (50176, 512, 128), (50176, 512, 256), (12544, 1024, 256), (12544, 1024, 512), (3136, 2048, 512)]:
        A, Bk = rnd(M, Kd), rnd(Kd, N)
        C, add = torch.empty(M, N, device=dev, dtype=torch.bfloat16), rnd(M, N)
        for cfg in [None, "4", "3", "2", "2:2", "2:1"]:
            setenv("MMSA_G2_NJ", cfg)
            t0 = bench(lambda: K.gemm(A, Bk, C, M, N, Kd, Kd, N, N, b_kmajor=1))
            t1 = bench(lambda: K.gemm(A, Bk, C, M, N, Kd, Kd, N, N, b_kmajor=1, add=add, ldadd=N))
            by0 = 2 * (M * Kd + M * N); by1 = by0 + 2 * M * N
            print(f"NN {M}x{N}x{Kd} cfg {str(cfg):5s} plain {t0:7.1f} us ({by0/t0/1e6:5.2f} TB/s)  +add {t1:7.1f} us ({by1/t1/1e6:5.2f} TB/s)", flush=True)
    setenv("MMSA_G2_NJ", None)
