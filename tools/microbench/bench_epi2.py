import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from multimodal_sentiment_aanalysis_amd import kernels as K
from multimodal_sentiment_aanalysis_amd._lib import ACT_GELU
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
M, N, Kd = 8192, 3072, 768
A, B = rnd(M, Kd), rnd(N, Kd)
C, C2, mul, add = (torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(4))
mul.normal_(); add.normal_()
bias = torch.randn(N, device=dev)
Bk = rnd(Kd, N)
print("cfg     plain   bias  bias+gelu  bias+gelu+C2  NN-mul  NT-add")
for c in ["auto", "4", "3", "2", "2:2"]:
    if c == "auto": os.environ.pop("MMSA_G2_NJ", None)
    else: os.environ["MMSA_G2_NJ"] = c
    t = [bench(lambda: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N)),
         bench(lambda: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias)),
         bench(lambda: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias, act=ACT_GELU)),
         bench(lambda: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias, act=ACT_GELU, C2=C2, ldc2=N)),
         bench(lambda: K.gemm(A, Bk, C, M, N, Kd, Kd, N, N, b_kmajor=1, mul=mul, ldmul=N)),
         bench(lambda: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, add=add, ldadd=N))]
    print("%-6s" % c + "".join("%9.1f" % x for x in t), flush=True)
