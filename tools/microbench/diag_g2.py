import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from multimodal_sentiment_aanalysis_amd import kernels as K
dev = torch.device("cuda")
def rnd(*s): return torch.randn(*s, device=dev).to(torch.bfloat16)
iters = int(os.environ.get("ITERS", "20"))
def bench(name, fn, flop):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"  {name:40s} {ms*1e3:8.1f} us {flop/ms/1e9:8.1f} TF/s", flush=True)
print("NOLOAD", os.environ.get("MMSA_GEMM_DBG_NOLOAD"), "NJ", os.environ.get("MMSA_G2_NJ"), "V1", os.environ.get("MMSA_GEMM_V1"))
for M, N, Kd in [(8192, 3072, 768), (8192, 768, 3072), (4096, 4096, 4096), (8192, 2304, 768)]:
    A, B, C = rnd(M, Kd), rnd(N, Kd), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bench(f"NT {M}x{N}x{Kd}", lambda: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N), 2*M*N*Kd)
    Bk = rnd(Kd, N)
    bench(f"NN {M}x{N}x{Kd}", lambda: K.gemm(A, Bk, C, M, N, Kd, Kd, N, N, b_kmajor=1), 2*M*N*Kd)
