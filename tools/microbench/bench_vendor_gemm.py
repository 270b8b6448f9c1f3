"""Vendor-library datapoint for DESIGN §3: torch.matmul (hipBLASLt / rocBLAS) on the BERT-base GEMM shapes, against mmsa_gemm."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from multimodal_sentiment_aanalysis_amd import kernels as K
dev = torch.device("cuda")
def bench(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, N, Kd in [(8192, 3072, 768), (8192, 768, 3072), (8192, 2304, 768), (8192, 768, 768), (4096, 4096, 4096), (16384, 3072, 768)]:
    A = torch.randn(M, Kd, device=dev).to(torch.bfloat16); W = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tv = bench(lambda: torch.matmul(A, W.t(), out=C))
    tm = bench(lambda: K.gemm(A, W, C, M, N, Kd, Kd, Kd, N))
    fl = 2.0 * M * N * Kd
    print(f"{M}x{N}x{Kd}: vendor {tv:7.1f} us ({fl/tv/1e6:6.0f} TF)   mmsa {tm:7.1f} us ({fl/tm/1e6:6.0f} TF)", flush=True)
