"""Ring-depth experiment: BERT-shaped NT GEMMs on the 256 x 64 tile (forced MMSA_G2_NJ=2), 3 stages vs 4 (variant build)."""
import sys, os, torch
os.environ["MMSA_G2_NJ"] = "2"
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
for M, N, Kd in [(8192, 768, 768), (8192, 3072, 768), (8192, 768, 3072), (16384, 768, 768), (4096, 4096, 4096)]:
    A = torch.randn(M, Kd, device=dev).to(torch.bfloat16); W = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: K.gemm(A, W, C, M, N, Kd, Kd, Kd, N))
    ref = (A[:64].float() @ W.float().t())
    err = ((C[:64].float() - ref).abs().max() / ref.abs().max()).item()
    print(f"{M}x{N}x{Kd} tile 256x64: {t:7.1f} us ({2.0*M*N*Kd/t/1e6:6.0f} TF)  err {err:.1e}", flush=True)
