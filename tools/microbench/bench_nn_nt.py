"""Data-gradient layout question: dX = dY W as NN (W [N][K] read k-major, transposing LDS reads) vs NT against a transposed copy."""
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
# dgrad of Linear(in=Kin, out=Nout): dX[M, Kin] = dY[M, Nout] @ W[Nout, Kin]  -> GEMM M x Kin x Nout
for M, Kin, Nout in [(8192, 768, 2304), (8192, 768, 768), (8192, 768, 3072), (8192, 3072, 768)]:
    dY = torch.randn(M, Nout, device=dev).to(torch.bfloat16)
    W = torch.randn(Nout, Kin, device=dev).to(torch.bfloat16)
    Wt = W.t().contiguous()
    dX = torch.empty(M, Kin, device=dev, dtype=torch.bfloat16)
    t_nn = bench(lambda: K.gemm(dY, W, dX, M, Kin, Nout, Nout, Kin, Kin, b_kmajor=1))
    t_nt = bench(lambda: K.gemm(dY, Wt, dX, M, Kin, Nout, Nout, Nout, Kin))
    fl = 2.0 * M * Kin * Nout
    print(f"dgrad M={M} in={Kin} out={Nout}: NN {t_nn:6.1f} us ({fl/t_nn/1e6:5.0f} TF)   NT(transposed copy) {t_nt:6.1f} us ({fl/t_nt/1e6:5.0f} TF)", flush=True)
