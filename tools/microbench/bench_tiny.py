"""fp32 batch-row GEMMs of the fusion head: one-launch MFMA kernel (gemm_f32_tiny.hip) vs the VALU kernel with its K split."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from multimodal_sentiment_aanalysis_amd import kernels as K, _lib
dev = torch.device("cuda")
def bench(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
g = torch.Generator(device="cpu").manual_seed(0)
def rnd(*s): return torch.randn(*s, generator=g).to(dev)
# (M, N, K, layout): NT = forward Linear (x[M,K], W[N,K]); NN = data gradient (dy[M,K] W[K,N]); TN = weight gradient
shapes = [(64, 256, 256, "NT"), (64, 256, 512, "NT"), (64, 256, 768, "NT"), (128, 768, 256, "NT"), (64, 128, 256, "NT"),
          (64, 3, 128, "NT"), (64, 64, 768, "NT"), (64, 256, 256, "NN"), (64, 768, 256, "NN"), (128, 256, 768, "NN"),
          (64, 128, 3, "NN"), (256, 256, 64, "TN"), (256, 768, 64, "TN"), (768, 256, 128, "TN"), (3, 128, 64, "TN")]
for M, N, Kd, lay in shapes:
    if lay == "NT":
        A, B = rnd(M, Kd), rnd(N, Kd); kw = dict(lda=Kd, ldb=Kd)
        ref = A.double() @ B.double().t()
    elif lay == "NN":
        A, B = rnd(M, Kd), rnd(Kd, N); kw = dict(lda=Kd, ldb=N, b_kmajor=1)
        ref = A.double() @ B.double()
    else:
        A, B = rnd(Kd, M), rnd(Kd, N); kw = dict(lda=M, ldb=N, a_kmajor=1, b_kmajor=1)
        ref = A.double().t() @ B.double()
    bias = rnd(N)
    C = torch.empty(M, N, device=dev)
    split = min(16, Kd // 64) if (Kd >= 256 and N % 4 == 0) else 1
    def run(impl, sp):
        return K.gemm(A, B, C, M, N, Kd, kw["lda"], kw["ldb"], N, a_kmajor=kw.get("a_kmajor", 0), b_kmajor=kw.get("b_kmajor", 0),
                      bias=bias, split_k=sp, impl=impl)
    run(K.GEMM_F32_SIMT, 1)
    err = ((C.double() - (ref + bias.double())).abs().max() / ref.abs().max()).item()
    t_tiny = bench(lambda: run(K.GEMM_F32_SIMT, 1))
    t_valu = bench(lambda: run(_lib.GEMM_F32_VALU, split))
    print(f"{lay} {M}x{N}x{Kd}: tiny {t_tiny:6.2f} us  valu(split {split}) {t_valu:6.2f} us   rel err {err:.2e}", flush=True)
