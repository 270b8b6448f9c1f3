"""The BERT-base GEMM shapes of the step (bf16, plain store) back to back: us and TFLOP/s each. MMSA_LIB selects a variant build."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from multimodal_sentiment_aanalysis_amd import kernels as K
dev = torch.device("cuda")

def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def rnd(*s): return torch.randn(*s, device=dev).to(torch.bfloat16)
out = []
for M, N, Kd in [(8192, 3072, 768), (8192, 768, 3072), (8192, 2304, 768), (8192, 768, 768), (8192, 2048, 3072), (4096, 4096, 4096)]:
    A, B = rnd(M, Kd), rnd(N, Kd)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N))
    out.append(f"NT {M}x{N}x{Kd}: {t:6.1f} us {2.0 * M * N * Kd / t / 1e6:6.0f} TF")
    Bk = rnd(Kd, N)
    t = timeit(lambda: K.gemm(A, Bk, C, M, N, Kd, Kd, N, N, b_kmajor=1))
    out.append(f"NN {M}x{N}x{Kd}: {t:6.1f} us {2.0 * M * N * Kd / t / 1e6:6.0f} TF")
Ak, Bk = rnd(8192, 3072), rnd(8192, 768)
Cf = torch.zeros(3072, 768, device=dev, dtype=torch.float32)
t = timeit(lambda: K.gemm(Ak, Bk, Cf, 3072, 768, 8192, 3072, 768, 768, a_kmajor=1, b_kmajor=1, out_f32=1))
out.append(f"TN 3072x768x8192: {t:6.1f} us {2.0 * 3072 * 768 * 8192 / t / 1e6:6.0f} TF")
print(os.environ.get("MMSA_LIB", "default"), " | ".join(out))
