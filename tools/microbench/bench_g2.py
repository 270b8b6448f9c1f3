import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from multimodal_sentiment_aanalysis_amd import kernels as K
dev = torch.device("cuda")
def bench(name, fn, flop, bytes_, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"  {name:40s} {ms*1e3:8.1f} us {flop/ms/1e9:8.1f} TF/s {bytes_/ms/1e9:7.2f} TB/s", flush=True)
def rnd(*s): return torch.randn(*s, device=dev).to(torch.bfloat16)
print("mode", "V1" if os.environ.get("MMSA_GEMM_V1") else "V2", "NJ", os.environ.get("MMSA_G2_NJ"))
nt = [(8192, 2304, 768), (8192, 768, 768), (8192, 3072, 768), (8192, 768, 3072), (4096, 4096, 4096),
      (200704, 64, 64), (200704, 256, 64), (200704, 64, 256), (200704, 128, 256), (50176, 512, 128), (50176, 128, 512),
      (12544, 1024, 256), (12544, 256, 1024), (3136, 2048, 512), (3136, 512, 2048), (802816, 64, 192)]
for M, N, Kd in nt:
    A, B, C = rnd(M, Kd), rnd(N, Kd), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    by = 2 * (M * Kd + N * Kd + M * N)
    bench(f"NT {M}x{N}x{Kd}", lambda: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N), 2*M*N*Kd, by)
    Bk = rnd(Kd, N)
    bench(f"NN {M}x{N}x{Kd}", lambda: K.gemm(A, Bk, C, M, N, Kd, Kd, N, N, b_kmajor=1), 2*M*N*Kd, by)
for No, Ki, Mr, split in [(768, 768, 8192, 16), (3072, 768, 8192, 16), (768, 3072, 8192, 16), (2304, 768, 8192, 16),
                          (64, 64, 200704, 64), (256, 64, 200704, 64), (64, 256, 200704, 64), (128, 512, 50176, 64),
                          (512, 128, 50176, 64), (1024, 256, 12544, 32), (256, 1024, 12544, 32), (2048, 512, 3136, 8), (64, 192, 802816, 64)]:
    A, B = rnd(Mr, No), rnd(Mr, Ki)
    C = torch.zeros(No, Ki, device=dev)
    by = 2 * (Mr * No + Mr * Ki) + 4 * No * Ki
    bench(f"TN wgrad {No}x{Ki} K={Mr} split<={split}", lambda: K.gemm(A, B, C, No, Ki, Mr, No, Ki, Ki, a_kmajor=1, b_kmajor=1, out_f32=1, split_k=split), 2*No*Ki*Mr, by)
