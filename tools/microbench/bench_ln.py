import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from multimodal_sentiment_aanalysis_amd import kernels as K
dev = torch.device("cuda")
def bench(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
H = 768
g = torch.randn(H, device=dev); b = torch.randn(H, device=dev)
for M in [512, 2048, 8192, 32768]:
    x = torch.randn(M, H, device=dev).to(torch.bfloat16); dy = torch.randn(M, H, device=dev).to(torch.bfloat16)
    y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-12)
    tf = bench(lambda: K.layernorm_fwd(x, g, b, 1e-12))
    tb = bench(lambda: K.layernorm_bwd(dy, x, mean, rstd, g))
    print(f"M={M}: fwd {tf:.1f} us ({2*M*H*2/tf/1e6:.2f} TB/s)  bwd {tb:.1f} us ({3*M*H*2/tb/1e6:.2f} TB/s)")
