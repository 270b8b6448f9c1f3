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
print("conv (B,H,Cin,Cout,k,s)        fwd-gather  fwd-plain   dgrad-gather  dgrad-plain  wgrad-gather wgrad-plain")
for (Bn, H, Cin, Cout, k, s) in [(64, 56, 64, 64, 3, 1), (64, 28, 128, 128, 3, 1), (64, 14, 256, 256, 3, 1), (64, 7, 512, 512, 3, 1), (64, 56, 128, 128, 3, 2)]:
    p = k // 2; OH = (H + 2 * p - k) // s + 1
    x, w = rnd(Bn, H, H, Cin), rnd(Cout, k, k, Cin)
    M = Bn * OH * OH; Kd = k * k * Cin
    y = torch.empty(M, Cout, device=dev, dtype=torch.bfloat16)
    g = K.conv_geom(H, H, OH, OH, k, k, s, 1, -p, 1, Cin, Cin)
    t_fg = bench(lambda: K.gemm(x, w, y, M, Cout, Kd, Cin, Kd, Cout, gather=1, geom=g))
    xa = rnd(M, Kd)
    t_fp = bench(lambda: K.gemm(xa, w, y, M, Cout, Kd, Kd, Kd, Cout))
    dy = rnd(M, Cout); dx = torch.empty(Bn * H * H, Cin, device=dev, dtype=torch.bfloat16)
    g2 = K.conv_geom(OH, OH, H, H, k, k, 1, -1, p, s, Cout, Cout)
    t_dg = bench(lambda: K.gemm(dy, w, dx, Bn * H * H, Cin, k * k * Cout, Cout, Kd, Cin, b_kmajor=1, gather=1, geom=g2, b_tap_stride=Cin))
    dya = rnd(Bn * H * H, k * k * Cout); wk = rnd(k * k * Cout, Cin)
    t_dp = bench(lambda: K.gemm(dya, wk, dx, Bn * H * H, Cin, k * k * Cout, k * k * Cout, Cin, Cin, b_kmajor=1))
    dw = torch.zeros(Cout, Kd, device=dev)
    t_wg = bench(lambda: K.gemm(dy, x, dw, Cout, Kd, M, Cout, Cin, Kd, a_kmajor=1, b_kmajor=1, gather=2, geom=g, out_f32=1, split_k=64))
    xb = rnd(M, Kd)
    t_wp = bench(lambda: K.gemm(dy, xb, dw, Cout, Kd, M, Cout, Kd, Kd, a_kmajor=1, b_kmajor=1, out_f32=1, split_k=64))
    print(f"{(Bn,H,Cin,Cout,k,s)!s:28s} {t_fg:10.1f} {t_fp:10.1f} {t_dg:12.1f} {t_dp:12.1f} {t_wg:12.1f} {t_wp:11.1f}", flush=True)
