"""The 256 x 256 tile of the persistent GEMM (gemm_mfma2.hip NJ = 8) against the planner's default tile, same process, interleaved:
correctness on a float64 reference of sampled rows, then time. MMSA_G2_NJ is read per launch, so the tile is switched in place."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from multimodal_sentiment_aanalysis_amd import kernels as K
dev = torch.device("cuda")

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def rnd(*s): return torch.randn(*s, device=dev).to(torch.bfloat16)

def check(C, ref_fn, M, what):
    rows = torch.randint(0, M, (64,), device=dev)
    ref = ref_fn(rows).double()
    got = C[rows].double()
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    print(f"    {what}: max rel err on 64 sampled rows {err:.2e}", "OK" if err < 2e-2 else "**** WRONG ****", flush=True)

shapes = [(4096, 4096, 4096), (8192, 2048, 768), (8192, 3072, 768), (8192, 2304, 768), (8192, 4096, 1024), (8192, 1024, 4096),
          (8192, 3072, 1024), (16384, 3072, 768), (8192, 768, 3072)]
for M, N, Kd in shapes:
    fl = 2.0 * M * N * Kd
    A, B, Bk = rnd(M, Kd), rnd(N, Kd), rnd(Kd, N)
    for mode in ("NT", "NN"):
        res = {}
        for tile in ("default", "4:8"):
            if tile == "default": os.environ.pop("MMSA_G2_NJ", None)
            else: os.environ["MMSA_G2_NJ"] = tile
            C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
            if mode == "NT":
                fn = lambda: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N)
                ref = lambda rows: A[rows].double() @ B.double().t()
            else:
                fn = lambda: K.gemm(A, Bk, C, M, N, Kd, Kd, N, N, b_kmajor=1)
                ref = lambda rows: A[rows].double() @ Bk.double()
            fn(); torch.cuda.synchronize()
            if tile != "default": check(C, ref, M, f"{mode} {M}x{N}x{Kd} tile {tile}")
            res[tile] = timeit(fn)
        print(f"{mode} {M}x{N}x{Kd}: default {res['default']:7.1f} us ({fl/res['default']/1e6:5.0f} TF)   256x256 {res['4:8']:7.1f} us ({fl/res['4:8']/1e6:5.0f} TF)", flush=True)
os.environ.pop("MMSA_G2_NJ", None)
# TN (weight gradients): C[No, Ki] = A[Mr, No]^T B[Mr, Ki], fp32 output
for No, Ki, Mr in [(3072, 768, 8192), (768, 3072, 8192), (2304, 768, 8192), (4096, 1024, 8192), (4096, 4096, 4096)]:
    fl = 2.0 * No * Ki * Mr
    A, B = rnd(Mr, No), rnd(Mr, Ki)
    res = {}
    for tile in ("default", "4:8"):
        if tile == "default": os.environ.pop("MMSA_G2_NJ", None)
        else: os.environ["MMSA_G2_NJ"] = tile
        C = torch.zeros(No, Ki, device=dev)
        fn = lambda: K.gemm(A, B, C, No, Ki, Mr, No, Ki, Ki, a_kmajor=1, b_kmajor=1, out_f32=1, split_k=1)
        fn(); torch.cuda.synchronize()
        if tile != "default":
            rows = torch.randint(0, No, (64,), device=dev)
            ref = A[:, rows].double().t() @ B.double()
            err = ((C[rows].double() - ref).abs().max() / ref.abs().max()).item()
            print(f"    TN {No}x{Ki} K={Mr} tile {tile}: max rel err {err:.2e}", "OK" if err < 1e-3 else "**** WRONG ****", flush=True)
        res[tile] = timeit(fn)
    print(f"TN {No}x{Ki} K={Mr}: default {res['default']:7.1f} us ({fl/res['default']/1e6:5.0f} TF)   256x256 {res['4:8']:7.1f} us ({fl/res['4:8']/1e6:5.0f} TF)", flush=True)
os.environ.pop("MMSA_G2_NJ", None)
