"""Timing-only ablations of the 256 x 256 loop of the persistent GEMM (results are wrong by design). MMSA_GEMM_DBG_NOLOAD=1: the
staging descriptors get zero records (every LDS-DMA load is dropped by the range check; the instruction stream stays).
MMSA_G2_DBG bits: 1 no LDS-DMA instructions, 2 no fragment reads, 4 no barrier, 8 no epilogue."""
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
for M, N, Kd in [(8192, 2048, 3072), (8192, 2048, 768)]:
    A, B = rnd(M, Kd), rnd(N, Kd)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * Kd
    fn = lambda: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N)
    for tile in ("4:8", "4:4"):
        os.environ["MMSA_G2_NJ"] = tile
        for noload in ("0", "1"):
            os.environ["MMSA_GEMM_DBG_NOLOAD"] = noload
            out = []
            for dbg in (0, 8, 1, 2, 4, 3, 6, 7, 15):
                os.environ["MMSA_G2_DBG"] = str(dbg)
                t = timeit(fn)
                out.append(f"dbg{dbg}: {t:6.1f}")
            print(f"NT {M}x{N}x{Kd} tile {tile} noload={noload}:  " + "  ".join(out), flush=True)
os.environ["MMSA_GEMM_DBG_NOLOAD"] = "0"; os.environ["MMSA_G2_DBG"] = "0"
