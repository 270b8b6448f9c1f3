"""In-kernel stamps vs HIP events on stand-alone launches (split-K and plain): do the two clocks agree?"""
import sys, os, ctypes, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from multimodal_sentiment_aanalysis_amd import kernels as K, _lib
L = _lib.load()
dev = torch.device("cuda")
def rnd(*s): return torch.randn(*s, device=dev).to(torch.bfloat16)
for M, N, Kd, sp in [(3136, 512, 2048, 16), (3136, 512, 2048, 1), (3136, 512, 4608, 16), (8192, 768, 768, 1)]:
    A, B, C = rnd(M, Kd), rnd(N, Kd), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fn = lambda: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N, split_k=sp)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ev = e0.elapsed_time(e1) / 20 * 1e3
    L.mmsa_prof_mode(1); L.mmsa_prof_begin(64)
    for _ in range(20): fn()
    torch.cuda.synchronize()
    ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
    os.environ["MMSA_PROF_DUMP"] = "/tmp/stamp_dump.csv"
    L.mmsa_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
    L.mmsa_prof_mode(0)
    last = open("/tmp/stamp_dump.csv").read().strip().splitlines()[-1]
    print(f"{M}x{N}x{Kd} split<={sp}: events {ev:6.1f} us/launch; stamps {ms.value / n.value * 1e3:6.1f} us/launch over {n.value}; last row {last}", flush=True)
