"""Bare MFMA loop (mmsa_mfma_clock_probe) with one and with two workgroups per CU (= one / two waves per SIMD): does the matrix
pipe sustain its rate when two waves interleave on a SIMD, as the 8-wave workgroups of the persistent GEMM do?"""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from multimodal_sentiment_aanalysis_amd import _lib
L = _lib.load()
dev = torch.device("cuda:0")
cus = torch.cuda.get_device_properties(dev).multi_processor_count
for mult in (1, 2, 3, 4):
    blocks = cus * mult
    ws = torch.zeros(blocks * 1040, dtype=torch.uint8, device=dev)
    st_ptr = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    iters, launches = 4000, 40
    L.mmsa_mfma_clock_probe(ctypes.c_void_p(ws.data_ptr()), blocks, iters, 10, st_ptr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    L.mmsa_mfma_clock_probe(ctypes.c_void_p(ws.data_ptr()), blocks, iters, launches, st_ptr)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / launches
    st = ws[:blocks * 16].view(torch.int64).view(blocks, 2).double()
    ghz = (st[:, 0] / st[:, 1] * 0.1).median().item()
    tf_wall = blocks * 4 * iters * 16 * 16384 / dt / 1e12
    tf_stamp = blocks * 4 * iters * 16 * 16384 / (st[:, 1].max().item() * 1e-8) / 1e12
    print(f"{mult} workgroup(s) per CU: wall {tf_wall:7.1f} TFLOP/s, by the slowest workgroup's own clock {tf_stamp:7.1f} (if all ran concurrently), clock {ghz:.3f} GHz")
