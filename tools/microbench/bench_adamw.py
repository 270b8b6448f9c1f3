import sys, os, ctypes, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from multimodal_sentiment_aanalysis_amd import _lib
from multimodal_sentiment_aanalysis_amd._lib import ptr, stream_ptr
L = _lib.load()
dev = torch.device("cuda")
n = 135_600_000
w = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 0.01
m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
w16 = torch.empty(n, device=dev, dtype=torch.bfloat16)
norm = torch.tensor([1.0, 1.0, 0.1, 0.0316], device=dev)
steps = torch.ones(1, dtype=torch.int32, device=dev)
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
host = lambda w16_: L.mmsa_adamw_step(ptr(w), ptr(g), ptr(m), ptr(v), ptr(w16_), n, 1e-4, 0.9, 0.999, 1e-8, 0.01, 1, ptr(norm), 1.0, stream_ptr())
devs = lambda w16_: L.mmsa_adamw_step_dev(ptr(w), ptr(g), ptr(m), ptr(v), ptr(w16_), n, 1e-4, 0.9, 0.999, 1e-8, 0.01, ptr(steps), ptr(norm), 1.0, stream_ptr())
for nm, fn in (("host-step bf16copy", lambda: host(w16)), ("dev-step bf16copy", lambda: devs(w16)), ("host-step no copy", lambda: host(None)), ("dev-step no copy", lambda: devs(None))):
    print(f"{nm:24s} {t(fn):8.1f} us")
