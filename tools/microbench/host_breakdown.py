"""Host time of one train step by C entry point (wall time inside each ctypes call, summed per step) and the Python remainder."""
import collections, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
import multimodal_sentiment_aanalysis_amd as mm
from multimodal_sentiment_aanalysis_amd import _lib
from multimodal_sentiment_aanalysis_amd.fused import FusedTrainStep
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = mm.MultimodalTransformerModel()
step = FusedTrainStep(model, dev, precision="bf16")
batch = bench.synth_batch(64, 128, 30522, dev, 1234)
for _ in range(5): step.step(*batch)
torch.cuda.synchronize()
L = _lib.load()
acc = collections.Counter(); cnt = collections.Counter()
def wrap(name, fn):
    def f(*a):
        t0 = time.perf_counter()
        r = fn(*a)
        acc[name] += time.perf_counter() - t0; cnt[name] += 1
        return r
    return f
for name in [n for n in dir(L) if n.startswith("mmsa_")]:
    try:
        setattr(L, name, wrap(name, getattr(L, name)))
    except Exception:
        pass
n = 20
t0 = time.perf_counter()
for _ in range(n): step.step(*batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
tot = (t1 - t0) / n * 1e3
print(f"host enqueue {tot:.2f} ms/step")
inside = 0.0
for k, v in acc.most_common(12):
    print(f"  {k:34s} {1e3 * v / n:7.3f} ms/step  x{cnt[k] / n:5.1f}")
    inside += 1e3 * v / n
print(f"  all C calls {1e3 * sum(acc.values()) / n:.2f} ms/step; Python + torch remainder {tot - 1e3 * sum(acc.values()) / n:.2f} ms/step")
