"""Which torch-level ops (not library kernels) run inside a fused train step: torch.profiler table."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
import multimodal_sentiment_aanalysis_amd as mm
from multimodal_sentiment_aanalysis_amd.fused import FusedTrainStep
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = mm.MultimodalTransformerModel()
step = FusedTrainStep(model, dev, precision="bf16")
batch = bench.synth_batch(64, 128, 30522, dev, 1234)
for _ in range(3): step.step(*batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step.step(*batch)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60))
rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.key.startswith("aten::") or "Memcpy" in e.key or "Memset" in e.key or "elementwise" in e.key or "copyBuffer" in e.key or "fillBuffer" in e.key]
print("\naten-level ops in one step (name, calls, self device us):")
for k, c, t in sorted(rows, key=lambda r: -r[1]):
    print(f"  {k[:70]:70s} {c:4d} {t:9.1f}")

print("\naten ops with their Python call sites (calls, op <- innermost repo frames):")
sites = {}
for e in prof.events():
    if not e.name.startswith("aten::") or e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    st = [f for f in (e.stack or []) if "multimodal_sentiment" in f or "bench.py" in f][:2]
    k = (e.name, " <- ".join(f.split("/")[-1] for f in st))
    sites[k] = sites.get(k, 0) + 1
for (n, st), c in sorted(sites.items(), key=lambda r: -r[1]):
    print(f"  {c:4d} {n:28s} {st}")
