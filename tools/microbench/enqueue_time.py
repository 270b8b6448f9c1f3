"""How long does the HOST take to enqueue one train step (no synchronisation inside the loop)? If it is close to the step time the
step is launch-bound on the host and a hipGraph of the step would pay; if it is well below, the GPU is the bottleneck."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
import multimodal_sentiment_aanalysis_amd as mm
from multimodal_sentiment_aanalysis_amd.fused import FusedTrainStep
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = mm.MultimodalTransformerModel()
step = FusedTrainStep(model, dev, precision="bf16")
batch = bench.synth_batch(64, 128, 30522, dev, 1234)
for _ in range(10): step.step(*batch)
torch.cuda.synchronize()
for rep in range(3):
    n = 40
    t0 = time.perf_counter()
    for _ in range(n): step.step(*batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step; until the GPU is done {1e3 * (t2 - t0) / n:.2f} ms/step; GPU backlog at the end of the loop {1e3 * (t2 - t1):.1f} ms")
# the loop above is throttled by back-pressure once the HIP queue is full (the host runs ~3 steps ahead and then waits for the
# GPU): the UNTHROTTLED cost of enqueueing one step is measured from an empty queue
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step.step(*batch)
    ts.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
print("host enqueue of one step from an empty queue:", " ".join(f"{t:.2f}" for t in ts), "ms")
