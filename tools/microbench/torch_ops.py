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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step.step(*batch)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60))
