"""Feasibility probe: capture one FusedTrainStep.step into a torch.cuda.CUDAGraph (hipGraph) and replay it. dropout = 0 (the fusion
head's dropout seed is a per-call host value), constant learning rate. Run it with MMSA_TWO_STREAMS=0 MMSA_WGRAD_STREAM=0: the
step refuses a capture with its side streams on (fused.py). PROBE_FORCE=1 bypasses that refusal for ONE diagnostic run of the
three-stream capture (round 3's crash; keep AMD_LOG_LEVEL=3 output and the faulthandler trace under gpurun_out/)."""
import faulthandler, os, sys, time, torch
faulthandler.enable()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
import multimodal_sentiment_aanalysis_amd as mm
from multimodal_sentiment_aanalysis_amd.fused import FusedTrainStep
if os.environ.get("PROBE_FORCE") == "1":
    torch.cuda.is_current_stream_capturing = lambda: False  # diagnostic only: lets the guarded step run under capture
dev = torch.device("cuda:0")
B = int(os.environ.get("PROBE_B", "64"))
batch = bench.synth_batch(B, 128, 30522, dev, 1234)

def make():
    torch.manual_seed(0)
    model = mm.MultimodalTransformerModel(dropout=0.0)
    return FusedTrainStep(model, dev, precision="bf16")

ref = make()
for _ in range(8): ref.step(*batch)
torch.cuda.synchronize()
w_ref = ref.state.flat_w.clone()
del ref

st = make()
for _ in range(3): st.step(*batch)   # warm-up outside the capture (plans, attributes, workspaces)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss, logits = st.step(*batch)   # step 4 is captured (not executed)
for _ in range(5): g.replay()        # steps 4..8
torch.cuda.synchronize()
print("captured; weights equal to 8 eager steps:", torch.equal(st.state.flat_w, w_ref), "max diff", (st.state.flat_w - w_ref).abs().max().item())
for rep in range(3):
    n = 50
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"graph replay: host {1e3 * (t1 - t0) / n:.2f} ms/step, GPU {1e3 * (t2 - t0) / n:.3f} ms/step")
