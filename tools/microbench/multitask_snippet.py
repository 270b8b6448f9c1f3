import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import matplotlib
matplotlib.use("Agg")
from multimodal_sentiment_aanalysis_amd.MultimodalModel import MultimodalTransformerModel
from multimodal_sentiment_aanalysis_amd.dataLoader import MultimodalDataLoader, MultiTaskTrainer
os.chdir("/tmp")
model = MultimodalTransformerModel(multitask=True)
_, train, test = MultimodalDataLoader(file_path=None, batch_size=32, n=96).load_data(1)
t = MultiTaskTrainer(model, train, test, device="cuda", test_person=1)
t.run(1, 1, 1, 1, 1)
print("OK", {k: round(v[-1], 4) for k, v in t.metrics["test"].items()})
