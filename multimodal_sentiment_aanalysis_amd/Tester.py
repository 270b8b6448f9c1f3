"""Tester with the reference's API (MML_ZYC/Tester.py): load_model (strips a `module.` prefix, Tester.py:29-35),
evaluate -> dict{loss, accuracy, predictions, labels, probabilities} (:37-84), predict_single (:112-127), run (:129-133).
Softmax probabilities and the CE come from the fused HIP kernel (mmsa_ce_fwd_bwd); report / plot helpers are optional
(sklearn / seaborn are imported lazily, the reference fails at import time without seaborn)."""
import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .Trainer import unpack


class Tester:
    def __init__(self, model, test_loader, device="cuda"):
        self.model = model.to(device)
        self.test_loader, self.device = test_loader, device
        self.loss, self.accuracy = 0.0, 0.0
        self.all_preds, self.all_labels, self.all_probs = [], [], []

    def load_model(self, model_path):
        state_dict = torch.load(model_path, map_location=self.device, weights_only=True)
        if all(k.startswith("module.") for k in state_dict.keys()):
            state_dict = {k[7:]: v for k, v in state_dict.items()}
        self.model.load_state_dict(state_dict)
        print(f"Loaded model weights from {model_path}")

    def _ce_probs(self, outputs, labels):
        L = _lib.load()
        B, C = outputs.shape
        loss = torch.empty((), dtype=torch.float32, device=outputs.device)
        probs = torch.empty_like(outputs)
        check(L.mmsa_ce_fwd_bwd(ptr(outputs.contiguous()), ptr(labels.long().contiguous()), ptr(loss), None, ptr(probs), B, C,
                                1.0, stream_ptr()), "mmsa_ce_fwd_bwd")
        return loss, probs

    def evaluate(self, verbose=True):
        """Tester.py:37-84. Per-batch results stay on the device; one host copy at the end (the reference syncs 3x / batch)."""
        self.model.eval()
        loss_sum = torch.zeros((), dtype=torch.float64, device=self.device)
        hits = torch.zeros((), dtype=torch.int64, device=self.device)
        seen, pred_chunks, label_chunks, prob_chunks = 0, [], [], []
        with torch.no_grad():
            for batch_inputs, batch_labels in self.test_loader:
                feeds = unpack(batch_inputs, self.device)
                y = batch_labels.to(self.device)
                logits = self.model(*feeds)  # no labels -> bare logits (Tester.py:53)
                ce, probs = self._ce_probs(logits, y)
                n = y.shape[0]
                winners = logits.argmax(dim=1)
                loss_sum += ce.double() * n
                hits += (winners == y).sum()
                seen += n
                pred_chunks.append(winners)
                label_chunks.append(y)
                prob_chunks.append(probs)
        self.loss = float(loss_sum.item()) / seen  # an empty loader raises ZeroDivisionError, as Tester.py:71 does
        self.accuracy = int(hits.item()) / seen
        self.all_preds.extend(torch.cat(pred_chunks).cpu().numpy())
        self.all_labels.extend(torch.cat(label_chunks).cpu().numpy())
        self.all_probs.extend(torch.cat(prob_chunks).cpu().numpy())
        if verbose:
            self._print_metrics()
        return dict(loss=self.loss, accuracy=self.accuracy, predictions=np.array(self.all_preds),
                    labels=np.array(self.all_labels), probabilities=np.array(self.all_probs))

    def _print_metrics(self):
        print(f"\n{'=' * 40}\nEvaluation Results:\n- Average Loss: {self.loss:.4f}\n- Accuracy: {self.accuracy:.2%}")
        try:
            from sklearn.metrics import classification_report
            print(classification_report(self.all_labels, self.all_preds, zero_division=0))
        except Exception:
            pass
        print("=" * 40)

    def predict_single(self, data_dict):
        """Tester.py:112-127: one sample (no batch dimension) -> class index + softmax vector."""
        self.model.eval()
        with torch.no_grad():
            feeds = [t.unsqueeze(0) for t in unpack(data_dict, self.device)]
            logits = self.model(*feeds)
            dummy = torch.zeros(1, dtype=torch.long, device=logits.device)
            probs = self._ce_probs(logits, dummy)[1]
        return {"prediction": int(logits.argmax(dim=1).item()), "probabilities": probs.squeeze().cpu().numpy()}

    def run(self, model_path=None):
        if model_path is not None:
            self.load_model(model_path)
        return self.evaluate()
