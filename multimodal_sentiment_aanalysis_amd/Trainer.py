"""Trainer with the reference's API and step sequence (MML_ZYC/Trainer.py), for (image, text) batches.

Same constructor, attributes and methods (`train_epoch`, `test`, `early_stop`, `plot_progress`,
`test_with_loaded_model`, `run`), same per-batch sequence (Trainer.py:51-89): to(device) -> zero_grad -> forward ->
NaN guard -> CE (+ learnable weight * aux loss) -> NaN-skip -> backward -> clip_grad_norm_(1.0) -> AdamW(lr 1e-4,
weight_decay 0.01) step -> running metrics; ReduceLROnPlateau(patience 3, factor 0.5); early stop patience 5.

`fused=True` (default on GPU) runs the step body through FusedTrainStep (HIP CE / clip / AdamW kernels on the flat
buffers, data-parallel all-reduce when torch.distributed is initialised); `fused=False` is literally the reference
sequence on torch.optim.AdamW, which also works because the model is an ordinary nn.Module.
Batches are `(data_dict, labels)`; the dict is read positionally (`eeg/eye/pps` in the reference, Trainer.py:53-55;
`image/text/mask` here), so either key set works.
"""
import torch
import torch.nn as nn
import torch.optim as optim
from torch.optim.lr_scheduler import ReduceLROnPlateau

from .engine import CrossEntropyLoss
from .fused import FusedTrainStep

_KEYSETS = (("image", "text", "mask"), ("image", "token_ids", "attention_mask"), ("eeg", "eye", "pps"))


def unpack(data_dict, device):
    for ks in _KEYSETS:
        if all(k in data_dict for k in ks):
            a, b, c = (data_dict[k] for k in ks)
            break
    else:
        a, b, c = list(data_dict.values())[:3]
    # Trainer.py:53-55 applies .float() to the first two inputs only
    return a.to(device).float(), b.to(device).float(), c.to(device)


class Trainer:
    def __init__(self, model, train_loader, test_loader, device="cuda", fused=None, precision="bf16"):
        self.device = device
        self.model = model.to(device)
        self.train_loader, self.test_loader = train_loader, test_loader
        self.fused = (torch.device(device).type == "cuda") if fused is None else fused
        if self.fused:
            self.fused_step = FusedTrainStep(self.model, device, precision=precision, lr=0.0001, weight_decay=0.01,
                                             max_norm=1.0)
            self.criterion = CrossEntropyLoss()
            self.optimizer = None
            self.scheduler = None
        else:
            self.criterion = nn.CrossEntropyLoss()
            self.optimizer = optim.AdamW(model.parameters(), lr=0.0001, weight_decay=0.01)
            self.contrastive_weight = nn.Parameter(torch.ones(1, device=device))
            self.optimizer.add_param_group({"params": [self.contrastive_weight]})
            self.scheduler = ReduceLROnPlateau(self.optimizer, "min", patience=3, factor=0.5)
        self.train_loss, self.test_loss, self.train_acc, self.test_acc = [], [], [], []
        self.best_val_loss = float("inf")
        self.patience, self.counter, self.early_stop_flag = 5, 0, False
        self._plateau_bad, self._plateau_best = 0, float("inf")

    # ---- one epoch (Trainer.py:42-105). Running sums live on the device: one host read per epoch instead of the
    # reference's three to four .item() syncs per batch (only the un-fused branch needs a per-batch NaN decision).
    def _tally(self):
        z = lambda dt: torch.zeros((), dtype=dt, device=self.device)  # noqa: E731
        return {"loss": z(torch.float64), "ce": z(torch.float64), "aux": z(torch.float64), "hit": z(torch.int64),
                "seen": z(torch.int64), "nan": z(torch.int64), "batches": z(torch.int64)}

    @staticmethod
    def _count(tally, logits, y, loss, ce, aux):
        ok = ~torch.isnan(loss.detach())
        tally["nan"] += (~ok).long()
        tally["batches"] += ok.long()
        tally["loss"] += torch.where(ok, loss.detach().double(), torch.zeros_like(tally["loss"]))
        tally["ce"] += torch.where(ok, ce.detach().double(), torch.zeros_like(tally["ce"]))
        tally["aux"] += torch.where(ok, aux.detach().double().reshape(()), torch.zeros_like(tally["aux"]))
        tally["hit"] += torch.where(ok, (logits.detach().argmax(dim=1) == y).sum(), torch.zeros_like(tally["hit"]))
        tally["seen"] += torch.where(ok, torch.full_like(tally["seen"], y.shape[0]), torch.zeros_like(tally["seen"]))

    def train_epoch(self, epoch):
        self.model.train()
        tally = self._tally()
        zero = torch.zeros((), device=self.device)
        for batch_inputs, batch_labels in self.train_loader:
            feeds = unpack(batch_inputs, self.device)
            y = batch_labels.to(self.device)
            if self.fused:
                step_loss, logits = self.fused_step.step(*feeds, y)
                self._count(tally, logits, y, step_loss, step_loss, zero)
                continue
            self.optimizer.zero_grad()
            logits, aux = self.model(*feeds, y)
            if torch.isnan(logits).any():
                print("Warning: Model output contains NaN!")
                logits = torch.nan_to_num(logits)
            ce = self.criterion(logits, y)
            objective = ce + self.contrastive_weight * aux
            if torch.isnan(objective):  # Trainer.py:74-76: the batch is skipped, no update
                print("NaN loss detected, skipping batch")
                tally["nan"] += 1
                continue
            objective.backward()
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), 1.0)
            self.optimizer.step()
            self._count(tally, logits, y, objective.reshape(()), ce, aux)
        t = {k: v.item() for k, v in tally.items()}
        if t["nan"] and self.fused:
            print(f"NaN loss detected in {t['nan']} batch(es); they are left out of the epoch averages")
        denom = max(t["seen"], 1)
        acc = t["hit"] / t["seen"] if t["seen"] > 0 else 0.0
        self.train_loss.append(t["loss"] / denom)
        self.train_acc.append(acc)
        return t["loss"] / denom, t["ce"] / denom, t["aux"] / denom, acc

    def early_stop(self, val_loss):
        if val_loss < self.best_val_loss:
            self.best_val_loss = val_loss
            self.counter = 0
            torch.save(self.model.state_dict(), "best_model.pth")
        else:
            self.counter += 1
            if self.counter >= self.patience:
                print(f"Early stopping triggered at epoch {len(self.train_loss)}")
                self.early_stop_flag = True
        return self.early_stop_flag

    def _eval(self):
        self.model.eval()
        tally = self._tally()
        zero = torch.zeros((), device=self.device)
        with torch.no_grad():
            for batch_inputs, batch_labels in self.test_loader:
                feeds = unpack(batch_inputs, self.device)
                y = batch_labels.to(self.device)
                logits, _ = self.model(*feeds, y)
                logits = torch.nan_to_num(logits)  # Trainer.py:135-137 repairs NaN logits; NaN losses are left out below
                ce = self.criterion(logits, y)
                self._count(tally, logits, y, ce, ce, zero)
        t = {k: v.item() for k, v in tally.items()}
        if t["nan"]:
            print(f"NaN loss in test: {t['nan']} batch(es) skipped")
        return t["loss"], t["hit"], t["seen"], t["batches"]

    def test(self):
        total_loss, correct, total_samples, _ = self._eval()
        avg = total_loss / total_samples if total_samples > 0 else float("nan")
        acc = correct / total_samples if total_samples > 0 else 0.0
        self.test_loss.append(avg)
        self.test_acc.append(acc)
        return avg, avg, 0.0, acc

    def plot_progress(self):
        import matplotlib.pyplot as plt
        plt.figure(figsize=(12, 5))
        plt.subplot(1, 2, 1)
        plt.plot(self.train_loss, label="Train Loss")
        plt.plot(self.test_loss, label="Test Loss")
        plt.title("Loss Curve"); plt.xlabel("Epoch"); plt.ylabel("Loss"); plt.legend()
        plt.subplot(1, 2, 2)
        plt.plot(self.train_acc, label="Train Acc")
        plt.plot(self.test_acc, label="Test Acc")
        plt.title("Accuracy Curve"); plt.xlabel("Epoch"); plt.ylabel("Accuracy"); plt.legend()
        plt.tight_layout()
        plt.show()

    def test_with_loaded_model(self, model_path):
        self.model.load_state_dict(torch.load(model_path, weights_only=True))
        total_loss, correct, total_samples, nbatch = self._eval()
        avg = total_loss / nbatch if nbatch > 0 else float("nan")
        acc = correct / total_samples if total_samples > 0 else 0.0
        print(f"Test Loss: {avg:.4f}, CE Loss: {avg:.4f}, Contrastive Loss: {0.0:.4f}, Acc: {acc:.4f}")
        return avg, avg, 0.0, acc

    def _plateau(self, val):  # ReduceLROnPlateau('min', patience=3, factor=0.5) for the fused optimizer
        if val < self._plateau_best * (1 - 1e-4):
            self._plateau_best, self._plateau_bad = val, 0
        else:
            self._plateau_bad += 1
            if self._plateau_bad > 3:
                self.fused_step.lr *= 0.5
                self._plateau_bad = 0

    def run(self, epochs, test_person):
        for epoch in range(1, epochs + 1):
            tr = self.train_epoch(epoch)
            te = self.test()
            if te[0] == te[0]:
                if self.fused:
                    self._plateau(te[0])
                else:
                    self.scheduler.step(te[0])
            print(f"Epoch {epoch}: Train Loss: {tr[0]:.4f}, CE Loss: {tr[1]:.4f}, Contrastive Loss: {tr[2]:.4f}, Acc: {tr[3]:.4f} | "
                  f"Test Loss: {te[0]:.4f}, CE Loss: {te[1]:.4f}, Contrastive Loss: {te[2]:.4f}, Acc: {te[3]:.4f}")
            if self.early_stop(te[0]):
                name = (f"TestPerson{test_person}_epoch{epoch}_TrainLoss{tr[0]:.4f}_Acc{tr[3]:.4f}_TestLoss{te[0]:.4f}"
                        f"_Acc{te[3]:.4f}.pth")
                torch.save(self.model.state_dict(), name)
                break
