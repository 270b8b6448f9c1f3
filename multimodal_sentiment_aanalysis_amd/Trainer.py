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

    # ---- one epoch (Trainer.py:42-105)
    def train_epoch(self, epoch):
        self.model.train()
        total_loss = total_ce = total_con = 0.0
        correct = total_samples = 0
        for data_dict, labels in self.train_loader:
            x1, x2, x3 = unpack(data_dict, self.device)
            labels = labels.to(self.device)
            if self.fused:
                loss_t, outputs = self.fused_step.step(x1, x2, x3, labels)
                loss_v = ce_v = loss_t.item()
                con_v = 0.0
                if loss_v != loss_v:  # NaN: the reference skips the batch (Trainer.py:74-76)
                    print("NaN loss detected, skipping batch")
                    continue
            else:
                self.optimizer.zero_grad()
                outputs, contrastive_loss = self.model(x1, x2, x3, labels)
                if torch.isnan(outputs).any():
                    print("Warning: Model output contains NaN!")
                    outputs = torch.nan_to_num(outputs)
                ce_loss = self.criterion(outputs, labels)
                loss = ce_loss + self.contrastive_weight * contrastive_loss
                if torch.isnan(loss):
                    print("NaN loss detected, skipping batch")
                    continue
                loss.backward()
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), 1.0)
                self.optimizer.step()
                loss_v, ce_v, con_v = loss.item(), ce_loss.item(), contrastive_loss.item()
            total_loss += loss_v
            total_ce += ce_v
            total_con += con_v
            _, predicted = torch.max(outputs.data, 1)
            correct += (predicted == labels).sum().item()
            total_samples += labels.size(0)
        n = max(total_samples, 1)
        acc = correct / total_samples if total_samples > 0 else 0.0
        self.train_loss.append(total_loss / n)
        self.train_acc.append(acc)
        return total_loss / n, total_ce / n, total_con / n, acc

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
        total_loss = 0.0
        correct = total_samples = 0
        nbatch = 0
        with torch.no_grad():
            for data_dict, labels in self.test_loader:
                x1, x2, x3 = unpack(data_dict, self.device)
                labels = labels.to(self.device)
                outputs, _ = self.model(x1, x2, x3, labels)
                if torch.isnan(outputs).any():
                    print("Warning: Test output contains NaN!")
                    outputs = torch.nan_to_num(outputs)
                loss = self.criterion(outputs, labels)
                if torch.isnan(loss):
                    print("NaN loss in test, skipping batch")
                    continue
                total_loss += loss.item()
                _, predicted = torch.max(outputs.data, 1)
                correct += (predicted == labels).sum().item()
                total_samples += labels.size(0)
                nbatch += 1
        return total_loss, correct, total_samples, nbatch

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
