"""MultiTaskTrainer with the reference's surface (MML_ZYC/dataLoader/MultiTaskTrainer.py) for (image, text) batches —
SURVEY.md §8(f) row N2: the five-phase curriculum with frozen sub-graphs, per-phase AdamW + ReduceLROnPlateau, running
on the HIP modules.

What the reference does (and this keeps): every `train_epoch_phase_*` call freezes everything, unfreezes the phase's
modules, builds a FRESH AdamW(lr 1e-4, weight_decay 1e-4) + ReduceLROnPlateau (MultiTaskTrainer.py:55-177: the setup runs
inside the epoch function, so optimizer state restarts every epoch), then per batch: zero_grad -> forward with labels ->
phase objective -> backward -> clip_grad_norm_(1.0) -> step (:179-467). Objectives: the three feature phases train on
one contrastive term each (c1 / c2 / c3), phase 2 on the arousal cross-entropy, phase 3 on the valence cross-entropy.
`evaluate` (:469-515) reports a_loss + v_loss, the summed contrastive terms and both accuracies; epoch means are divided
by `len(loader.dataset)`. Batches are 5-tuples `(x1, x2, x3, arousal, valence)` (:186).

Module names: the reference's attribute names are kept as aliases on the model (`eeg_net`, `eye_net`, `pps_net`,
`cross_attn_e2p`, `cross_attn_p2e`; MultimodalModel.py here) — slot 1 is the ME-MHACL fusion token (the whole encoder),
slot 2 the text encoder, slot 3 the image encoder.

How the phases reach the kernels: parameters are views of the engines' flat buffers, so torch.optim works on any subset;
an engine whose parameters are all frozen and whose inputs need no gradient is not entered by the backward at all (no
data-gradient, no weight-gradient kernels); after a foreign optimizer step the engines refresh their bf16 working copies
on the next forward. Sums are kept on the device: one host read per epoch.
"""
import torch
import torch.optim as optim
from torch.optim.lr_scheduler import ReduceLROnPlateau

from ..engine import CrossEntropyLoss
from ..fused import PhaseOptimizer, Plateau

_KEYS = ("loss", "a_loss", "v_loss", "c_loss", "a_acc", "v_acc")

# phase -> (modules unfrozen, modules handed to the optimizer (None: every unfrozen parameter), objective,
#           scheduler (patience, factor), attribute prefix of the stored optimizer / scheduler)
_PHASES = {
    "eeg": (("eeg_net",), ("eeg_net",), "c1", (3, 0.5), "phase1"),
    "eye": (("eye_net",), ("eye_net",), "c2", (3, 0.5), "phase1"),
    "pps": (("pps_net",), ("pps_net",), "c3", (3, 0.5), "phase1"),
    "2": (("eeg_net", "eye_net", "pps_net", "cross_attn_e2p", "cross_attn_p2e", "attention_weights", "fusion",
           "arousal_head"), None, "arousal", (2, 0.2), "phase2"),
    "3": (("cross_attn_e2p", "cross_attn_p2e", "attention_weights", "fusion", "valence_head"), ("valence_head",),
          "valence", (2, 0.1), "phase3"),
}


class MultiTaskTrainer:
    def __init__(self, model, train_loader, test_loader, device="cuda", test_person=-1, hip_optimizer=None):
        """hip_optimizer: None = on a GPU (the default), the per-phase clip + AdamW run as HIP kernels over the phase's
        sub-ranges of the flat buffers (fused.PhaseOptimizer); False = torch.optim.AdamW + clip_grad_norm_, literally the
        reference's objects (works because the parameters are ordinary views)."""
        self.hip_optimizer = hip_optimizer
        self.model = model.to(device)
        self.train_loader, self.test_loader = train_loader, test_loader
        self.device, self.test_person = device, test_person
        for slot in ("phase1", "phase2", "phase3"):
            setattr(self, slot + "_optimizer", None)
            setattr(self, slot + "_scheduler", None)
        self.criterion = {"arousal": CrossEntropyLoss(), "valence": CrossEntropyLoss()}
        self.metrics = {split: {k: [] for k in _KEYS} for split in ("train", "test", "val")}
        self.best_val_loss, self.patience, self.counter = float("inf"), 5, 0
        self.clip_norms = None  # a list: receives every batch's total gradient norm (a device scalar; what clip_grad_norm_ returns)

    # ---- helpers
    def _compute_metrics(self, outputs, labels):
        return {"a_acc": (outputs[0].argmax(1) == labels[0]).float().mean().item(),
                "v_acc": (outputs[1].argmax(1) == labels[1]).float().mean().item()}

    def _freeze_all(self):
        for p in self.model.parameters():
            p.requires_grad = False

    def _setup(self, phase):
        unfreeze, opt_on, _, (patience, factor), slot = _PHASES[phase]
        self._freeze_all()
        for name in unfreeze:
            for p in getattr(self.model, name).parameters():
                p.requires_grad = True
        state = self._flat_state()
        if state is not None:  # HIP kernels over sub-ranges: AdamW on opt_on, clip over every parameter that holds a gradient
            trainable = [getattr(self.model, name) for name in unfreeze]
            owned = trainable if opt_on is None else [getattr(self.model, name) for name in opt_on]
            opt = PhaseOptimizer(state, owned, self.model, lr=1e-4, weight_decay=1e-4, max_norm=1.0)
            sched = Plateau(opt, patience=patience, factor=factor)
        else:
            if opt_on is None:
                params = [p for p in self.model.parameters() if p.requires_grad]
            else:
                params = [p for name in opt_on for p in getattr(self.model, name).parameters()]
            opt = optim.AdamW(params, lr=1e-4, weight_decay=1e-4)
            sched = ReduceLROnPlateau(opt, mode="min", patience=patience, factor=factor)
        setattr(self, slot + "_optimizer", opt)
        setattr(self, slot + "_scheduler", sched)
        return opt

    def _flat_state(self):
        """The model's flat parameter state when the HIP optimizer path applies (GPU, engines materialized), else None."""
        if self.hip_optimizer is False or torch.device(self.device).type != "cuda":
            return None
        from ..engine import engines_of, materialize
        st = getattr(self.model, "_flat_state", None)
        if st is None or not st.valid():
            if not engines_of(self.model):
                return None
            st = materialize(self.model, torch.device(self.device) if torch.device(self.device).index is not None
                             else torch.device("cuda", torch.cuda.current_device()))
        return st

    def _setup_phase_EEGnet(self):
        return self._setup("eeg")

    def _setup_phase_EYEnet(self):
        return self._setup("eye")

    def _setup_phase_PPSnet(self):
        return self._setup("pps")

    def _setup_phase2(self):
        return self._setup("2")

    def _setup_phase3(self):
        return self._setup("3")

    def _feeds(self, batch):
        x1, x2, x3, arousal, valence = batch
        # the reference casts its three inputs with .float(); token ids stay integer here (the text encoder takes int64)
        ids = x2.to(self.device)
        ids = ids if not ids.is_floating_point() else ids.long()
        return (x1.to(self.device).float(), ids, x3.to(self.device).float()), (arousal.to(self.device), valence.to(self.device))

    def _zeros(self):
        return {k: torch.zeros((), dtype=torch.float64, device=self.device) for k in _KEYS}

    def _add(self, total, n, outs, labels, loss, a_loss, v_loss, c_loss):
        total["loss"] += loss.detach().double().reshape(()) * n
        total["a_loss"] += a_loss.detach().double().reshape(()) * n
        total["v_loss"] += v_loss.detach().double().reshape(()) * n
        total["c_loss"] += c_loss.detach().double().reshape(()) * n
        total["a_acc"] += (outs[0].detach().argmax(1) == labels[0]).double().sum()
        total["v_acc"] += (outs[1].detach().argmax(1) == labels[1]).double().sum()

    def _record(self, split, total, n):
        host = {k: v.item() for k, v in total.items()}
        for k in _KEYS:
            self.metrics[split][k].append(host[k] / n)
        return {k: v[-1] for k, v in self.metrics[split].items()}

    # ---- one training epoch of a phase
    def _train_phase(self, phase, epoch):
        self.model.train()
        opt = self._setup(phase)
        objective = _PHASES[phase][2]
        total, zero = self._zeros(), torch.zeros((), device=self.device)
        for batch in self.train_loader:
            feeds, labels = self._feeds(batch)
            opt.zero_grad()
            a_out, v_out, c1, c2, c3 = self.model(*feeds, labels=labels)
            a_loss = v_loss = c_loss = zero
            if objective == "arousal":
                loss = a_loss = self.criterion["arousal"](a_out, labels[0])
            elif objective == "valence":
                loss = v_loss = self.criterion["valence"](v_out, labels[1])
            else:
                loss = c_loss = {"c1": c1, "c2": c2, "c3": c3}[objective]
            loss.backward()
            if not isinstance(opt, PhaseOptimizer):  # (the HIP optimizer's step() is clip + AdamW in one)
                # every parameter that holds a gradient — also stale ones of modules frozen in this phase (MultiTaskTrainer.py:205)
                total_norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), 1.0)
            opt.step()
            if self.clip_norms is not None:
                self.clip_norms.append(opt.adamw.norm_out[0].clone() if isinstance(opt, PhaseOptimizer) else total_norm.detach())
            self._add(total, labels[0].shape[0], (a_out, v_out), labels, loss, a_loss, v_loss, c_loss)
        return self._record("train", total, len(self.train_loader.dataset))

    def train_epoch_phase_eeg(self, epoch):
        return self._train_phase("eeg", epoch)

    def train_epoch_phase_eye(self, epoch):
        return self._train_phase("eye", epoch)

    def train_epoch_phase_pps(self, epoch):
        return self._train_phase("pps", epoch)

    def train_epoch_phase2(self, epoch):
        return self._train_phase("2", epoch)

    def train_epoch_phase3(self, epoch):
        return self._train_phase("3", epoch)

    def evaluate(self, mode="test"):
        self.model.eval()
        total = self._zeros()
        with torch.no_grad():
            for batch in self.test_loader:
                feeds, labels = self._feeds(batch)
                a_out, v_out, c1, c2, c3 = self.model(*feeds, labels=labels)
                a_loss = self.criterion["arousal"](a_out, labels[0])
                v_loss = self.criterion["valence"](v_out, labels[1])
                self._add(total, labels[0].shape[0], (a_out, v_out), labels, a_loss + v_loss, a_loss, v_loss, c1 + c2 + c3)
        return self._record(mode, total, len(self.test_loader.dataset))

    def early_stopping(self, val_loss):
        if val_loss < self.best_val_loss:
            self.best_val_loss, self.counter = val_loss, 0
            torch.save(self.model.state_dict(), "best_model.pth")
            return False
        self.counter += 1
        if self.counter >= self.patience:
            print("Early stopping triggered!")
            return True
        return False

    def visualize_progress(self):
        import matplotlib.pyplot as plt
        fig, axes = plt.subplots(1, 3, figsize=(15, 6))
        for ax, key, title in zip(axes, ("loss", "a_acc", "v_acc"), ("Loss", "Arousal accuracy", "Valence accuracy")):
            for split in ("train", "test"):
                ax.plot(self.metrics[split][key], label=f"{split} {key}")
            ax.set_title(title); ax.set_xlabel("epoch"); ax.legend()
        fig.tight_layout()
        plt.show()

    def run(self, epochs_phaseEEG, epochs_phaseEYE, epochs_phasePPS, epochs_phase2, epochs_phase3):
        plan = (("Phase EEGnet : Training Feature Extractors with Contrastive Loss", self.train_epoch_phase_eeg, epochs_phaseEEG, "phase1"),
                ("Phase EYEnet : Training Feature Extractors with Contrastive Loss", self.train_epoch_phase_eye, epochs_phaseEYE, "phase1"),
                ("Phase PPSnet : Training Feature Extractors with Contrastive Loss", self.train_epoch_phase_pps, epochs_phasePPS, "phase1"),
                ("Phase 2: Training Fusion Module and Arousal Head", self.train_epoch_phase2, epochs_phase2, "phase2"),
                ("Phase 3: Training Valence Head Only", self.train_epoch_phase3, epochs_phase3, "phase3"))
        name = f"TestPerson{self.test_person}"
        for title, epoch_fn, epochs, slot in plan:
            print(title)
            for epoch in range(1, epochs + 1):
                tr = epoch_fn(epoch)
                te = self.evaluate()
                getattr(self, slot + "_scheduler").step(te["loss"])
                print(f"\nEpoch {epoch} Results:\nTrain Loss: {tr['loss']:.4f} | A Acc: {tr['a_acc']:.2%} | V Acc: {tr['v_acc']:.2%}"
                      f" | C Loss: {tr['c_loss']:.4f}\nTest  Loss: {te['loss']:.4f} | A Acc: {te['a_acc']:.2%} | V Acc: {te['v_acc']:.2%}")
                if slot == "phase3":
                    name = f"TestPerson{self.test_person}_ArousalAcc{te['a_acc']:.2f}_ValenceAcc{te['v_acc']:.2f}"
        self.visualize_progress()
        torch.save(self.model.state_dict(), name + ".pth")
