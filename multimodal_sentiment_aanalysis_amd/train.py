"""The reference's two-stage ME-MHACL pipeline (MML_ZYC/train.py) for (image, text) pairs, same function names/signatures:
`contrastive_loss`, `contrastive_pretrain_trainer` (train.py:45-80), `finetune_trainer` (train.py:83-138).

Stage 2 (the CE path: frozen encoder -> Classifier -> CE_a + CE_v, Adam lr as given) and stage 1's supervised-contrastive
loss on the [2B, 2B] similarity matrix (train.py:16-40) run on the HIP modules / kernels.
The optimizer of both stages (`optim.Adam(..., lr=lr)`, train.py:52,94) is the HIP AdamW kernel with zero decay and no clip
(fused.FlatAdam) when the modules live on a GPU; `hip_optimizer=False` keeps torch.optim.Adam on the same parameter views.
Per-batch `.item()` syncs of the reference's loop are replaced by a device-side sum read once per epoch.
This file is host code, like the reference's."""
import torch
import torch.optim as optim

from .engine import CrossEntropyLoss, nt_xent_loss, supcon_loss  # noqa: F401  (nt_xent_loss: ME-MHACL/train.py:47-66)
from .fused import FlatAdam


def _flat_capable(module):
    """Every parameter of `module` belongs to an engine inside it (only those live in flat buffers the HIP optimizer can step)."""
    from .engine import engines_of
    owned = {id(p) for e in engines_of(module) for p in e.parameters()}
    params = list(module.parameters())
    return bool(params) and all(id(p) in owned for p in params)


def _adam(modules, lr, device, hip_optimizer):
    # The reference's train.py accepts ANY nn.Module as encoder / head / classifier: a plain torch module (no engine inside, or
    # parameters beside its engines) keeps torch.optim.Adam on its own tensors instead of failing in FlatAdam.
    if hip_optimizer and torch.device(device).type == "cuda" and all(_flat_capable(m) for m in modules):
        return FlatAdam(modules, lr=lr, device=device)
    return optim.Adam([p for m in modules for p in m.parameters()], lr=lr)


def contrastive_loss(z1, z2, labels, temperature=0.1):
    """train.py:16-40 (SupCon-style, two views): one fused forward + backward launch (mmsa_supcon_fwd_bwd)."""
    return supcon_loss(z1, z2, labels, temperature)


def contrastive_pretrain_trainer(encoder, projection_head, contrastive_loader, num_epochs=20, lr=1e-3, device=None,
                                 hip_optimizer=True):
    device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
    encoder.to(device); projection_head.to(device)
    optimizer = _adam([encoder, projection_head], lr, device, hip_optimizer)
    for epoch in range(num_epochs):
        encoder.train(); projection_head.train()
        total = torch.zeros((), device=device)
        for batch in contrastive_loader:
            a1, b1, c1, a2, b2, c2, labels = [t.to(device) for t in batch]
            z1, z2 = projection_head(encoder(a1, b1, c1)), projection_head(encoder(a2, b2, c2))
            loss = contrastive_loss(z1, z2, labels)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            total += loss.detach().reshape(())
        print(f"Epoch [{epoch + 1}] Contrastive Loss: {total.item() / max(len(contrastive_loader), 1):.4f}")
    return encoder, projection_head


def finetune_trainer(encoder, classifier, train_loader, test_loader, num_epochs=20, lr=1e-3, device=None, hip_optimizer=True):
    device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
    encoder.to(device); classifier.to(device)
    for p in encoder.parameters():
        p.requires_grad = False
    criterion = CrossEntropyLoss()
    optimizer = _adam([classifier], lr, device, hip_optimizer)
    for epoch in range(num_epochs):
        classifier.train()
        total = torch.zeros((), device=device)
        for x1, x2, x3, arousal, valence in train_loader:
            x1, x2, x3 = x1.to(device), x2.to(device), x3.to(device)
            arousal, valence = arousal.to(device), valence.to(device)
            features = encoder(x1, x2, x3).detach()
            out_a, out_v = classifier(features)
            loss = criterion(out_a, arousal) + criterion(out_v, valence)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            total += loss.detach().reshape(())
        print(f"Epoch [{epoch + 1}] Finetune Loss: {total.item() / max(len(train_loader), 1):.4f}")
        classifier.eval()
        hits = torch.zeros(2, dtype=torch.long, device=device)
        n = 0
        with torch.no_grad():
            for x1, x2, x3, arousal, valence in test_loader:
                out_a, out_v = classifier(encoder(x1.to(device), x2.to(device), x3.to(device)))
                hits[0] += (out_a.argmax(1) == arousal.to(device)).sum()
                hits[1] += (out_v.argmax(1) == valence.to(device)).sum()
                n += arousal.size(0)
        ca, cv = hits.tolist()
        print(f"Test Accuracy - Arousal: {ca / max(n, 1):.4f}, Valence: {cv / max(n, 1):.4f}")
    return classifier
