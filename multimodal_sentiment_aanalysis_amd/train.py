"""The reference's two-stage ME-MHACL pipeline (MML_ZYC/train.py) for (image, text) pairs, same function names/signatures:
`contrastive_loss`, `contrastive_pretrain_trainer` (train.py:45-80), `finetune_trainer` (train.py:83-138).

Stage 2 (the CE path: frozen encoder -> Classifier -> CE_a + CE_v, Adam lr as given) and stage 1's supervised-contrastive
loss on the [2B, 2B] similarity matrix (train.py:16-40) run on the HIP modules / kernels.
This file is host code, like the reference's."""
import torch
import torch.optim as optim

from .engine import CrossEntropyLoss, nt_xent_loss, supcon_loss  # noqa: F401  (nt_xent_loss: ME-MHACL/train.py:47-66)


def contrastive_loss(z1, z2, labels, temperature=0.1):
    """train.py:16-40 (SupCon-style, two views): one fused forward + backward launch (mmsa_supcon_fwd_bwd)."""
    return supcon_loss(z1, z2, labels, temperature)


def contrastive_pretrain_trainer(encoder, projection_head, contrastive_loader, num_epochs=20, lr=1e-3, device=None):
    device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
    encoder.to(device); projection_head.to(device)
    optimizer = optim.Adam(list(encoder.parameters()) + list(projection_head.parameters()), lr=lr)
    for epoch in range(num_epochs):
        encoder.train(); projection_head.train()
        total = 0.0
        for batch in contrastive_loader:
            a1, b1, c1, a2, b2, c2, labels = [t.to(device) for t in batch]
            z1, z2 = projection_head(encoder(a1, b1, c1)), projection_head(encoder(a2, b2, c2))
            loss = contrastive_loss(z1, z2, labels)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            total += loss.item()
        print(f"Epoch [{epoch + 1}] Contrastive Loss: {total / max(len(contrastive_loader), 1):.4f}")
    return encoder, projection_head


def finetune_trainer(encoder, classifier, train_loader, test_loader, num_epochs=20, lr=1e-3, device=None):
    device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
    encoder.to(device); classifier.to(device)
    for p in encoder.parameters():
        p.requires_grad = False
    criterion = CrossEntropyLoss()
    optimizer = optim.Adam(classifier.parameters(), lr=lr)
    for epoch in range(num_epochs):
        classifier.train()
        total = 0.0
        for x1, x2, x3, arousal, valence in train_loader:
            x1, x2, x3 = x1.to(device), x2.to(device), x3.to(device)
            arousal, valence = arousal.to(device), valence.to(device)
            features = encoder(x1, x2, x3).detach()
            out_a, out_v = classifier(features)
            loss = criterion(out_a, arousal) + criterion(out_v, valence)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            total += loss.item()
        print(f"Epoch [{epoch + 1}] Finetune Loss: {total / max(len(train_loader), 1):.4f}")
        classifier.eval()
        ca = cv = n = 0
        with torch.no_grad():
            for x1, x2, x3, arousal, valence in test_loader:
                out_a, out_v = classifier(encoder(x1.to(device), x2.to(device), x3.to(device)))
                ca += (out_a.argmax(1).cpu() == arousal).sum().item()
                cv += (out_v.argmax(1).cpu() == valence).sum().item()
                n += arousal.size(0)
        print(f"Test Accuracy - Arousal: {ca / max(n, 1):.4f}, Valence: {cv / max(n, 1):.4f}")
    return classifier
