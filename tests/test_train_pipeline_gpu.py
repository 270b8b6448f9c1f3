"""GPU: the reference's two-stage ME-MHACL pipeline (MML_ZYC/train.py:45-80 contrastive pre-training, :83-138 frozen-encoder
fine-tuning) through the mirror `multimodal_sentiment_aanalysis_amd/train.py` on mini encoders — executed, not just imported:
stage 1 trains encoder + projection head on the two-view supervised-contrastive loss, stage 2 freezes the encoder and trains
the classifier on CE_a + CE_v. The Adam steps run on the HIP optimizer kernel (fused.FlatAdam = AdamW with zero decay, no clip)
and are compared with torch.optim.Adam over the same parameter views."""
import copy

import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

pytestmark = pytest.mark.gpu

from multimodal_sentiment_aanalysis_amd import Classifier, MultiModalEncoder, ProjectionHead
from multimodal_sentiment_aanalysis_amd import train as T
from multimodal_sentiment_aanalysis_amd.fused import FlatAdam

from util import MINI_BERT, MINI_RESNET


def _modules(seed):
    torch.manual_seed(seed)
    enc = MultiModalEncoder(MINI_BERT, MINI_RESNET)
    enc.image_net.precision = enc.text_net.precision = "fp32"
    proj, clf = ProjectionHead(), Classifier()
    proj.dropout_p = clf.dropout_p = 0.0  # the two optimizer paths must see the same function
    return enc, proj, clf


def _loaders():
    g = torch.Generator().manual_seed(5)
    n, S = 16, 16

    def view():
        return (torch.randn(n, 3, 64, 64, generator=g), torch.randint(0, MINI_BERT["vocab"], (n, S), generator=g), torch.ones(n, S))

    a, b = view(), view()
    labels = torch.randint(0, 3, (n,), generator=g)
    arousal, valence = torch.randint(0, 3, (n,), generator=g), torch.randint(0, 3, (n,), generator=g)
    con = DataLoader(TensorDataset(*a, *b, labels.float()), batch_size=8)
    sup = DataLoader(TensorDataset(*a, arousal, valence), batch_size=8)
    return con, sup


def _params(*mods):
    return {f"{i}.{n}": p.detach().cpu().clone() for i, m in enumerate(mods) for n, p in m.named_parameters()}


def _agree(a, b, lr, steps, what):
    """Two Adam implementations from the same start: an early Adam step moves every element by ~lr whatever its gradient's size,
    so an element whose gradient is rounding noise may step the other way (bound 2.2 lr per step); all but a sliver agree."""
    worst, frac, tot = 0.0, 0, 0
    for k in a:
        d = (a[k].double() - b[k].double()).abs()
        worst = max(worst, d.max().item())
        frac += (d > 0.02 * lr).sum().item()
        tot += d.numel()
    assert worst <= 2.2 * lr * steps, f"{what}: {worst:.3e}"
    assert frac / tot < 1e-2, f"{what}: {frac / tot:.3e} of the elements differ by more than 2 % of a step"


def test_train_py_two_stage_pipeline(dev, capsys):
    con, sup = _loaders()
    runs = {}
    for hip in (True, False):
        enc, proj, clf = _modules(0)
        start = _params(enc, proj)
        T.contrastive_pretrain_trainer(enc, proj, con, num_epochs=1, lr=1e-3, device=dev, hip_optimizer=hip)
        after1 = _params(enc, proj)
        moved = sum(not torch.equal(start[k], after1[k]) for k in start)
        assert moved > 0.9 * len(start), f"stage 1 moved {moved} of {len(start)} tensors"
        assert all(torch.isfinite(v).all() for v in after1.values())
        # stage 2 is compared from IDENTICAL encoders (the two stage-1 runs differ by Adam-step rounding, which would show up as
        # different features, hence different classifier gradients): the second run adopts the first run's encoder
        if "enc_state" in runs:
            enc.load_state_dict(runs["enc_state"])
        else:
            runs["enc_state"] = {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}
        enc_before, clf_before = _params(enc), _params(clf)
        T.finetune_trainer(enc, clf, sup, sup, num_epochs=1, lr=1e-3, device=dev, hip_optimizer=hip)
        enc_after, clf_after = _params(enc), _params(clf)
        assert all(torch.equal(enc_before[k], enc_after[k]) for k in enc_before), "stage 2 must leave the frozen encoder bit-exact"
        assert all(not p.requires_grad for p in enc.parameters())
        assert any(not torch.equal(clf_before[k], clf_after[k]) for k in clf_before)
        runs[hip] = (after1, clf_after)
    out = capsys.readouterr().out
    assert out.count("Contrastive Loss:") == 2 and out.count("Finetune Loss:") == 2 and "nan" not in out.lower(), out
    _agree(runs[True][0], runs[False][0], 1e-3, 2, "stage 1: HIP Adam vs torch.optim.Adam")
    _agree(runs[True][1], runs[False][1], 1e-3, 2, "stage 2: HIP Adam vs torch.optim.Adam")


def test_flat_adam_is_torch_adam(dev):
    """fused.FlatAdam (AdamW kernel, zero decay, no clip) against torch.optim.Adam on the same gradients, three steps."""
    _, proj, _ = _modules(1)
    proj.to(dev)
    opt = FlatAdam([proj], lr=1e-3, device=dev)
    ref = copy.deepcopy({n: p.detach().clone() for n, p in proj.named_parameters()})
    ref_p = {n: torch.nn.Parameter(v.clone()) for n, v in ref.items()}
    ref_opt = torch.optim.Adam(list(ref_p.values()), lr=1e-3)
    g = torch.Generator().manual_seed(3)
    for step in range(3):
        for n, p in proj.named_parameters():
            gr = torch.randn(p.shape, generator=g) * 10.0 ** float(torch.randint(-4, 2, (1,), generator=g))
            p.grad = None
            ref_p[n].grad = gr.to(dev)
        # hand the same gradients to the flat buffer through the views the engine attaches
        proj._attach_grads(zero=True)
        for n, p in proj.named_parameters():
            p.grad.copy_(ref_p[n].grad)
        opt.step()
        ref_opt.step()
    for n, p in proj.named_parameters():
        err = (p.detach() - ref_p[n].detach()).abs().max().item()
        assert err <= 2e-6 + 1e-5 * ref_p[n].detach().abs().max().item(), f"{n}: {err:.3e}"
