"""Shared helpers for the parity tests (tests only: may import the oracle)."""
import torch

MINI_BERT = dict(hidden=128, layers=2, heads=2, intermediate=512, vocab=1000, max_pos=64, type_vocab=2, ln_eps=1e-12)
MINI_RESNET = dict(blocks=(1, 1, 1, 1), widths=(64, 64, 128, 128))
MINI_RESNET2 = dict(blocks=(2, 1, 1, 1), widths=(64, 64, 64, 64))


def oracle_cfg(bert, resnet, **kw):
    c = dict(bert=dict(bert), resnet=dict(blocks=tuple(resnet["blocks"]), widths=tuple(resnet["widths"]), expansion=4))
    c.update(kw)
    return c


def cpu_state(module):
    """Detached fp32 CPU copy of a product module's state_dict (conv weights become plain contiguous OIHW)."""
    return {k: v.detach().to("cpu").clone().contiguous() for k, v in module.state_dict().items()}


def rel_err(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


def synth_batch(B, S, H, W, vocab, seed=1234, dev="cpu"):
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, 3, H, W, generator=g)
    ids = torch.randint(0, vocab, (B, S), generator=g)
    ids[:, 0] = 101 % vocab
    mask = torch.ones(B, S)
    labels = torch.randint(0, 3, (B,), generator=g)
    return image.to(dev), ids.to(dev), mask.to(dev), labels.to(dev)
