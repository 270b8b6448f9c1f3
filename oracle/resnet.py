"""ResNet-50 v1.5 image encoder restated with torch.nn.functional conv/pool primitives (TEST INFRASTRUCTURE ONLY).

NOT IN THE REFERENCE: /root/reference contains no ResNet (w.r.t. the reference: "parity unpinned"). Pinned against the
third-party transformers==5.15.0 `ResNetModel(ResNetConfig(downsample_in_bottleneck=False))` run in float64
(tests/golden/make_golden.py::gen_resnet -> e2_resnet_mini.npz, e2_resnet50_seed1234.npz; tests/test_oracle_golden.py:
outputs, every parameter gradient, BN running statistics, eval mode: 1e-12 in float64). It restates the public architecture (He et al. 2015; v1.5 = stride on
the 3x3 conv of each downsampling bottleneck): 7x7/2 conv(3->64)+BN+ReLU, 3x3/2 max-pool (pad 1), stages of
[3,4,6,3] bottlenecks (1x1 -> 3x3 -> 1x1(x4), BN after each conv, ReLU after the first two and after the
residual add; 1x1 strided conv + BN downsample on the first block of a stage), global average pool -> [B,2048].
BatchNorm2d: eps 1e-5, momentum 0.1, batch statistics in training. Parameter names follow torchvision's.
Anchors checked in tests: 23 508 032 parameters without fc; output [B, 2048].
"""
import torch
import torch.nn.functional as F

from .policy import FP32

RESNET50 = dict(blocks=(3, 4, 6, 3), widths=(64, 128, 256, 512), expansion=4)
RESNET101 = dict(blocks=(3, 4, 23, 3), widths=(64, 128, 256, 512), expansion=4)


def resnet_param_shapes(cfg):
    """(name, shape, is_buffer) in torchvision order (no prefix); conv weights are logical [O, I, KH, KW]."""
    out = []

    def bn(n, c):
        out.extend([(n + ".weight", (c,), False), (n + ".bias", (c,), False), (n + ".running_mean", (c,), True),
                    (n + ".running_var", (c,), True), (n + ".num_batches_tracked", (), True)])

    out.append(("conv1.weight", (64, 3, 7, 7), False))
    bn("bn1", 64)
    inp = 64
    for si, (nb, w) in enumerate(zip(cfg["blocks"], cfg["widths"])):
        for b in range(nb):
            p = f"layer{si + 1}.{b}."
            stride = 2 if (b == 0 and si > 0) else 1
            out.append((p + "conv1.weight", (w, inp, 1, 1), False)); bn(p + "bn1", w)
            out.append((p + "conv2.weight", (w, w, 3, 3), False)); bn(p + "bn2", w)
            out.append((p + "conv3.weight", (w * 4, w, 1, 1), False)); bn(p + "bn3", w * 4)
            if b == 0:
                out.append((p + "downsample.0.weight", (w * 4, inp, 1, 1), False)); bn(p + "downsample.1", w * 4)
            inp = w * 4
    return out


def _bn(sd, p, x, training, pol, eps=1e-5, momentum=0.1):
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean((0, 2, 3))
        var = ((x - mean.view(1, -1, 1, 1)) ** 2).mean((0, 2, 3))
        with torch.no_grad():
            sd[p + ".running_mean"].mul_(1 - momentum).add_(momentum * mean.detach())
            sd[p + ".running_var"].mul_(1 - momentum).add_(momentum * var.detach() * (n / max(n - 1, 1)))
            sd[p + ".num_batches_tracked"].add_(1)
    else:
        mean, var = sd[p + ".running_mean"], sd[p + ".running_var"]
    inv = 1.0 / torch.sqrt(var + eps)
    return (x - mean.view(1, -1, 1, 1)) * (inv * sd[p + ".weight"]).view(1, -1, 1, 1) + sd[p + ".bias"].view(1, -1, 1, 1)


def resnet_forward(sd, p, image, cfg, training, pol=FP32, trace=None):
    """image [B,3,H,W] fp32 -> pooled features [B, 512*expansion]. `trace` (dict) collects per-conv tensors for tests.
    Every stored activation is a named rounding point of the policy (names: stem.z, stem.y, layer{s}.{b}.{c1,c2,c3,ds}.{z,y},
    pooled; c3.y is the block output) — the tensors the HIP engine keeps in its workspace (resnet_engine.hip ConvWs)."""
    q, qw = pol.q, pol.qw
    conv = lambda x, n, s, pad, nm: q(F.conv2d(x, qw(sd[p + n]), stride=s, padding=pad), nm)
    x = q(image, "image")
    x = q(torch.relu(_bn(sd, p + "bn1", conv(x, "conv1.weight", 2, 3, "stem.z"), training, pol)), "stem.y")
    x = F.max_pool2d(x, 3, 2, 1)
    for si, nb in enumerate(cfg["blocks"]):
        for b in range(nb):
            L = f"layer{si + 1}.{b}."
            bp = p + L
            stride = 2 if (b == 0 and si > 0) else 1
            idn = x
            z1 = conv(x, L + "conv1.weight", 1, 0, L + "c1.z")
            y1 = q(torch.relu(_bn(sd, bp + "bn1", z1, training, pol)), L + "c1.y")
            if trace is not None:
                for nm, t in (("c1.z", z1), ("c1.y", y1)):
                    if t.requires_grad:
                        t.retain_grad()
                    trace[L + nm] = t
            y = q(torch.relu(_bn(sd, bp + "bn2", conv(y1, L + "conv2.weight", stride, 1, L + "c2.z"), training, pol)), L + "c2.y")
            z = _bn(sd, bp + "bn3", conv(y, L + "conv3.weight", 1, 0, L + "c3.z"), training, pol)
            if b == 0:
                idn = q(_bn(sd, bp + "downsample.1", conv(x, L + "downsample.0.weight", stride, 0, L + "ds.z"), training, pol),
                        L + "ds.y")
            x = q(torch.relu(z + idn), L + "c3.y")
    return q(x.mean((2, 3)), "pooled")
