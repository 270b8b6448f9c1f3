"""ResNet-50 v1.5 image encoder restated with torch.nn.functional conv/pool primitives (TEST INFRASTRUCTURE ONLY).

NOT IN THE REFERENCE: /root/reference contains no ResNet and torchvision is not installed, so this is
"parity unpinned" (oracle/__init__.py). It restates the public architecture (He et al. 2015; v1.5 = stride on
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
    """image [B,3,H,W] fp32 -> pooled features [B, 512*expansion]. `trace` (dict) collects per-conv tensors for tests."""
    q = pol.q
    conv = lambda x, n, s, pad: q(F.conv2d(x, q(sd[p + n]), stride=s, padding=pad))
    x = q(image)
    x = q(torch.relu(_bn(sd, p + "bn1", conv(x, "conv1.weight", 2, 3), training, pol)))
    x = F.max_pool2d(x, 3, 2, 1)
    for si, nb in enumerate(cfg["blocks"]):
        for b in range(nb):
            bp = f"{p}layer{si + 1}.{b}."
            stride = 2 if (b == 0 and si > 0) else 1
            idn = x
            z1 = conv(x, f"layer{si + 1}.{b}.conv1.weight", 1, 0)
            y1 = q(torch.relu(_bn(sd, bp + "bn1", z1, training, pol)))
            if trace is not None:
                for nm, t in (("c1.z", z1), ("c1.y", y1)):
                    if t.requires_grad:
                        t.retain_grad()
                    trace[f"layer{si + 1}.{b}.{nm}"] = t
            y = q(torch.relu(_bn(sd, bp + "bn2", conv(y1, f"layer{si + 1}.{b}.conv2.weight", stride, 1), training, pol)))
            z = _bn(sd, bp + "bn3", conv(y, f"layer{si + 1}.{b}.conv3.weight", 1, 0), training, pol)
            if b == 0:
                idn = q(_bn(sd, bp + "downsample.1", conv(x, f"layer{si + 1}.{b}.downsample.0.weight", stride, 0),
                            training, pol))
            x = q(torch.relu(z + idn))
    return q(x.mean((2, 3)))
