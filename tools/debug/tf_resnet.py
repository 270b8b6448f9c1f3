"""Teacher-forced ResNet backward: per-tensor error table (debug aid for tests/test_engines_gpu.py)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_engines_gpu as T
from oracle.policy import Policy
from oracle.resnet import resnet_forward
dev = torch.device("cuda:0")
cfgs = {"mini": (T.MINI_RESNET, 4, 96), "r50": (dict(blocks=(3, 4, 6, 3), widths=(64, 128, 256, 512)), 16, 96)}
rcfg, B, HW = cfgs[sys.argv[1] if len(sys.argv) > 1 else "mini"]
round_grads = (sys.argv[2] != "noround") if len(sys.argv) > 2 else True
net, sd, ocfg, image, wgt = T._resnet_case(rcfg, B, HW, dev)
tr = {}
with torch.no_grad():
    resnet_forward({k: v.clone() for k, v in sd.items()}, "resnet.", image, ocfg, True, Policy("bf16", trace=tr))
shapes = {k: tuple(v.shape) for k, v in tr.items()}
net.to(dev); net.train()
out = net(image.to(dev))
forced = T._saved_resnet_activations(out, ocfg, shapes)
(out * wgt.to(dev)).sum().backward()
pol = Policy("bf16", round_grads=round_grads, forced=forced)
def feat_fn(work):
    f = resnet_forward(work, "resnet.", image, ocfg, True, pol)
    return f @ pol.qw(work["proj.weight"]).t() + work["proj.bias"]
names = [n for n, _ in net.named_parameters()]
ref = T._oracle_grads({k: v.clone() for k, v in sd.items()}, names, lambda w: (feat_fn(w) * wgt).sum())
got = T._grads(net)
for n in names:
    r, g = ref[n].double(), got[n].double()
    print(f"{n:45s} relL2 {((g - r).norm() / r.norm()).item():.3e}  |ref| {r.norm().item():.3e}  cos {(g.flatten() @ r.flatten() / (g.norm() * r.norm())).item():.6f} ratio {(g.norm()/r.norm()).item():.4f}")
