"""CPU restatement of the metric network.  TEST INFRASTRUCTURE ONLY (same rule as gan_oracle.py).

The reference's feature extractor is torchvision's Inception-v3 (third-party dependency, absent from /root/reference and from
this image; the reference pins no version - metrics.py:46-51 just calls `models.inception_v3()`), with `aux_logits = False`, an
`fc` of `Sequential(Linear(2048, 100))`, eval mode.  Its published topology (Szegedy et al., "Rethinking the Inception
Architecture", as implemented in torchvision/models/inception.py) is restated here with plain torch CPU ops from the same
layer table the HIP chain walks (jck-generation_amd/inception.py::conv_specs / STEM / BLOCKS): BasicConv2d = conv(bias=False)
+ BatchNorm(eps 1e-3, running statistics) + ReLU; max pools 3x3 stride 2; average pools 3x3 stride 1 pad 1 with
count_include_pad; adaptive average pool; dropout is the identity in eval mode.

PARITY UNPINNED against the reference: neither torchvision nor the fine-tuned weights (./save/iception_v3/loss_bset.pt) are
available offline, so there is no golden output to hold this restatement to.  What it pins is the HIP
chain against an independent implementation on seeded random weights (tests/test_inception_gpu.py), plus the parameter count
of the published architecture (tests/test_inception_cpu.py).
"""
import os
import sys

import torch
import torch.nn.functional as F

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jck-generation_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

BN_EPS = 1e-3


def random_state_dict(seed=0, num_classes=100):
    """A state dict with torchvision's key names and seeded random values of sane scale (He-style conv weights, BatchNorm
    statistics near identity) - stands in for the unavailable fine-tuned weights."""
    from inception import conv_specs
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, cin, cout, k, s, p in conv_specs():
        fan_in = cin * k[0] * k[1]
        sd[f"{name}.conv.weight"] = torch.randn(cout, cin, k[0], k[1], generator=g) * (2.0 / fan_in) ** 0.5
        sd[f"{name}.bn.weight"] = 0.75 + 0.5 * torch.rand(cout, generator=g)
        sd[f"{name}.bn.bias"] = 0.1 * torch.randn(cout, generator=g)
        sd[f"{name}.bn.running_mean"] = 0.1 * torch.randn(cout, generator=g)
        sd[f"{name}.bn.running_var"] = 0.5 + torch.rand(cout, generator=g)
        sd[f"{name}.bn.num_batches_tracked"] = torch.tensor(1)
    sd["fc.0.weight"] = torch.randn(num_classes, 2048, generator=g) * (1.0 / 2048) ** 0.5
    sd["fc.0.bias"] = 0.1 * torch.randn(num_classes, generator=g)
    sd["AuxLogits.conv0.conv.weight"] = torch.randn(128, 768, 1, 1, generator=g)      # present in a torchvision dict; must be ignored
    return sd


def _basic(sd, name, x, k, s, p):
    y = F.conv2d(x, sd[f"{name}.conv.weight"], None, s, p)
    y = F.batch_norm(y, sd[f"{name}.bn.running_mean"], sd[f"{name}.bn.running_var"], sd[f"{name}.bn.weight"], sd[f"{name}.bn.bias"],
                     False, 0.1, BN_EPS)
    return F.relu(y)


def _run(sd, prefix, ops, x):
    from inception import _pair
    for op in ops:
        if op[0] == "max":
            x = F.max_pool2d(x, op[1], op[2], op[3])
        elif op[0] == "avg":
            x = F.avg_pool2d(x, op[1], op[2], op[3])
        elif op[0] == "split":
            x = torch.cat([_basic(sd, f"{prefix}{sub[0]}", x, _pair(sub[2]), _pair(sub[3]), _pair(sub[4])) for sub in op[1:]], 1)
        else:
            x = _basic(sd, f"{prefix}{op[0]}", x, _pair(op[2]), _pair(op[3]), _pair(op[4]))
    return x


@torch.no_grad()
def inception_logits(sd, images):
    """images NCHW fp32 [N,3,299,299] (CPU) -> [N, classes]."""
    from inception import BLOCKS, STEM
    x = _run(sd, "", STEM, images.float())
    for bname, branches in BLOCKS:
        x = torch.cat([_run(sd, bname + ".", ops, x) for ops in branches.values()], 1)
    x = F.adaptive_avg_pool2d(x, (1, 1)).flatten(1)
    return F.linear(x, sd["fc.0.weight"], sd["fc.0.bias"])
