"""Inception-v3 inference on the MI355X for the evaluation branch (reference metrics.py:46-51,80-94).

The reference builds `torchvision.models.inception_v3()`, sets `aux_logits = False`, replaces `fc` by
`Sequential(Linear(2048, 100))`, loads `./save/iception_v3/loss_bset.pt` and runs it in eval mode on 299x299 ImageNet-
normalised images; Inception Score and FID are computed on its 100 logits.  This module is that network as a chain of
hand-written NHWC fp32 kernels behind the C ABI (csrc/infer.hip): every BasicConv2d (conv without bias + BatchNorm(eps 1e-3)
+ ReLU) is ONE launch with the eval-mode BatchNorm folded into a per-channel scale and shift, and every branch of an Inception
block writes straight into its channel slice of the block's output (no concatenation pass).

`InceptionV3Hip.from_file(path)` is the local-weights loader: it takes a state dict with torchvision's key names
(`Conv2d_1a_3x3.conv.weight`, `Mixed_5b.branch1x1.bn.running_mean`, ..., `fc.0.weight`; `AuxLogits.*` entries are ignored like
the reference ignores that head at inference) and needs neither torchvision nor network access.  The fine-tuned weights are
not in the reference repository, so feature parity against the reference is unpinned; the kernels are
checked against a plain-torch CPU restatement of the same topology with seeded random weights (tests/test_inception_gpu.py).
No CPU fallback: a CPU tensor raises.
"""
import torch

from hipgan._lib import JckError, cur_stream, lib

BN_EPS = 1e-3            # torchvision BasicConv2d: nn.BatchNorm2d(out_channels, eps=0.001)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


# (name, out_channels, kernel, stride, padding) of every BasicConv2d, block by block (torchvision/models/inception.py)
def inception_a(pool_features):
    return {"branch1x1": [("branch1x1", 64, 1, 1, 0)],
            "branch5x5": [("branch5x5_1", 48, 1, 1, 0), ("branch5x5_2", 64, 5, 1, 2)],
            "branch3x3dbl": [("branch3x3dbl_1", 64, 1, 1, 0), ("branch3x3dbl_2", 96, 3, 1, 1), ("branch3x3dbl_3", 96, 3, 1, 1)],
            "branch_pool": [("avg", 3, 1, 1), ("branch_pool", pool_features, 1, 1, 0)]}


def inception_c(c7):
    return {"branch1x1": [("branch1x1", 192, 1, 1, 0)],
            "branch7x7": [("branch7x7_1", c7, 1, 1, 0), ("branch7x7_2", c7, (1, 7), 1, (0, 3)), ("branch7x7_3", 192, (7, 1), 1, (3, 0))],
            "branch7x7dbl": [("branch7x7dbl_1", c7, 1, 1, 0), ("branch7x7dbl_2", c7, (7, 1), 1, (3, 0)),
                             ("branch7x7dbl_3", c7, (1, 7), 1, (0, 3)), ("branch7x7dbl_4", c7, (7, 1), 1, (3, 0)),
                             ("branch7x7dbl_5", 192, (1, 7), 1, (0, 3))],
            "branch_pool": [("avg", 3, 1, 1), ("branch_pool", 192, 1, 1, 0)]}


INCEPTION_B = {"branch3x3": [("branch3x3", 384, 3, 2, 0)],
               "branch3x3dbl": [("branch3x3dbl_1", 64, 1, 1, 0), ("branch3x3dbl_2", 96, 3, 1, 1), ("branch3x3dbl_3", 96, 3, 2, 0)],
               "branch_pool": [("max", 3, 2, 0)]}
INCEPTION_D = {"branch3x3": [("branch3x3_1", 192, 1, 1, 0), ("branch3x3_2", 320, 3, 2, 0)],
               "branch7x7x3": [("branch7x7x3_1", 192, 1, 1, 0), ("branch7x7x3_2", 192, (1, 7), 1, (0, 3)),
                               ("branch7x7x3_3", 192, (7, 1), 1, (3, 0)), ("branch7x7x3_4", 192, 3, 2, 0)],
               "branch_pool": [("max", 3, 2, 0)]}
# InceptionE: two of its branches end in a pair of parallel convs whose outputs are concatenated ("split")
INCEPTION_E = {"branch1x1": [("branch1x1", 320, 1, 1, 0)],
               "branch3x3": [("branch3x3_1", 384, 1, 1, 0), ("split", ("branch3x3_2a", 384, (1, 3), 1, (0, 1)), ("branch3x3_2b", 384, (3, 1), 1, (1, 0)))],
               "branch3x3dbl": [("branch3x3dbl_1", 448, 1, 1, 0), ("branch3x3dbl_2", 384, 3, 1, 1),
                                ("split", ("branch3x3dbl_3a", 384, (1, 3), 1, (0, 1)), ("branch3x3dbl_3b", 384, (3, 1), 1, (1, 0)))],
               "branch_pool": [("avg", 3, 1, 1), ("branch_pool", 192, 1, 1, 0)]}

STEM = [("Conv2d_1a_3x3", 32, 3, 2, 0), ("Conv2d_2a_3x3", 32, 3, 1, 0), ("Conv2d_2b_3x3", 64, 3, 1, 1), ("max", 3, 2, 0),
        ("Conv2d_3b_1x1", 80, 1, 1, 0), ("Conv2d_4a_3x3", 192, 3, 1, 0), ("max", 3, 2, 0)]
BLOCKS = [("Mixed_5b", inception_a(32)), ("Mixed_5c", inception_a(64)), ("Mixed_5d", inception_a(64)), ("Mixed_6a", INCEPTION_B),
          ("Mixed_6b", inception_c(128)), ("Mixed_6c", inception_c(160)), ("Mixed_6d", inception_c(160)), ("Mixed_6e", inception_c(192)),
          ("Mixed_7a", INCEPTION_D), ("Mixed_7b", INCEPTION_E), ("Mixed_7c", INCEPTION_E)]


def conv_specs():
    """Every BasicConv2d of the network as (state-dict prefix, in_channels, out_channels, kernel, stride, padding) in forward
    order - the single description the HIP chain, the random-weight generator of the tests and the CPU restatement share."""
    out, c = [], 3
    for op in STEM:
        if op[0] in ("max", "avg"):
            continue
        out.append((op[0], c, op[1], _pair(op[2]), _pair(op[3]), _pair(op[4])))
        c = op[1]
    for bname, branches in BLOCKS:
        cin, total = c, 0
        for ops in branches.values():
            cb = cin
            for op in ops:
                if op[0] in ("max", "avg"):
                    continue
                if op[0] == "split":
                    for sub in op[1:]:
                        out.append((f"{bname}.{sub[0]}", cb, sub[1], _pair(sub[2]), _pair(sub[3]), _pair(sub[4])))
                    cb = sum(sub[1] for sub in op[1:])
                else:
                    out.append((f"{bname}.{op[0]}", cb, op[1], _pair(op[2]), _pair(op[3]), _pair(op[4])))
                    cb = op[1]
            total += cb
        c = total
    return out


class InceptionV3Hip:
    """logits = net(images): images NCHW fp32 [N,3,299,299] on the GPU (ImageNet-normalised) -> [N, num_classes] fp32."""

    def __init__(self, state_dict, device="cuda", chunk=64):
        if not torch.cuda.is_available():
            raise JckError("InceptionV3Hip needs a GPU: the HIP path has no CPU fallback")
        self.device = torch.device(device)
        self.chunk = chunk                 # images per pass (bounds the activation memory: ~1.3 GB at 64)
        self.w = {}
        sd = {k: v for k, v in state_dict.items() if not k.startswith("AuxLogits.")}
        for name, cin, cout, k, s, p in conv_specs():
            w = sd[f"{name}.conv.weight"].detach().float()
            if tuple(w.shape) != (cout, cin, k[0], k[1]):
                raise JckError(f"{name}.conv.weight has shape {tuple(w.shape)}, expected {(cout, cin, k[0], k[1])}")
            g, b = sd[f"{name}.bn.weight"].detach().double(), sd[f"{name}.bn.bias"].detach().double()
            rm, rv = sd[f"{name}.bn.running_mean"].detach().double(), sd[f"{name}.bn.running_var"].detach().double()
            scale = g / torch.sqrt(rv + BN_EPS)
            # [co][ci][kh][kw] -> [(kh, kw, ci)][co]: k-major rows, output channels contiguous
            wk = w.permute(2, 3, 1, 0).reshape(k[0] * k[1] * cin, cout).contiguous()
            self.w[name] = (wk.to(self.device), scale.float().to(self.device), (b - rm * scale).float().to(self.device), cin, cout, k, s, p)
        fcw = sd["fc.0.weight"] if "fc.0.weight" in sd else sd["fc.weight"]
        fcb = sd["fc.0.bias"] if "fc.0.bias" in sd else sd["fc.bias"]
        self.num_classes = fcw.shape[0]
        self.fc_w = fcw.detach().float().t().contiguous().to(self.device)          # [2048][classes] = [k][co]
        self.fc_b = fcb.detach().float().contiguous().to(self.device)

    @classmethod
    def from_file(cls, path, device="cuda", **kw):
        """Local-weights loader: a `torch.save`d state dict (or a checkpoint holding one under 'state_dict' / 'model')."""
        obj = torch.load(path, map_location="cpu", weights_only=False)
        if isinstance(obj, dict) and not any(k.endswith("conv.weight") for k in obj):
            for key in ("state_dict", "model", "model_state_dict"):
                if key in obj:
                    obj = obj[key]
                    break
        if not isinstance(obj, dict):
            obj = obj.state_dict()
        return cls({k.replace("module.", "", 1) if k.startswith("module.") else k: v for k, v in obj.items()}, device, **kw)

    def eval(self):
        return self

    def to(self, device):
        return self

    # ---- kernels ---------------------------------------------------------------------------------------------
    def _conv(self, name, x, out=None, coff=0):
        wk, sc, sh, cin, cout, k, s, p = self.w[name]
        n, h, w_, c = x.shape
        if c != cin:
            raise JckError(f"{name}: input has {c} channels, expected {cin}")
        oh, ow = (h + 2 * p[0] - k[0]) // s[0] + 1, (w_ + 2 * p[1] - k[1]) // s[1] + 1
        if out is None:
            out = torch.empty(n, oh, ow, cout, dtype=torch.float32, device=x.device)
        lib.jck_conv2d_nhwc_f32(x, wk, sc, sh, out, n, h, w_, cin, k[0], k[1], s[0], s[1], p[0], p[1], cout, out.shape[3], coff, 1,
                                cur_stream())
        return out

    def _pool(self, x, k, s, p, mode, out=None, coff=0):
        n, h, w_, c = x.shape
        oh, ow = (h + 2 * p - k) // s + 1, (w_ + 2 * p - k) // s + 1
        if out is None:
            out = torch.empty(n, oh, ow, c, dtype=torch.float32, device=x.device)
        lib.jck_pool2d_nhwc_f32(x, out, n, h, w_, c, k, s, p, mode, out.shape[3], coff, cur_stream())
        return out

    @staticmethod
    def _out_hw(h, w_, ops):
        for op in ops:
            if op[0] == "split":
                continue
            k, s, p = (_pair(op[1]), _pair(op[2]), _pair(op[3])) if op[0] in ("max", "avg") else (_pair(op[2]), _pair(op[3]), _pair(op[4]))
            h, w_ = (h + 2 * p[0] - k[0]) // s[0] + 1, (w_ + 2 * p[1] - k[1]) // s[1] + 1
        return h, w_

    def _block(self, bname, branches, x):
        n, h, w_, cin = x.shape
        widths = []
        for ops in branches.values():
            cb = cin
            for op in ops:
                if op[0] == "split":
                    cb = sum(sub[1] for sub in op[1:])
                elif op[0] not in ("max", "avg"):
                    cb = op[1]
            widths.append(cb)
        oh, ow = self._out_hw(h, w_, next(iter(branches.values())))
        out = torch.empty(n, oh, ow, sum(widths), dtype=torch.float32, device=x.device)
        off = 0
        for ops, width in zip(branches.values(), widths):
            t = x
            for i, op in enumerate(ops):
                last = i == len(ops) - 1
                if op[0] in ("max", "avg"):
                    t = self._pool(t, op[1], op[2], op[3], 0 if op[0] == "max" else 1, out if last else None, off if last else 0)
                elif op[0] == "split":                     # two parallel convs on t, concatenated (always the end of a branch)
                    o = off
                    for sub in op[1:]:
                        self._conv(f"{bname}.{sub[0]}", t, out, o)
                        o += sub[1]
                else:
                    t = self._conv(f"{bname}.{op[0]}", t, out if last else None, off if last else 0)
            off += width
        return out

    def _forward_chunk(self, x_nchw):
        n = x_nchw.shape[0]
        x = torch.empty(n, 299, 299, 3, dtype=torch.float32, device=x_nchw.device)
        lib.jck_nchw_to_nhwc_f32(x_nchw, x, n, 3, 299, 299, cur_stream())
        for op in STEM:
            x = self._pool(x, op[1], op[2], op[3], 0) if op[0] == "max" else self._conv(op[0], x)
        for bname, branches in BLOCKS:
            x = self._block(bname, branches, x)
        n, h, w_, c = x.shape
        pooled = torch.empty(n, 1, 1, c, dtype=torch.float32, device=x.device)        # adaptive_avg_pool2d((1,1)); dropout: eval no-op
        lib.jck_global_avgpool_nhwc_f32(x, pooled, n, h * w_, c, cur_stream())
        logits = torch.empty(n, 1, 1, self.num_classes, dtype=torch.float32, device=x.device)
        lib.jck_conv2d_nhwc_f32(pooled, self.fc_w, None, self.fc_b, logits, n, 1, 1, c, 1, 1, 1, 1, 0, 0, self.num_classes,
                                self.num_classes, 0, 0, cur_stream())
        return logits.view(n, self.num_classes)

    def __call__(self, images):
        if not images.is_cuda:
            raise JckError("InceptionV3Hip: got a CPU tensor - the MI355X path has no CPU fallback")
        if images.dim() != 4 or tuple(images.shape[1:]) != (3, 299, 299):
            raise JckError(f"InceptionV3Hip expects [N,3,299,299], got {tuple(images.shape)}")
        images = images.to(torch.float32).contiguous()
        return torch.cat([self._forward_chunk(images[i:i + self.chunk]) for i in range(0, images.shape[0], self.chunk)])
