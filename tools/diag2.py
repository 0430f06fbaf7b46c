import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/jck-generation_amd'); sys.path.insert(0, '/root/repo/tests')
import torch, torch.nn.functional as F
from model import DCGAN
from hipgan import functional as HF
from util import synth_images
torch.manual_seed(12345)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = DCGAN.Discriminator(); d.apply(DCGAN.weights_init)
x = synth_images(B, seed=5)
# --- CPU reference with captured activation gradients
ref = {}
def cap(name):
    def h(g): ref[name] = g.detach().clone()
    return h
sd = {k: v.clone() for k, v in d.state_dict().items()}
ps = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and 'running' not in k}
h = x.clone().requires_grad_(True); xin = h
acts = {}
for i in (1, 2, 3, 4):
    y = F.conv2d(h, ps[f'conv{i}.weight'], None, 2, 1); y.register_hook(cap(f'gy{i}')); acts[f'y{i}'] = y.detach()
    bn = F.batch_norm(y, None, None, ps[f'norm{i}.weight'], ps[f'norm{i}.bias'], True, 0.1, 1e-5)
    h = F.leaky_relu(bn, 0.2); h.register_hook(cap(f'ga{i}')); acts[f'a{i}'] = h.detach()
p = torch.sigmoid(F.conv2d(h, ps['conv5.weight'])).view(-1)
loss = F.binary_cross_entropy(p, torch.full((B,), 0.9)); loss.backward()
ref['gx'] = xin.grad
# --- GPU path through the autograd Functions, hooks on stage outputs
for prec in ('f32',):
    dg = DCGAN.Discriminator(); dg.load_state_dict(sd); dg = dg.cuda(); dg.prec = prec
    got = {}
    P = HF._PREC[prec]
    xg = x.cuda().requires_grad_(True)
    hh = HF._ToNHWC.apply(xg, P)
    for i in (1, 2, 3, 4):
        hh = HF._stage(dg, i, hh, 'down', 0.2, P)
        hh.register_hook((lambda n: (lambda g: got.__setitem__(n, g.detach().float().cpu().permute(0, 3, 1, 2))))(f'ga{i}'))
        a = hh.detach().float().cpu().permute(0, 3, 1, 2)
        print(f'fwd a{i}: maxerr/max {((a-acts[f"a{i}"]).abs().max()/acts[f"a{i}"].abs().max()):.3e}')
    pg = HF._HeadSigmoid.apply(hh, dg.conv5.weight, P).view(-1)
    lg = F.binary_cross_entropy(pg, torch.full((B,), 0.9, device='cuda')); lg.backward()
    print(prec, 'loss', float(lg), float(loss))
    for i in (4, 3, 2, 1):
        r, g = ref[f'ga{i}'], got[f'ga{i}']
        e = (g - r)
        print(f'  ga{i}: max|ref| {r.abs().max():.3e} maxerr/max {e.abs().max()/r.abs().max():.3e} rel-l2 {e.norm()/r.norm():.3e}  per-channel mean err / mean|ref|: {(e.mean((0,2,3)).abs().max()/r.abs().mean()):.3e}')
        # where are the errors? border vs interior
        H = r.shape[-1]
        m = torch.zeros(H, H, dtype=torch.bool); m[0, :] = m[-1, :] = m[:, 0] = m[:, -1] = True
        print(f'       border maxerr {e[..., m].abs().max():.3e} interior maxerr {e[..., ~m].abs().max():.3e}')
    gx = xg.grad.cpu(); e = gx - ref['gx']
    print(f'  gx: maxerr/max {e.abs().max()/ref["gx"].abs().max():.3e} rel-l2 {e.norm()/ref["gx"].norm():.3e}')
    for k, v in dg.named_parameters():
        r = ps[k].grad; e = v.grad.cpu() - r
        print(f'  {k:14s} maxerr/max {e.abs().max()/r.abs().max():.3e} rel-l2 {e.norm()/r.norm():.3e}')
