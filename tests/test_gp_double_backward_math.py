"""The CGAN gradient penalty is back-propagated (train/cgan_trainer.py:200-203): a double backward through conv,
train-mode BatchNorm, LeakyReLU, Linear, Dropout and Sigmoid.  This file holds the closed-form reverse pass that the
HIP path implements (csrc/engine.hip, family 1) and checks it against autograd(create_graph=True) in fp64 on a small
net of the same structure.  CPU only."""
import torch, torch.nn.functional as F
torch.manual_seed(0)
dt = torch.float64
B, L = 4, 2
chans = [3, 8, 16]
H = 16
lam = 10.0
Ws = [torch.randn(chans[i+1], chans[i], 4, 4, dtype=dt) * 0.1 for i in range(L)]
gam = [1 + 0.1 * torch.randn(chans[i+1], dtype=dt) for i in range(L)]
bet = [0.1 * torch.randn(chans[i+1], dtype=dt) for i in range(L)]
feat = chans[-1] * (H >> L) ** 2
E = 5
We, be = torch.randn(E, 6, dtype=dt) * 0.3, torch.randn(E, dtype=dt) * 0.1
W1, b1 = torch.randn(7, feat + E, dtype=dt) * 0.1, torch.randn(7, dtype=dt) * 0.1
W2, b2 = torch.randn(1, 7, dtype=dt) * 0.3, torch.randn(1, dtype=dt) * 0.1
params = Ws + gam + bet + [We, be, W1, b1, W2, b2]
for p in params: p.requires_grad_(True)
x = torch.randn(B, 3, H, H, dtype=dt)
lab = torch.randn(B, 6, dtype=dt)
mask = (torch.rand(B, 7) > 0.25).to(dt)
pdrop = 0.25
eps = 1e-5

# ---- autograd reference
xi = x.clone().requires_grad_(True)
h = xi
for i in range(L):
    h = F.conv2d(h, Ws[i], None, 2, 1)
    h = F.batch_norm(h, None, None, gam[i], bet[i], True, 0.1, eps)
    h = F.leaky_relu(h, 0.2)
e = F.leaky_relu(F.linear(lab, We, be), 0.2)
c = torch.cat([h.flatten(1), e], 1)
hh = F.linear(c, W1, b1) * mask / (1 - pdrop)
p = torch.sigmoid(F.linear(hh, W2, b2))
gx = torch.autograd.grad(p, xi, torch.ones_like(p), create_graph=True)[0]
gp = ((gx.view(B, -1).norm(2, dim=1) - 1) ** 2).mean()
ref = torch.autograd.grad(lam * gp, params, allow_unused=True)

# ---- manual
with torch.no_grad():
    n = lambda t: t.shape[0] * t.shape[2] * t.shape[3]
    cm = lambda t: t.mean((0, 2, 3), keepdim=True)
    cs = lambda t: t.sum((0, 2, 3))
    a = [x]; ys = []; xh = []; sig = []; s = []
    for i in range(L):
        y = F.conv2d(a[-1], Ws[i], None, 2, 1); ys.append(y)
        mu = cm(y); var = cm((y - mu) ** 2); sg = torch.sqrt(var + eps); sig.append(sg)
        xhat = (y - mu) / sg; xh.append(xhat)
        z = gam[i].view(1, -1, 1, 1) * xhat + bet[i].view(1, -1, 1, 1)
        s.append(torch.where(z > 0, torch.ones_like(z), torch.full_like(z, 0.2)))
        a.append(torch.where(z > 0, z, 0.2 * z))
    le = F.linear(lab, We, be); se = torch.where(le > 0, torch.ones_like(le), torch.full_like(le, 0.2)); e = le * se
    f = a[-1].flatten(1); c = torch.cat([f, e], 1)
    hpre = F.linear(c, W1, b1); dm = mask / (1 - pdrop); hd = hpre * dm
    sl = F.linear(hd, W2, b2); p = torch.sigmoid(sl)
    # B1: first backward
    ds = p * (1 - p)                                   # [B,1]
    ghd = ds @ W2                                      # [B,7]
    gh = ghd * dm
    gc = gh @ W1                                       # [B, feat+E]
    ga = gc[:, :feat].view_as(a[-1])
    gz = [None] * L; gy = [None] * L; m2 = [None] * L
    for i in reversed(range(L)):
        gz[i] = ga * s[i]
        m1 = cm(gz[i]); m2[i] = cm(gz[i] * xh[i])
        gy[i] = gam[i].view(1, -1, 1, 1) / sig[i] * (gz[i] - m1 - xh[i] * m2[i])
        ga = F.conv_transpose2d(gy[i], Ws[i], None, 2, 1)
    gx_m = ga
    nrm = gx_m.view(B, -1).norm(2, dim=1)
    u = (lam * 2.0 / B * (nrm - 1) / nrm).view(B, 1, 1, 1) * gx_m          # dL/dgx
    G = {id(p_): torch.zeros_like(p_) for p_ in params}
    # reverse of B1 (a forward-like sweep)
    xdir = [None] * L; sigexp = [None] * L
    ui = u
    for i in range(L):
        # g_{a_{i-1}} = convT(gy_i; W_i)  ->  adjoint: v_i = conv(u_{i-1}; W_i), dW_i += wgrad(gy_i, u_{i-1})
        v = F.conv2d(ui, Ws[i], None, 2, 1)
        G[id(Ws[i])] += torch.nn.grad.conv2d_weight(ui, Ws[i].shape, gy[i], 2, 1)
        gs = gam[i].view(1, -1, 1, 1) / sig[i]
        G[id(gam[i])] += cs(v * gy[i]) / gam[i]
        mvx = cm(v * xh[i])
        xdir[i] = -gs * (v * m2[i] + gz[i] * mvx)
        sigexp[i] = -(cs(v * gy[i]).view(1, -1, 1, 1) / sig[i])           # adjoint of sigma (explicit 1/sigma factor)
        ugz = gs * (v - cm(v) - xh[i] * mvx)
        ui = ugz * s[i]
    u4 = ui.flatten(1)                                                     # adjoint of gc[:, :feat]
    ugc = torch.cat([u4, torch.zeros(B, E, dtype=dt)], 1)
    ugh = ugc @ W1.t()
    G[id(W1)] += gh.t() @ ugc
    ughd = ugh * dm
    uds = ughd @ W2.t()                                                    # [B,1]
    G[id(W2)] += ds.t() @ ughd
    rs = uds * (1 - 2 * p) * p * (1 - p)                                   # adjoint at the logit
    # reverse of the forward pass with the injected adjoints
    G[id(W2)] += rs.t() @ hd; G[id(b2)] += rs.sum(0)
    uhd = rs @ W2; uh = uhd * dm
    G[id(W1)] += uh.t() @ c; G[id(b1)] += uh.sum(0)
    uc = uh @ W1
    ue = uc[:, feat:] * se
    G[id(We)] += ue.t() @ lab; G[id(be)] += ue.sum(0)
    ua = uc[:, :feat].view_as(a[-1])
    for i in reversed(range(L)):
        uz = ua * s[i]
        G[id(gam[i])] += cs(uz * xh[i]); G[id(bet[i])] += cs(uz)
        q = gam[i].view(1, -1, 1, 1) * uz + xdir[i]
        uy = (q - cm(q) - xh[i] * cm(q * xh[i])) / sig[i] + sigexp[i] * xh[i] / n(q)
        G[id(Ws[i])] += torch.nn.grad.conv2d_weight(a[i], Ws[i].shape, uy, 2, 1)
        ua = F.conv_transpose2d(uy, Ws[i], None, 2, 1)
names = [f"W{i}" for i in range(L)] + [f"gam{i}" for i in range(L)] + [f"bet{i}" for i in range(L)] + ["We", "be", "W1", "b1", "W2", "b2"]


def test_manual_double_backward_matches_autograd():
    for nm, p_, r in zip(names, params, ref):
        r = torch.zeros_like(p_) if r is None else r
        assert float((G[id(p_)] - r).abs().max()) < 1e-12 * max(1.0, float(r.abs().max())), nm
    assert float((gx_m - gx.detach()).abs().max()) < 1e-14
