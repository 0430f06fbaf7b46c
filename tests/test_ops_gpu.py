"""Per-op parity of the HIP kernels (through the C ABI) against plain PyTorch fp32 CPU ops of the same
function (the library the reference's hot path is made of).  Tolerances are relative to max|ref|:
JCK_PREC_F32 (exact fp32 MFMA v_mfma_f32_16x16x4_f32, fp32 storage) 3e-6;  JCK_PREC_BF16 (bf16 storage) 1.5e-2 against a
reference computed from bf16-rounded inputs."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_util
    return gpu_util


PRECS = [1, 0]


def test_tr_read_semantics(G):
    """ds_read_b64_tr_b16 as wgrad.hpp addresses it: lane l gets in[8*(l>>4)+j][l&15], j=0..7."""
    ld = 48
    ids = torch.arange(32 * ld, dtype=torch.int16).cuda()
    out = torch.zeros(64 * 8, dtype=torch.int16).cuda()
    G.lib.jck_debug_tr_read(ids, ld, out, G.cur_stream())
    torch.cuda.synchronize()
    got = out.cpu().view(64, 8)
    exp = torch.tensor([[(8 * (l >> 4) + j) * ld + (l & 15) for j in range(8)] for l in range(64)], dtype=torch.int16)
    if not torch.equal(got, exp):
        rows, cols = got // ld, got % ld
        raise AssertionError("tr-read mapping differs; (row,col) per lane:\n" +
                             "\n".join(f"lane {l}: {[(int(rows[l, j]), int(cols[l, j])) for j in range(8)]}" for l in range(64)))


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("shape", [(2, 8, 64, 128), (3, 64, 3, 64), (2, 16, 128, 256), (4, 8, 256, 512), (1, 4, 64, 128),
                                   (5, 32, 64, 128), (3, 32, 3, 64), (2, 128, 3, 64)])
def test_conv_down(G, prec, shape):
    n, hb, cb, cs = shape
    g = torch.Generator().manual_seed(1)
    x = G.rnd(torch.randn(n, cb, hb, hb, generator=g), prec)
    w = torch.randn(cs, cb, 4, 4, generator=g) * 0.05
    wr = G.rnd(w, prec)
    ref = F.conv2d(x, wr, None, 2, 1)
    out = torch.empty(n, hb // 2, hb // 2, cs, dtype=G.DT[prec], device="cuda")
    stats, slots = G.stats_buf(n * (hb // 2) ** 2, cs)
    G.lib.jck_conv_down(prec, G.to_nhwc(x, prec), G.pack_down(w, prec), out, stats, ctypes.byref(slots), n, hb, hb, cb, cs,
                        G.cur_stream())
    torch.cuda.synchronize()
    G.check(G.from_nhwc(out), ref, G.TOL[prec], "conv_down")
    ssum, ssq = G.stats_sum(stats, slots, cs)
    G.check(ssum, ref.sum((0, 2, 3)), 2e-3 if prec == 0 else 1e-5, "stats sum")
    G.check(ssq, (ref * ref).sum((0, 2, 3)), 2e-3 if prec == 0 else 1e-5, "stats sumsq")


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("shape", [(2, 4, 512, 256), (2, 8, 256, 128), (3, 16, 128, 64), (2, 32, 64, 3), (5, 4, 128, 64),
                                   (3, 16, 64, 3), (1, 64, 64, 3)])
def test_conv_up(G, prec, shape):
    n, hs, cs, cb = shape
    g = torch.Generator().manual_seed(2)
    x = G.rnd(torch.randn(n, cs, hs, hs, generator=g), prec)
    w = torch.randn(cs, cb, 4, 4, generator=g) * 0.05
    wr = G.rnd(w, prec)
    tanh = cb == 3
    ref = F.conv_transpose2d(x, wr, None, 2, 1)
    cbp = G.lib.jck_pad_chan(cb)
    out = torch.full((n, 2 * hs, 2 * hs, cbp), 7.0, dtype=G.DT[prec], device="cuda")
    stats, slots = G.stats_buf(n * 4 * hs * hs, cbp)
    G.lib.jck_conv_up(prec, G.to_nhwc(x, prec), G.pack_up(w, prec), out, None if tanh else stats,
                      None if tanh else ctypes.byref(slots), 1 if tanh else 0, n, hs, hs, cs, cb, G.cur_stream())
    torch.cuda.synchronize()
    G.check(G.from_nhwc(out, cb), torch.tanh(ref) if tanh else ref, G.TOL[prec] if not tanh else max(G.TOL[prec], 1e-6),
            "conv_up")
    if cbp != cb:
        assert float(out[..., cb:].float().abs().max()) == 0.0, "padding channel must be zero"
    if not tanh:
        ssum, ssq = G.stats_sum(stats, slots, cbp)
        G.check(ssum[:cb], ref.sum((0, 2, 3)), 2e-3 if prec == 0 else 1e-5, "stats sum")
        G.check(ssq[:cb], (ref * ref).sum((0, 2, 3)), 2e-3 if prec == 0 else 1e-5, "stats sumsq")


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("shape", [(2, 8, 64, 128), (3, 64, 3, 64), (2, 16, 128, 256), (4, 8, 256, 512), (7, 8, 64, 128),
                                   (16, 32, 64, 128), (3, 64, 64, 128), (9, 16, 32, 64), (5, 8, 32, 192), (21, 8, 64, 64)])
def test_conv_wgrad(G, prec, shape):
    """dW of Conv2d(big->small) == dW of ConvTranspose2d(small->big): one kernel, checked against both (ragged image
    counts, split-K over several workgroups)."""
    n, hb, cb, cs = shape
    g = torch.Generator().manual_seed(3)
    big = G.rnd(torch.randn(n, cb, hb, hb, generator=g), prec)
    small = G.rnd(torch.randn(n, cs, hb // 2, hb // 2, generator=g), prec)
    w = torch.zeros(cs, cb, 4, 4, requires_grad=True)
    (F.conv2d(big, w, None, 2, 1) * small).sum().backward()
    ref = w.grad.clone()
    w.grad = None
    (F.conv_transpose2d(small, w, None, 2, 1) * big).sum().backward()
    torch.testing.assert_close(w.grad, ref, rtol=1e-3, atol=1e-3 * ref.abs().max().item())
    ws_bytes = G.lib.jck_conv_wgrad_ws_bytes(n, hb, hb, cb, cs)
    ws = torch.empty(ws_bytes // 4, device="cuda")
    grad = torch.full((cs, cb, 4, 4), 1.0, device="cuda")
    G.lib.jck_conv_wgrad(prec, G.to_nhwc(small, prec), G.to_nhwc(big, prec), ws, ws_bytes, grad, 1, n, hb, hb, cb, cs,
                         G.cur_stream())
    torch.cuda.synchronize()
    G.check(grad.cpu() - 1.0, ref, 1e-5 if prec == 1 else 2e-3, "conv_wgrad(accumulate)")
    G.lib.jck_conv_wgrad(prec, G.to_nhwc(small, prec), G.to_nhwc(big, prec), ws, ws_bytes, grad, 0, n, hb, hb, cb, cs,
                         G.cur_stream())
    torch.cuda.synchronize()
    G.check(grad.cpu(), ref, 3e-6 if prec == 1 else 2e-3, "conv_wgrad(overwrite)")


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("cfg", [(8, 100, 128, 512), (256, 100, 128, 512), (5, 200, 256, 512)])
def test_g1(G, prec, cfg):
    b, ci, cip, co = cfg
    g = torch.Generator().manual_seed(4)
    z = G.rnd(torch.randn(b, ci, 1, 1, generator=g), prec)
    w = torch.randn(ci, co, 4, 4, generator=g) * 0.05
    w.requires_grad_(True)
    wr = G.rnd(w.detach(), prec)
    ref = F.conv_transpose2d(z, wr, None, 1, 0)
    zp = torch.zeros(b, cip)
    zp[:, :ci] = z.view(b, ci)
    zp = zp.to(G.DT[prec]).cuda()
    wp = torch.empty(16 * co * cip, dtype=G.DT[prec], device="cuda")
    G.lib.jck_pack_g1(prec, w.detach().cuda(), ci, co, cip, wp, G.cur_stream())
    out = torch.empty(b, 4, 4, co, dtype=G.DT[prec], device="cuda")
    stats, slots = G.stats_buf(b * 16, co, 16)
    G.lib.jck_g1_fwd(prec, zp, wp, out, stats, ctypes.byref(slots), b, cip, co, G.cur_stream())
    torch.cuda.synchronize()
    G.check(G.from_nhwc(out), ref, G.TOL[prec], "g1_fwd")
    ssum, ssq = G.stats_sum(stats, slots, co)
    G.check(ssum, ref.sum((0, 2, 3)), 2e-3 if prec == 0 else 1e-5, "g1 stats sum")
    G.check(ssq, (ref * ref).sum((0, 2, 3)), 2e-3 if prec == 0 else 1e-5, "g1 stats sumsq")
    dy = G.rnd(torch.randn(b, co, 4, 4, generator=g), prec)
    (F.conv_transpose2d(z, w, None, 1, 0) * dy).sum().backward()
    ws_bytes = G.lib.jck_g1_wgrad_ws_bytes(b, cip, co)
    ws = torch.empty(ws_bytes // 4, device="cuda")
    grad = torch.zeros(ci, co, 4, 4, device="cuda")
    G.lib.jck_g1_wgrad(prec, zp, G.to_nhwc(dy, prec), ws, ws_bytes, grad, 0, b, ci, cip, co, G.cur_stream())
    torch.cuda.synchronize()
    G.check(grad.cpu(), w.grad, 3e-6 if prec == 1 else 2e-3, "g1_wgrad")


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("cfg", [(2, 8, 64, 0.2), (3, 4, 512, 0.0), (4, 16, 128, 0.2), (2, 32, 64, 0.0)])
def test_bn_act(G, prec, cfg):
    n, h, c, slope = cfg
    g = torch.Generator().manual_seed(5)
    y = G.rnd(torch.randn(n, c, h, h, generator=g) * 1.5 + 0.3, prec).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(c, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(c, generator=g)).requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    bn = F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5)
    a = F.leaky_relu(bn, slope) if slope else F.relu(bn)
    ga = G.rnd(torch.randn(n, c, h, h, generator=g), prec)
    a.backward(ga)
    rows = n * h * h
    yd = G.to_nhwc(y.detach(), prec)
    yf = yd.float().view(rows, c)
    half = rows // 2        # three partial slots: two halves and an all-zero one
    stats = torch.stack([torch.stack([yf[:half].sum(0), (yf[:half] ** 2).sum(0)]),
                         torch.stack([yf[half:].sum(0), (yf[half:] ** 2).sum(0)]), torch.zeros(2, c, device="cuda")]).contiguous()
    aux = torch.empty(4 * c, device="cuda")
    rmd, rvd = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
    G.lib.jck_bn_finalize(stats, 3, float(rows), gamma.detach().cuda(), beta.detach().cuda(), rmd, rvd, nbt, 0.1, 1e-5, aux, c,
                          G.cur_stream())
    ad = torch.empty_like(yd)
    G.lib.jck_bn_act_fwd(prec, yd, aux, slope, ad, rows, c, G.cur_stream())
    sums = torch.full((G.lib.jck_bn_bwd_ws_floats(c),), float("nan"), device="cuda")
    gy = torch.empty_like(yd)
    dgam, dbet = torch.ones(c, device="cuda"), torch.ones(c, device="cuda")
    G.lib.jck_bn_act_bwd(prec, G.to_nhwc(ga, prec), yd, aux, slope, sums, gy, dgam, dbet, rows, c, G.cur_stream())
    torch.cuda.synchronize()
    tol = 1e-5 if prec == 1 else 1.5e-2
    G.check(G.from_nhwc(ad), a.detach(), tol, "bn_act_fwd")
    G.check(rmd.cpu(), rm, 1e-5, "running_mean")
    G.check(rvd.cpu(), rv, 1e-5, "running_var")
    assert int(nbt) == 1
    G.check(G.from_nhwc(gy), y.grad, 2e-5 if prec == 1 else 2e-2, "bn_act_bwd dx")
    G.check(dgam.cpu() - 1, gamma.grad, 2e-5 if prec == 1 else 1e-2, "dgamma")
    G.check(dbet.cpu() - 1, beta.grad, 2e-5 if prec == 1 else 1e-2, "dbeta")


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("cfg", [(1, 700, 64, 0.2, 3), (3, 4096 + 5, 128, 0.0, 37), (2, 1024, 512, 0.2, 256), (1, 64, 256, 0.2, 1),
                                 (2, 300, 128, 0.2, 400)])          # the last: more rows than the fused form takes - the two launches
def test_bn_fwd_one_launch(G, prec, cfg):
    """jck_bn_fwd (finalize + apply as one launch: every workgroup sums the statistics rows of its 64-channel slice itself) against
    aten::native_batch_norm + the activation (F.batch_norm in training mode on the CPU) group by group, and against the two launches
    it replaces (jck_bn_finalize_grouped + jck_bn_act_fwd_grouped): aux table, deferred running-statistics record, output."""
    groups, rows, c, slope, slots = cfg
    g = torch.Generator().manual_seed(11)
    y = G.rnd(torch.randn(groups, rows, c, generator=g) * 1.5 + 0.3, prec)
    gamma, beta = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    yd = y.to(G.DT[prec]).cuda().contiguous()
    # statistics rows: the group's rows dealt over `slots` partial sums (some empty), as a producer would leave them
    stats = torch.zeros(groups, slots, 2, c, device="cuda")
    owner = torch.randint(0, slots, (rows,), generator=g).cuda()
    for k in range(groups):
        yf = yd[k].float()
        stats[k, :, 0].index_add_(0, owner, yf)
        stats[k, :, 1].index_add_(0, owner, yf * yf)
    gam, bet = gamma.cuda(), beta.cuda()
    a1, aux1, rec1 = torch.empty_like(yd), torch.empty(groups, 4 * c, device="cuda"), torch.empty(groups, 2 * c, device="cuda")
    G.lib.jck_bn_fwd(prec, yd, stats, slots, float(rows), gam, bet, 1e-5, slope, a1, aux1, rec1, None, None, None, 0.1, rows, c, groups, G.cur_stream())
    a2, aux2, rec2 = torch.empty_like(yd), torch.empty(groups, 4 * c, device="cuda"), torch.empty(groups, 2 * c, device="cuda")
    G.lib.jck_bn_finalize_grouped(stats, slots, float(rows), gam, bet, 1e-5, aux2, rec2, c, groups, G.cur_stream())
    G.lib.jck_bn_act_fwd_grouped(prec, yd, aux2, slope, a2, rows, c, groups, G.cur_stream())
    torch.cuda.synchronize()
    G.check(aux1.cpu(), aux2.cpu(), 2e-6, "aux vs the finalize launch")
    G.check(rec1.cpu(), rec2.cpu(), 2e-6, "record vs the finalize launch")
    G.check(a1.float().cpu(), a2.float().cpu(), 1e-6 if prec == 1 else 8e-3, "output vs the two launches")       # <= one bf16 ulp of the largest value
    for k in range(groups):
        yk = y[k].t().contiguous().unsqueeze(0)                          # [1, C, rows]
        ref = F.batch_norm(yk, None, None, gamma, beta, True, 0.1, 1e-5)
        ref = (F.leaky_relu(ref, slope) if slope else F.relu(ref))[0].t()
        G.check(a1[k].float().cpu(), ref, 1e-5 if prec == 1 else 1.5e-2, f"group {k} vs F.batch_norm")
    if groups == 1:                                                      # the in-place running statistics of the one-group form (G's layers)
        rm, rv, nbt = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda")
        G.lib.jck_bn_fwd(prec, yd, stats, slots, float(rows), gam, bet, 1e-5, slope, a1, aux1, None, rm, rv, nbt, 0.1, rows, c, 1, G.cur_stream())
        rmr, rvr = torch.zeros(c), torch.ones(c)
        F.batch_norm(y[0].t().contiguous().unsqueeze(0), rmr, rvr, gamma, beta, True, 0.1, 1e-5)
        G.check(rm.cpu(), rmr, 1e-5, "running_mean")
        G.check(rv.cpu(), rvr, 1e-5, "running_var")
        assert int(nbt) == 1


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("cfg", [(1, 700, 64, 0.2, 1), (3, 4096 + 5, 128, 0.0, 2), (3, 3000, 512, 0.2, 3), (2, 64, 256, 0.2, 0)])
def test_bn_bwd_two_launches(G, prec, cfg):
    """jck_bn_act_bwd_grouped as two launches (reduce, then an apply whose workgroups sum the partial rows of their own 64-channel
    slice: bn_bwd_apply_fused_kernel) against the three launches it replaces (jck_tune bn_bwd_fuse 0: reduce, sums, apply) - g_y, the
    per-group sums CGAN's double backward reads, dgamma / dbeta over the first grad_groups groups - and against
    aten::native_batch_norm_backward + the activation's backward (autograd on the CPU), group by group."""
    groups, rows, c, slope, gg = cfg
    g = torch.Generator().manual_seed(23)
    y = G.rnd(torch.randn(groups, rows, c, generator=g) * 1.5 + 0.3, prec)
    ga = G.rnd(torch.randn(groups, rows, c, generator=g), prec)
    gamma, beta = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    yd, gad = y.to(G.DT[prec]).cuda().contiguous(), ga.to(G.DT[prec]).cuda().contiguous()
    stats = torch.stack([torch.stack([yd[k].float().sum(0), (yd[k].float() ** 2).sum(0)]) for k in range(groups)]).contiguous()
    aux = torch.empty(groups, 4 * c, device="cuda")
    G.lib.jck_bn_finalize_grouped(stats, 1, float(rows), gamma.cuda(), beta.cuda(), 1e-5, aux, None, c, groups, G.cur_stream())
    wsf = G.lib.jck_bn_bwd_ws_floats(c)
    out = {}
    for fuse in (1, 0):
        G.lib.jck_tune(b"bn_bwd_fuse", 2 * fuse)         # 2: also on the fp32 path (by default the fast path's form only)
        try:
            sums = torch.full((groups, wsf), float("nan"), device="cuda")
            gy = torch.empty_like(yd)
            dgam, dbet = torch.ones(c, device="cuda"), torch.ones(c, device="cuda")
            G.lib.jck_bn_act_bwd_grouped(prec, gad, yd, aux, slope, sums, gy, dgam if gg else None, dbet if gg else None, rows, c, groups, gg,
                                         G.cur_stream())
            torch.cuda.synchronize()
            out[fuse] = (gy.float().cpu(), sums[:, :2 * c].cpu(), dgam.cpu(), dbet.cpu())
        finally:
            G.lib.jck_tune(b"bn_bwd_fuse", 1)
    G.check(out[1][1], out[0][1], 2e-6, "sums vs the sums launch")
    G.check(out[1][2], out[0][2], 2e-6, "dgamma vs the sums launch")
    G.check(out[1][3], out[0][3], 2e-6, "dbeta vs the sums launch")
    G.check(out[1][0], out[0][0], 1e-6 if prec == 1 else 8e-3, "g_y vs the three launches")
    dg_ref, db_ref = torch.zeros(c), torch.zeros(c)
    for k in range(groups):
        yk = y[k].t().contiguous().unsqueeze(0).requires_grad_(True)                          # [1, C, rows]
        gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        bn = F.batch_norm(yk, None, None, gm, bt, True, 0.1, 1e-5)
        (F.leaky_relu(bn, slope) if slope else F.relu(bn)).backward(ga[k].t().contiguous().unsqueeze(0))
        G.check(out[1][0][k], yk.grad[0].t(), 2e-5 if prec == 1 else 2e-2, f"group {k} vs autograd")
        if k < gg:
            dg_ref += gm.grad
            db_ref += bt.grad
    if gg:
        G.check(out[1][2] - 1, dg_ref, 2e-5 if prec == 1 else 1e-2, "dgamma vs autograd")
        G.check(out[1][3] - 1, db_ref, 2e-5 if prec == 1 else 1e-2, "dbeta vs autograd")


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("cfg", [(4096 + 7, 64), (3000, 128), (700, 512)])
def test_bn2_chains_two_launches(G, prec, cfg):
    """The second-order BatchNorm steps of CGAN's back-propagated penalty (jck_bn2_vchain, jck_bn2_reverse; the closed form is held
    to fp64 autograd in tests/test_gp_double_backward_math.py and the step to the oracle in tests/test_cgan_gpu.py) as two launches -
    reduce, then an apply whose workgroups sum the partial rows of their own channel slice - against the three launches they replace
    (jck_tune bn_bwd_fuse 0): outputs, the sums the reverse sweep reads from the v-chain, the direct parameter gradients."""
    rows, c = cfg
    g = torch.Generator().manual_seed(41)
    rnd = lambda: G.rnd(torch.randn(rows, c, generator=g), prec).to(G.DT[prec]).cuda().contiguous()
    v, gy, ua = rnd(), rnd(), rnd()
    y = G.rnd(torch.randn(rows, c, generator=g) * 1.5 + 0.3, prec).to(G.DT[prec]).cuda().contiguous()
    gamma = (1 + 0.1 * torch.randn(c, generator=g)).cuda()
    beta = (0.1 * torch.randn(c, generator=g)).cuda()
    stats = torch.stack([y.float().sum(0), (y.float() ** 2).sum(0)]).unsqueeze(0).contiguous()
    aux = torch.empty(4 * c, device="cuda")
    G.lib.jck_bn_finalize_grouped(stats, 1, float(rows), gamma, beta, 1e-5, aux, None, c, 1, G.cur_stream())
    s1 = torch.randn(2 * c, generator=g).cuda() * rows ** 0.5        # sums of a first backward: any values of that size
    wsf = G.lib.jck_bn2_ws_floats(c)
    out = {}
    for fuse in (2, 0):
        G.lib.jck_tune(b"bn_bwd_fuse", fuse)
        try:
            ws1 = torch.full((wsf,), float("nan"), device="cuda")
            u, xdir = torch.empty_like(v), torch.empty_like(v)
            dg1 = torch.ones(c, device="cuda")
            G.lib.jck_bn2_vchain(prec, v, y, gy, aux, s1, gamma, 0.2, ws1, u, xdir, dg1, rows, c, G.cur_stream())
            ws2 = torch.full((wsf,), float("nan"), device="cuda")
            uy = torch.empty_like(v)
            dg2, db2 = torch.ones(c, device="cuda"), torch.ones(c, device="cuda")
            G.lib.jck_bn2_reverse(prec, ua, y, xdir, aux, gamma, ws1, 0.2, ws2, uy, dg2, db2, rows, c, G.cur_stream())
            torch.cuda.synchronize()
            out[fuse] = dict(u=u.float().cpu(), xdir=xdir.float().cpu(), uy=uy.float().cpu(), s3=ws1[:3 * c].cpu(), s4=ws2[:4 * c].cpu(),
                             dg1=dg1.cpu(), dg2=dg2.cpu(), db2=db2.cpu())
        finally:
            G.lib.jck_tune(b"bn_bwd_fuse", 1)
    for k in ("s3", "s4", "dg1", "dg2", "db2"):
        G.check(out[2][k], out[0][k], 3e-6, f"{k} vs the sums launch")
    for k in ("u", "xdir", "uy"):
        assert torch.isfinite(out[2][k]).all()
        G.check(out[2][k], out[0][k], 2e-6 if prec == 1 else 8e-3, f"{k} vs the three launches")


@pytest.mark.parametrize("B", [16, 300, 768])
def test_linear_fwd_split_k_on_the_lds_dma_kernel(G, B):
    """CGAN's Linear(8392,256) forward (model/CGAN.py:104) as a split-K plain GEMM: fp32 slabs [ksplit][B][N] from the LDS-DMA
    gather-GEMM (round 5) against the register-staged kernel it replaces there (jck_tune igemm_dma_ksplit 0) - same products in the
    same order per accumulator: the same bits - and, summed, against torch (bf16 operands, fp32 accumulation)."""
    prec, N, K, KP, KS = 0, 256, 8392, 8448, 12
    g = torch.Generator().manual_seed(31)
    x = (torch.randn(B, K, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.02)
    xp = torch.zeros(B, KP, dtype=torch.bfloat16)
    xp[:, :K] = x
    xd = xp.cuda()
    wp = torch.empty(N * KP, dtype=torch.bfloat16, device="cuda")
    G.lib.jck_pack_linear(prec, w.cuda(), N, K, N, KP, 0, 0, 0, wp, G.cur_stream())
    out = {}
    for dma in (1, 0):
        G.lib.jck_tune(b"igemm_dma_ksplit", dma)
        try:
            slab = torch.full((KS, B, N), float("nan"), device="cuda")
            G.lib.jck_linear_fwd(prec, xd, wp, None, slab, B, KP, N, N, KS, G.cur_stream())
            torch.cuda.synchronize()
            out[dma] = slab.cpu()
        finally:
            G.lib.jck_tune(b"igemm_dma_ksplit", 1)
    assert torch.equal(out[1], out[0]), float((out[1] - out[0]).abs().max())
    ref = x.float() @ w.to(torch.bfloat16).float().t()
    G.check(out[1].sum(0), ref, 2e-5, "sum of the slabs vs torch")


@pytest.mark.parametrize("prec", PRECS)
def test_head(G, prec):
    b, c = 16, 512
    g = torch.Generator().manual_seed(6)
    a4 = G.rnd(torch.randn(b, c, 4, 4, generator=g), prec).requires_grad_(True)
    w = (torch.randn(1, c, 4, 4, generator=g) * 0.02).requires_grad_(True)
    p = torch.sigmoid(F.conv2d(a4, w)).view(-1)
    loss = F.binary_cross_entropy(p, torch.full((b,), 0.9))
    loss.backward()
    wp = torch.empty(16 * c, device="cuda")
    G.lib.jck_pack_head(w.detach().cuda(), c, wp, G.cur_stream())
    a4d = G.to_nhwc(a4.detach(), prec)
    prob, ds = torch.empty(b, device="cuda"), torch.empty(b, device="cuda")
    ld = 64                                                  # per-image scalar table [8][ld]: plain stores, no atomics
    scal = torch.zeros(8, ld, device="cuda")
    G.lib.jck_head_fwd(prec, a4d, wp, None, b, 16 * c, 0.9, 0, prob, ds, scal, 0, 3, ld, G.cur_stream())
    ga = torch.empty_like(a4d)
    dwp = torch.full((16 * c,), 7.0, device="cuda")          # accumulate = 0 overwrites
    hws = torch.empty(G.lib.jck_head_bwd_ws_floats(16 * c), device="cuda")
    G.lib.jck_head_bwd(prec, ds, wp, a4d, b, 16 * c, ga, dwp, 0, hws, G.cur_stream())
    gw = torch.zeros(1, c, 4, 4, device="cuda")
    G.lib.jck_head_unpack_grad(dwp, c, gw, 0, G.cur_stream())
    torch.cuda.synchronize()
    G.check(prob.cpu(), p.detach(), 1e-5, "prob")
    assert abs(scal[0].sum().item() / b - loss.item()) < 1e-5 * abs(loss.item())
    assert abs(scal[3].sum().item() / b - p.mean().item()) < 1e-5
    assert float(scal[0, b:].abs().max()) == 0.0 and float(scal[1].abs().max()) == 0.0      # only [slot][0..b) is written
    G.check(G.from_nhwc(ga), a4.grad, 1e-5 if prec == 1 else 1e-2, "head dgrad")
    G.check(gw.cpu(), w.grad, 1e-5, "head wgrad")
    # the one-launch form the engine uses: dgrad + weight gradient accumulated straight into the PyTorch layout
    ga2 = torch.empty_like(a4d)
    gw2 = torch.full((1, c, 4, 4), 0.5, device="cuda")
    G.lib.jck_head_bwd_conv(prec, ds, wp, a4d, b, c, ga2, gw2, hws, G.cur_stream())
    torch.cuda.synchronize()
    assert torch.equal(ga2, ga)
    G.check(gw2.cpu() - 0.5, w.grad, 2e-5, "fused head wgrad")
    # two-stage sums instead of float atomics: a second launch gives the same bits
    gw3 = torch.full((1, c, 4, 4), 0.5, device="cuda")
    G.lib.jck_head_bwd_conv(prec, ds, wp, a4d, b, c, None, gw3, hws, G.cur_stream())
    torch.cuda.synchronize()
    assert torch.equal(gw3, gw2)
    # the batched D pass: loss rows (input + weight gradient) and, behind them, rows that only get their input gradient - one
    # launch with the bits of the two it replaces
    b1 = 10
    ga4 = torch.full_like(a4d, 3.0)
    gw4 = torch.full((1, c, 4, 4), 0.5, device="cuda")
    G.lib.jck_head_bwd_conv2(prec, ds, wp, a4d, b1, b - b1, c, ga4, gw4, hws, G.cur_stream())
    ga5, gw5 = torch.full_like(a4d, 3.0), torch.full((1, c, 4, 4), 0.5, device="cuda")
    G.lib.jck_head_bwd_conv(prec, ds, wp, a4d, b1, c, ga5, gw5, hws, G.cur_stream())
    G.lib.jck_head_bwd_conv(prec, ds[b1:], wp, a4d[b1:], b - b1, c, ga5[b1:], None, None, G.cur_stream())
    torch.cuda.synchronize()
    assert torch.equal(ga4, ga5) and torch.equal(ga4, ga) and torch.equal(gw4, gw5)
    # gradient-penalty mode: d sum(sigmoid) / d logit
    G.lib.jck_head_fwd(prec, a4d, wp, None, b, 16 * c, 0.0, 1, prob, ds, scal, -1, -1, ld, G.cur_stream())
    torch.cuda.synchronize()
    G.check(ds.cpu(), (p * (1 - p)).detach(), 1e-5, "gp ds")


def test_head_clamp(G):
    """saturated D: BCELoss clamps log at -100 and ATen's backward divides by max(p(1-p), 1e-12)."""
    b, c = 4, 512
    a4 = torch.ones(b, c, 4, 4)
    w = torch.full((1, c, 4, 4), 0.05)
    p = torch.sigmoid(F.conv2d(a4, w)).view(-1)
    assert float(p[0]) == 1.0
    ref = F.binary_cross_entropy(p, torch.full((b,), 0.1))
    wp = torch.empty(16 * c, device="cuda")
    G.lib.jck_pack_head(w.cuda(), c, wp, G.cur_stream())
    prob, ds, scal = torch.empty(b, device="cuda"), torch.empty(b, device="cuda"), torch.zeros(8, 8, device="cuda")
    G.lib.jck_head_fwd(1, G.to_nhwc(a4, 1), wp, None, b, 16 * c, 0.1, 0, prob, ds, scal, 0, 3, 8, G.cur_stream())
    torch.cuda.synchronize()
    assert scal[0].sum().item() / b == pytest.approx(ref.item(), rel=1e-6) == pytest.approx(90.0)
    assert float(ds.abs().max()) == 0.0


@pytest.mark.parametrize("off", [0, 1])
def test_adam(G, off):
    """off = 0: 16-byte aligned arenas (four elements per thread + a scalar tail, n % 4 = 3); off = 1: views that start one
    element into their buffers (the element-by-element path)."""
    g = torch.Generator().manual_seed(7)
    n = 10007
    p0 = torch.randn(n, generator=g)
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=2e-4, betas=[0.5, 0.999])
    buf = lambda t: torch.cat([torch.zeros(off), t]).cuda()[off:]
    pd, m, v = buf(p0.clone()), buf(torch.zeros(n)), buf(torch.zeros(n))
    for step in range(1, 6):
        gr = torch.randn(n, generator=g) * (10.0 ** (step - 3))
        p.grad = gr.clone()
        opt.step()
        G.lib.jck_adam(pd, buf(gr), m, v, n, 2e-4, 0.5, 0.999, 1e-8, step, 1.0, G.cur_stream())
        torch.cuda.synchronize()
        assert (pd.cpu() - p.detach()).abs().max().item() < 3e-7, step


@pytest.mark.parametrize("prec", PRECS)
def test_image_ops(G, prec):
    n, hw = 3, 64 * 64
    g = torch.Generator().manual_seed(8)
    img = torch.rand(n, 3, 64, 64, generator=g) * 2 - 1
    nz = torch.randn(n, 3, 64, 64, generator=g)
    out = torch.empty(n, 64, 64, 4, dtype=G.DT[prec], device="cuda")
    G.lib.jck_img_prep(prec, img.cuda(), nz.cuda(), 0.9, 0.1, out, n, hw, G.cur_stream())
    ref = 0.9 * img + 0.1 * nz
    tol = 1e-6 if prec == 1 else 8e-3
    G.check(G.from_nhwc(out, 3), ref, tol, "img_prep")
    back = torch.empty(n, 3, 64, 64, device="cuda")
    G.lib.jck_nhwc4_to_nchw(prec, out, back, n, hw, G.cur_stream())
    G.check(back.cpu(), G.from_nhwc(out, 3), 0, "nhwc4_to_nchw")
    x = G.rnd(torch.randn(n, 3, 64, 64, generator=g), prec)
    o2 = torch.empty_like(out)
    G.lib.jck_axpy_noise(prec, G.to_nhwc(x, prec), nz.cuda(), 0.9, 0.1, o2, n, hw, G.cur_stream())
    G.check(G.from_nhwc(o2, 3), 0.9 * x + 0.1 * nz, tol, "axpy_noise")
    al = torch.rand(n, generator=g)
    o3 = torch.empty_like(out)
    G.lib.jck_interp(prec, out, o2, al.cuda(), o3, n, hw, G.cur_stream())
    a_, b_ = G.from_nhwc(out, 3), G.from_nhwc(o2, 3)
    G.check(G.from_nhwc(o3, 3), al.view(n, 1, 1, 1) * a_ + (1 - al.view(n, 1, 1, 1)) * b_, tol, "interp")
    # noise mix + interpolate as one launch (explicit noise, and the in-kernel Philox stream): the bits of the two launches
    o2b, o3b = torch.empty_like(out), torch.empty_like(out)
    G.lib.jck_mix_interp(prec, G.to_nhwc(x, prec), nz.cuda(), None, 0, 0.9, 0.1, o2b, out, al.cuda(), o3b, n, hw, G.cur_stream())
    torch.cuda.synchronize()
    assert torch.equal(o2b, o2) and torch.equal(o3b, o3)
    rng = torch.tensor([123, 456, 7, 0], dtype=torch.int32, device="cuda")
    o2c, o3c, o2d, o3d = (torch.empty_like(out) for _ in range(4))
    G.lib.jck_axpy_noise_rng(prec, G.to_nhwc(x, prec), rng, 1, 0.9, 0.1, o2c, n, hw, G.cur_stream())
    G.lib.jck_interp(prec, out, o2c, al.cuda(), o3c, n, hw, G.cur_stream())
    G.lib.jck_mix_interp(prec, G.to_nhwc(x, prec), None, rng, 1, 0.9, 0.1, o2d, out, al.cuda(), o3d, n, hw, G.cur_stream())
    torch.cuda.synchronize()
    assert torch.equal(o2c, o2d) and torch.equal(o3c, o3d)
    scal = torch.zeros(8, 4, device="cuda")
    norms = torch.empty(n, device="cuda")
    G.lib.jck_gp_norm(prec, o3, n, hw, scal, 6, 4, norms, G.cur_stream())
    gi = G.from_nhwc(o3, 3).view(n, -1)
    G.check(norms.cpu(), gi.norm(2, dim=1), 1e-5, "gp norms")
    G.check(scal[6, :n].cpu(), (gi.norm(2, dim=1) - 1) ** 2, 1e-3, "gp penalty terms")
    o4 = torch.empty_like(out)
    yv = G.rnd(torch.tanh(torch.randn(n, 3, 64, 64, generator=g)), prec)
    G.lib.jck_tanh_bwd(prec, G.to_nhwc(x, prec), G.to_nhwc(yv, prec), 0.9, o4, n * hw * 4, G.cur_stream())
    G.check(G.from_nhwc(o4, 3), 0.9 * x * (1 - yv * yv), tol, "tanh_bwd")


def test_inkernel_philox_noise(G):
    """The instance noise drawn inside the image kernels (jck_*_rng; perf mode of train/dcgan_trainer.py:160,171): N(0,1) moments,
    independent across channels / tensors / steps, and a pure function of (seed, tensor, step, pixel)."""
    n, hw = 16, 64 * 64
    img = torch.zeros(n, 3, 64, 64, device="cuda")

    def draw(seed, step, tensor):
        rng = torch.tensor([seed & 0xFFFFFFFF, seed >> 32, step, 0], dtype=torch.int64).to(torch.int32).cuda()     # uint32 words
        out = torch.empty(n, 64, 64, 4, device="cuda")
        G.lib.jck_img_prep_rng(G.PREC_F32, img, rng, tensor, 0.9, 1.0, out, n, hw, G.cur_stream())
        torch.cuda.synchronize()
        return out

    a = draw(12345, 7, 0)
    assert torch.equal(a, draw(12345, 7, 0))                         # deterministic
    x = a[..., :3].double().reshape(-1)
    assert float(a[..., 3].abs().max()) == 0.0
    assert abs(x.mean().item()) < 3e-3 and abs(x.var().item() - 1.0) < 5e-3
    assert abs((x ** 4).mean().item() - 3.0) < 0.05 and abs((x ** 3).mean().item()) < 0.02      # kurtosis / skew of a normal
    assert 4.5 < x.abs().max().item() < 7.0                          # tails exist and are finite
    c = torch.corrcoef(a[..., :3].double().reshape(-1, 3).t())
    assert float((c.cpu() - torch.eye(3, dtype=torch.float64)).abs().max()) < 5e-3    # channels of a pixel are independent
    for other in (draw(12345, 7, 1), draw(12345, 8, 0), draw(12346, 7, 0)):          # tensor id, step, seed each change the stream
        y = other[..., :3].double().reshape(-1)
        assert abs(torch.corrcoef(torch.stack([x, y]))[0, 1].item()) < 5e-3
    # the same words through the other two kernels: axpy on an NHWC4 tensor, and the uint8 gather path
    rng = torch.tensor([12345, 0, 7, 0], dtype=torch.int32).cuda()
    o2 = torch.empty(n, 64, 64, 4, device="cuda")
    G.lib.jck_axpy_noise_rng(G.PREC_F32, torch.zeros(n, 64, 64, 4, device="cuda"), rng, 0, 0.9, 1.0, o2, n, hw, G.cur_stream())
    torch.cuda.synchronize()
    assert torch.equal(o2, a)
    data = torch.full((n, 3, 32, 32), 127, dtype=torch.uint8, device="cuda")
    o3 = torch.empty(n, 64, 64, 4, device="cuda")
    G.lib.jck_img_prep_u8_rng(G.PREC_F32, data, None, rng, 0, 0.0, 1.0, o3, n, 32, 32, G.cur_stream())
    torch.cuda.synchronize()
    assert torch.equal(o3, a)


def test_step_rng_small_inputs(G):
    """z ~ N(0,1), alpha ~ U[0,1), Dropout keep masks ~ Bernoulli(0.75) as the engine draws them when a step is given none
    (jck_engine_set_step -> adam_hp_kernel; perf mode of train/dcgan_trainer.py:168,111 and model/CGAN.py:105): moments,
    independence across the three tensors / steps / seeds, and a pure function of (seed, step, index)."""
    nz, na, nm = 256 * 100, 4099, 4 * 256 * 256

    def draw(seed, step):
        hp = torch.zeros(8, device="cuda")
        z, a, m = torch.empty(nz, device="cuda"), torch.empty(na, device="cuda"), torch.empty(nm, device="cuda")
        G.lib.jck_step_rng(hp, step, seed, z, nz, a, na, m, nm, 0.75, G.cur_stream())
        torch.cuda.synchronize()
        return z.double().cpu(), a.double().cpu(), m.double().cpu(), hp.cpu()
    z, a, m, hp = draw(99, 5)
    z2, a2, m2, _ = draw(99, 5)
    assert torch.equal(z, z2) and torch.equal(a, a2) and torch.equal(m, m2)
    assert hp.view(torch.int32)[4:7].tolist() == [99, 0, 5]                      # the Philox words the image kernels read
    assert abs(z.mean().item()) < 0.02 and abs(z.var().item() - 1.0) < 0.03 and abs((z ** 4).mean().item() - 3.0) < 0.15
    assert 0.0 <= a.min().item() and a.max().item() < 1.0 and abs(a.mean().item() - 0.5) < 0.02 and abs(a.var().item() - 1 / 12) < 0.005
    assert set(m.unique().tolist()) == {0.0, 1.0} and abs(m.mean().item() - 0.75) < 3e-3
    for other in (draw(99, 6), draw(100, 5)):                                     # step and seed each change every stream
        for x, y in zip((z, a, m), other[:3]):
            assert abs(torch.corrcoef(torch.stack([x, y]))[0, 1].item()) < 0.03
    k = min(nz, nm)
    assert abs(torch.corrcoef(torch.stack([z[:k], m[:k]]))[0, 1].item()) < 0.02  # tensor ids decorrelate z from the masks
    # ragged counts: nothing is written past the ends
    pad = torch.full((nz + 8,), 7.0, device="cuda")
    G.lib.jck_step_rng(torch.zeros(8, device="cuda"), 5, 99, pad, nz - 3, None, 0, None, 0, 0.75, G.cur_stream())
    torch.cuda.synchronize()
    assert float(pad[nz - 3:].min()) == 7.0 and torch.equal(pad[:nz - 3].double().cpu(), z[:nz - 3])


@pytest.mark.parametrize("cfg", [("down", 96, 32, 64, 128, 32), ("down", 192, 32, 64, 256, 64), ("up", 96, 16, 256, 128, 32),
                                 ("up", 48, 16, 128, 64, 16), ("down", 8, 16, 128, 256, 4), ("up", 64, 16, 256, 128, 64)])
def test_grouped_forward_statistics(G, cfg):
    """jck_conv_down_grouped / jck_conv_up_grouped: per-group sum y and sum y^2 of the conv output for jck_bn_finalize_grouped -
    on the persistent kernels one row per (workgroup, group), accumulated over the workgroup's tiles (large cases), elsewhere one
    row per (tile, wave) (the small case); either way row r of group k is stats[k * slots/groups + r]."""
    kind, n, hin, cin, cout, gimg = cfg
    prec = G.PREC_BF16
    g = torch.Generator().manual_seed(21)
    x = G.rnd(torch.randn(n, cin, hin, hin, generator=g), prec)
    w = torch.randn(cin, cout, 4, 4, generator=g) * 0.05 if kind == "up" else torch.randn(cout, cin, 4, 4, generator=g) * 0.05
    wr = G.rnd(w, prec)
    ref = F.conv_transpose2d(x, wr, None, 2, 1) if kind == "up" else F.conv2d(x, wr, None, 2, 1)
    hout = ref.shape[-1]
    out = torch.empty(n, hout, hout, cout, dtype=torch.bfloat16, device="cuda")
    stats, slots = G.stats_buf(n * hout * hout, cout)
    stats.fill_(float("nan"))
    if kind == "up":
        G.lib.jck_conv_up_grouped(prec, G.to_nhwc(x, prec), G.pack_up(w, prec), out, stats, ctypes.byref(slots), n, hin, hin, cin, cout,
                                  gimg, G.cur_stream())
    else:
        G.lib.jck_conv_down_grouped(prec, G.to_nhwc(x, prec), G.pack_down(w, prec), out, stats, ctypes.byref(slots), n, hin, hin, cin,
                                    cout, gimg, G.cur_stream())
    torch.cuda.synchronize()
    groups = n // gimg
    assert slots.value % groups == 0
    G.check(G.from_nhwc(out), ref, G.TOL[prec], "conv")
    per = slots.value // groups
    rows = stats[:slots.value * 2 * cout].view(groups, per, 2, cout).double().cpu()
    assert torch.isfinite(rows).all()                                 # every row of every group was written
    for k in range(groups):
        r = ref[k * gimg:(k + 1) * gimg].double()
        G.check(rows[k, :, 0].sum(0), r.sum((0, 2, 3)), 2e-3, f"sum, group {k}")
        G.check(rows[k, :, 1].sum(0), (r * r).sum((0, 2, 3)), 2e-3, f"sum of squares, group {k}")


# ---- BatchNorm statistics as exact integer-atomic sums (csrc/bnstat.hpp): the `_x` entry points -------------------------------
