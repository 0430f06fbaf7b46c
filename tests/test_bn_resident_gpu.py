"""The resident one-launch BatchNorm(+activation) backward (csrc/bnres.hpp, jck_bn_act_bwd_res) against
aten::native_batch_norm_backward + the activation's backward (autograd through F.batch_norm -> (leaky_)relu on the CPU,
model/DCGAN.py:30-33,62-65) and against the three-launch form it replaces (jck_bn_act_bwd_grouped): grouped batches, ragged
row counts, every channel-slice count, in place, two runs bitwise identical, repeated launches on one barrier state, and a
chip kept busy unevenly by another stream while the barrier spins."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_util
    gpu_util.lib.jck_tune(b"bn_res", 2)        # the resident form whenever it fits (the default leaves small multi-group passes to the three launches)
    yield gpu_util
    gpu_util.lib.jck_tune(b"bn_res", 1)


def _sync_ws(G):
    return torch.zeros(G.lib.jck_grid_sync_bytes() // 4, dtype=torch.int32, device="cuda")


def _case(G, rows, c, groups, slope, seed=3):
    g = torch.Generator().manual_seed(seed)
    y = (torch.randn(groups, rows, c, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    ga = torch.randn(groups, rows, c, generator=g).to(torch.bfloat16)
    gamma = 1 + 0.1 * torch.randn(c, generator=g)
    beta = 0.1 * torch.randn(c, generator=g)
    aux = torch.empty(groups, 4 * c)
    for k in range(groups):
        yf = y[k].float()
        mean, var = yf.mean(0), yf.var(0, unbiased=False)
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        aux[k] = torch.cat([gamma * invstd, beta - mean * gamma * invstd, mean, invstd])
    return y, ga, gamma, beta, aux


def _reference(y, ga, gamma, beta, slope, grad_groups):
    """autograd through batch_norm (training) -> activation, group by group; returns g_y, dgamma, dbeta"""
    gys, dg, db = [], torch.zeros_like(gamma), torch.zeros_like(beta)
    for k in range(y.shape[0]):
        yk = y[k].float().t().contiguous().unsqueeze(0).requires_grad_(True)          # [1, C, rows]
        gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        z = F.batch_norm(yk, None, None, gm, bt, True, 0.1, 1e-5)
        a = F.leaky_relu(z, slope) if slope else F.relu(z)
        a.backward(ga[k].float().t().contiguous().unsqueeze(0))
        gys.append(yk.grad[0].t())
        if k < grad_groups:
            dg += gm.grad
            db += bt.grad
    return torch.stack(gys), dg, db


def _run(G, fn_res, y, ga, aux, slope, grad_groups, ws=None, inplace=False):
    groups, rows, c = y.shape
    yd, gad, auxd = y.cuda(), ga.cuda(), aux.cuda().contiguous()
    sums = torch.full((groups * G.lib.jck_bn_bwd_ws_floats(c),), float("nan"), device="cuda")
    gy = gad if inplace else torch.full_like(gad, float("nan"))
    dgam, dbet = torch.ones(c, device="cuda"), torch.ones(c, device="cuda")
    if fn_res:
        rc = G.lib.jck_bn_act_bwd_res(0, gad, yd, auxd, slope, sums, gy, dgam, dbet, rows, c, groups, grad_groups, ws, G.cur_stream())
    else:
        rc = G.lib.jck_bn_act_bwd_grouped(0, gad, yd, auxd, slope, sums, gy, dgam, dbet, rows, c, groups, grad_groups, G.cur_stream())
    assert rc == 0, G.lib.jck_last_error()
    torch.cuda.synchronize()
    st = sums.view(groups, -1)[:, :2 * c].clone()
    return gy.float().cpu(), dgam.cpu() - 1, dbet.cpu() - 1, st.cpu()


# rows chosen to cover: one chunk per thread with most rows empty, ragged tails, several chunks, the production layer shapes
# at a small batch (C = 64 / 128 / 256 / 512 -> 1 / 2 / 4 / 8 channel slices)
CASES = [(512, 64, 1, 0.2, 1), (4096 + 37, 64, 3, 0.2, 2), (20000, 128, 2, 0.0, 2), (16384, 256, 3, 0.2, 2), (1000, 512, 1, 0.0, 1),
         (4096, 512, 3, 0.2, 2), (70000, 64, 1, 0.0, 1), (3, 128, 2, 0.2, 0),
         # groups that do NOT fit the registers together (groups * chunks > 16): the launch walks them in several passes, refills the
         # registers between the stores of one pass and the loads of the next and meets a second / third barrier generation - the
         # path D.conv1's layer takes in production at batch 256 (16 chunks, 2 groups behind the loss / penalty split)
         (262144, 64, 2, 0.2, 2), (140000, 64, 3, 0.2, 2)]


@pytest.mark.parametrize("case", CASES)
def test_resident_backward_matches_autograd_and_the_three_launch_form(G, case):
    rows, c, groups, slope, gg = case
    y, ga, gamma, beta, aux = _case(G, rows, c, groups, slope)
    ws = _sync_ws(G)
    gy, dg, db, st = _run(G, True, y, ga, aux, slope, gg, ws)
    assert G.lib.jck_grid_sync_error(ws) == 0
    ref_gy, ref_dg, ref_db = _reference(y, ga, gamma, beta, slope, gg)
    G.check(gy, ref_gy, 2e-2, "resident g_y vs autograd")
    if gg:
        G.check(dg, ref_dg, 1e-2, "dgamma")
        G.check(db, ref_db, 1e-2, "dbeta")
    gy3, dg3, db3, st3 = _run(G, False, y, ga, aux, slope, gg)
    # same arithmetic per element, another summation order of s1 / s2: equal to fp32 rounding, i.e. at most one bf16 ulp
    G.check(gy, gy3, 8e-3, "resident vs three-launch g_y")
    G.check(st, st3, 2e-4, "s1 | s2")
    G.check(dg, dg3, 2e-4, "dgamma vs three-launch")


@pytest.mark.parametrize("shape", [(16384 + 5, 128, 3), (262144, 64, 2)])        # the second: two passes, two barrier generations per launch
def test_in_place_repeated_and_bitwise_reproducible(G, shape):
    (rows, c, groups), slope = shape, 0.2
    y, ga, gamma, beta, aux = _case(G, rows, c, groups, slope, seed=9)
    ws = _sync_ws(G)
    first = None
    for it in range(6):                      # one barrier state over many launches: generation / counter bookkeeping
        gy, dg, db, st = _run(G, True, y, ga, aux, slope, 2, ws, inplace=(it % 2 == 1))
        if first is None:
            first = (gy, dg, db, st)
        else:
            for a, b in zip(first, (gy, dg, db, st)):
                assert torch.equal(a, b), f"launch {it} differs from launch 0"
    assert G.lib.jck_grid_sync_error(ws) == 0


def test_barrier_under_uneven_load_from_a_second_stream(G):
    """Another stream keeps a varying part of the chip busy (and the caches warm with the tensors) while the resident launch
    runs: the workgroups become resident at different times and the barrier must still hand over every partial row."""
    rows, c, groups, slope = 140000, 64, 3, 0.2          # 9 chunks x 3 groups: more than the registers hold, three passes
    y, ga, gamma, beta, aux = _case(G, rows, c, groups, slope, seed=5)
    ws = _sync_ws(G)
    base = _run(G, True, y, ga, aux, slope, 2, ws)
    side = torch.cuda.Stream()
    junk = torch.randn(64 << 20, device="cuda")
    yd, gad = y.cuda(), ga.cuda()
    for it in range(8):
        with torch.cuda.stream(side):
            for k in range(1 + it % 4):
                junk[: (4 << 20) * (1 + (it * 7 + k) % 13)].mul_(1.0001)
            _ = (yd.float().sum() + gad.float().sum())            # pre-read the tensors: warm lines in L1 / L2
        got = _run(G, True, y, ga, aux, slope, 2, ws)
        side.synchronize()
        for a, b in zip(base, got):
            assert torch.equal(a, b), f"round {it}: result changed under load"
    assert G.lib.jck_grid_sync_error(ws) == 0


def test_fallbacks_take_the_three_launch_form(G):
    y, ga, gamma, beta, aux = _case(G, 256, 32, 1, 0.2)          # C < 64
    a = _run(G, True, y, ga, aux, 0.2, 1, _sync_ws(G))
    b = _run(G, False, y, ga, aux, 0.2, 1)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    y, ga, gamma, beta, aux = _case(G, 2048, 64, 1, 0.2)         # no barrier state given
    a = _run(G, True, y, ga, aux, 0.2, 1, None)
    b = _run(G, False, y, ga, aux, 0.2, 1)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
