"""Two engine replicas as two processes on ONE GPU (gloo carries the CUDA tensors; RCCL needs one device per rank): the
native step with gradient exchange reproduces the N-replica CPU oracle.  RCCL itself runs here through a ONE-rank "nccl"
process group (test_rccl_world1_*): that is the real ProcessGroupNCCL stream path - the collectives are enqueued on RCCL's
stream behind the issuing stream - and SUM over one rank must leave every bit unchanged.
No run on two or more devices has happened yet (no multi-GPU node was available to the build; the driver's SCALE run is the
first)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _noise(B, seed):
    g = torch.Generator().manual_seed(seed)
    return {"n1": torch.randn(B, 3, 64, 64, generator=g), "z": torch.randn(B, 100, 1, 1, generator=g),
            "n2": torch.randn(B, 3, 64, 64, generator=g), "alpha": torch.rand(B, 1, 1, 1, generator=g)}


def _worker(rank, world, port, q, steps, B):
    for p in (ROOT, os.path.join(ROOT, "jck-generation_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ["JCK_BN_RES"] = "0"                 # two ranks share ONE GPU here: a resident (grid-barrier) launch needs the chip to itself
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hipgan.dist import GradReducer
    from hipgan.engine import DcganEngine
    from oracle.gan_oracle import build_params
    from util import synth_images
    torch.manual_seed(12345)
    g, d = build_params("dcgan")
    eng = DcganEngine(batch=B, prec="f32", device="cuda:0")
    eng.load_state(g, d)
    red = GradReducer(world)
    imgs = synth_images(B * world * steps)
    out = []
    batch = lambda s: (imgs[(s * world + rank) * B:(s * world + rank + 1) * B].cuda(),
                       {k: v.cuda() for k, v in _noise(B, 1000 + 10 * s + rank).items()})
    cur = batch(0)
    for s in range(steps):
        nxt = batch(s + 1) if s + 1 < steps else (None, None)
        # the next batch is announced: with the batched schedule (B = 8) the forward half of its D(real) pass runs under G's
        # all-reduce; the per-pass schedule (B = 4) has no such split and ignores the announcement
        eng.step_async(cur[0], cur[1], 2e-4, reduce_d=red.start, reduce_g=red.start, grad_scale=1.0 / world,
                       next_real=nxt[0], next_noise=nxt[1])
        out.append(eng.scalars())
        cur = nxt
    gs, ds = eng.state_dicts()
    np_ = lambda sd: {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}      # numpy: no shared-memory handles
    q.put((rank, out, np_(gs), np_(ds)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 8])
def test_two_replicas_match_ddp_oracle(B):
    """B = 4: per-pass schedule, one all-reduce per network.  B = 8: batched schedule - the tail of D's gradient arena is
    all-reduced between the two halves of D's backward (PHASE_D_LOSS_A / _B), the rest at the end of the phase."""
    from oracle.gan_oracle import GanOracle, ddp_step
    from util import synth_images
    world, steps = 2, 2                  # the second step's D(real) forward is announced by the first (B = 8)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, steps, B)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, out, gs, ds = q.get(timeout=240)
        got[r] = (out, {k: torch.from_numpy(v) for k, v in gs.items()}, {k: torch.from_numpy(v) for k, v in ds.items()})
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    reps = [GanOracle("dcgan", lr=2e-4, seed=12345) for _ in range(world)]
    imgs = synth_images(B * world * steps)
    for s in range(steps):
        res = ddp_step(reps, [imgs[(s * world + r) * B:(s * world + r + 1) * B] for r in range(world)], None,
                       [_noise(B, 1000 + 10 * s + r) for r in range(world)])
        for r in range(world):
            for k in ("loss_d", "loss_g", "gp"):
                a, b = got[r][0][s][k], res[r][k]
                # step 0 starts from identical state (1e-3).  Later steps run on weights that Adam has stepped: an element
                # whose gradient was within rounding of 0 moved the other way (2*lr), which the penalty term feels most
                assert abs(a - b) <= (1e-3 if s == 0 else 3e-3) * abs(b), (s, r, k, a, b)
    for r in range(world):
        for k, v in reps[r].g.items():
            if v.dtype == torch.float32 and "running" not in k:
                # two Adam steps of lr*sign(g): near-zero gradients may step the other way (<= 2*lr per step) - max-norm
                # slack 4*lr; the conv weights (scale 0.02) are additionally held to 2e-3 in relative L2
                assert (got[r][1][k] - v).abs().max() <= 8.5e-4, ("g", r, k)
                if k.startswith("conv"):
                    e = (got[r][1][k] - v).norm() / (v.norm() + 1e-30)
                    assert e < 2e-3, ("g", r, k, float(e))
    for k, v in got[0][1].items():
        if v.dtype == torch.float32 and "running" not in k:
            assert torch.equal(v, got[1][1][k]), k          # replicas in lock-step


def _trainer_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "jck-generation_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import argparse
    import tempfile
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ["JCK_BN_RES"] = "0"                 # two ranks share ONE GPU here (see _worker)
    os.chdir(tempfile.mkdtemp())
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import main          # noqa: F401  (seeds every rank with RANDOMSEED, as `python main.py` does)
    from model import DCGAN
    from preprocess.dcgan_data_preprocessor import DCGANDataPreprocessor
    from train.dcgan_trainer import DCGANTrainer
    args = argparse.Namespace(epoch=1, max_learning_rate=2e-4, model_path="t", log_file=0, save_path="save", batch_size=8, num_worker=0)
    pre = DCGANDataPreprocessor(args, synthetic_size=32)
    pre.transform_data()
    tr = DCGANTrainer(args, DCGAN.Generator(), DCGAN.Discriminator(), pre, prec="f32")
    w0 = float(tr.engine.arenas["g_params"].double().sum())
    nz = tr.engine.draw_noise(tr.noise_gen)
    zsum, asum = float(nz["z"].double().sum()), float(nz["alpha"].double().sum())
    tr.train()                                     # 2 iterations per rank (32 images / 2 ranks / batch 8), gradients exchanged
    w1 = float(tr.engine.arenas["g_params"].double().sum())
    q.put((rank, w0, zsum, asum, w1))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_trainer_draws_its_own_noise_per_rank():
    """main.py seeds every rank alike (identical initial weights); the trainer must still give every rank its own noise
    stream, or all replicas generate the same fake batch and the all-reduce averages N identical gradients."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 7) % 2000)
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r = q.get(timeout=240)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0]                              # same initial weights
    assert got[0][1] != got[1][1] and got[0][2] != got[1][2]   # different z and alpha
    assert got[0][3] == got[1][3] and got[0][3] != got[0][0]   # replicas stay in lock-step and did move


_RCCL1 = r"""
import os, sys, torch, torch.distributed as dist
ROOT = sys.argv[1]
for p in (ROOT, os.path.join(ROOT, "jck-generation_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from hipgan.dist import GradReducer
from hipgan.engine import DcganEngine
from oracle.gan_oracle import build_params
from util import synth_images
import bf16_error as be
B = 16
torch.manual_seed(12345)
g, d = build_params("dcgan")
imgs = synth_images(B * 3)
def run(reduce, overlap, native=False):
    eng = DcganEngine(batch=B, prec="bf16", device="cuda:0")
    eng.ddp_overlap = overlap                # D's arena in two pieces under its own backward / one all-reduce per network
    eng.load_state(g, d)
    # native: the library's own communicator (jck_comm_*: RCCL on a stream of its own, ordered by events) instead of ProcessGroupNCCL
    red = GradReducer(1, force=True, native=native) if reduce else None
    for s in range(3):
        nz = {k: v.cuda() for k, v in be.noise_for("dcgan", B, 70 + s).items()}
        kw = dict(reduce_d=red.start, reduce_g=red.start) if red else {}
        eng.step_async(imgs[s * B:(s + 1) * B].cuda(), nz, 2e-4, **kw)
    eng.join()
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in eng.arenas.items()}, eng.scalars()
base, sb = run(False, True)
for native in (False, True):
    for overlap in (False, True):
        got, sg = run(True, overlap, native)
        assert sg == sb, (native, overlap, sg, sb)
        for k in base:
            assert torch.equal(base[k], got[k]), (native, overlap, k)
# the C-ABI communicator by itself: SUM over one rank leaves a buffer as it is, in stream order behind its producer
red = GradReducer(1, force=True, native=True)
assert red._comm is not None
x = torch.randn(1 << 20, device="cuda")
y = x * 2.0
w = red.start(y)
w()
assert torch.equal(y, x * 2.0)
red.close()
dist.destroy_process_group()
print("RCCL1-OK")
"""


def test_rccl_world1_reduce_paths_leave_every_bit_unchanged(tmp_path):
    """GradReducer.start and the full step_async(reduce_d=, reduce_g=) through a ONE-rank RCCL group, with D's arena reduced
    in two pieces under its own backward and as one all-reduce: the results must equal the no-reduce step bit for bit (SUM over one rank is the
    identity, the step has no float atomics) - which they only do if every collective is ordered behind the gradient
    kernels it reduces and ahead of the optimiser kernels that read it."""
    import subprocess
    script = tmp_path / "rccl1.py"
    script.write_text(_RCCL1)
    port = str(29500 + ((os.getpid() + 13) % 2000))
    r = subprocess.run([sys.executable, str(script), ROOT, port], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL1-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("prec,explicit", [("bf16", False), ("f32", True)])
def test_d_real_forward_under_the_g_allreduce_changes_no_bit(prec, explicit):
    """step_async(next_real=...) moves the forward half of the next step's D(real) pass in front of this step's G optimiser
    phase (PHASE_D_REAL_FWD - the work that hides G's gradient all-reduce when data parallel), and with a reducer the D pass is
    issued as PHASE_D_LOSS_A / _B with the all-reduce of the gradient arena's tail started in between.  Same kernels on the same data
    in the same per-tensor order: every weight, Adam moment, BatchNorm statistic and logged scalar must be bit-identical to
    the plain order, with in-kernel instance noise and with explicit noise tensors."""
    sys.path.insert(0, os.path.join(ROOT, "jck-generation_amd"))
    from hipgan.engine import DcganEngine
    from oracle.gan_oracle import build_params
    from util import synth_images
    B, steps = 16, 4
    imgs = synth_images(B * steps).cuda()
    res = []
    # plain order and one all-reduce per net  vs  everything that overlaps  vs  one GPU (no reducer) with the next batch announced
    for announce, split in ((False, "0"), (True, "1"), (True, None)):
        single = split is None
        os.environ["JCK_DDP_SPLIT"] = split or "1"           # PHASE_D_LOSS_A / _B: D's arena tail is reduced under the rest of the backward
        torch.manual_seed(12345)
        g, d = build_params("dcgan")
        eng = DcganEngine(batch=B, prec=prec, device="cuda:0")
        eng.graphs = False
        eng.load_state(g, d)
        eng.set_noise_seed(77)
        gen = torch.Generator(device="cuda").manual_seed(5)
        waits = []
        red = lambda flat, **kw: (lambda: waits.append(1))            # a reducer that changes nothing (world 1)
        scal = []
        for s in range(steps):
            nz = {k: v.cuda() for k, v in _noise(B, 300 + s).items()} if explicit else None
            nn = ({k: v.cuda() for k, v in _noise(B, 300 + s + 1).items()} if explicit else None) if s + 1 < steps else None
            kw = dict(next_real=imgs[(s + 1) * B:(s + 2) * B], next_noise=nn) if (announce and s + 1 < steps) else {}
            eng.step_async(imgs[s * B:(s + 1) * B], nz, 2e-4, reduce_d=None if single else red, reduce_g=None if single else red, generator=gen, **kw)
            if announce and s + 1 < steps:
                assert eng._prefetched_real is not None          # PHASE_D_REAL_FWD of the next step is enqueued
            scal.append(eng.scalars())
        assert len(waits) == (0 if single else 3 if split == "1" else 2) * steps
        if announce:
            assert getattr(eng, "_prefetch_ok", True)
        gs, ds = eng.state_dicts()
        res.append((scal, {k: v.clone() for k, v in gs.items()}, {k: v.clone() for k, v in ds.items()},
                    {t: {k: v.clone() for k, v in eng.named_views(t, w).items()} for t in "gd" for w in ("m",)}))
    os.environ.pop("JCK_DDP_SPLIT", None)
    for other in (res[1], res[2]):
        assert res[0][0] == other[0], (res[0][0], other[0])
        for a, b in ((res[0][1], other[1]), (res[0][2], other[2]), (res[0][3]["g"], other[3]["g"]), (res[0][3]["d"], other[3]["d"])):
            for k in a:
                assert torch.equal(a[k], b[k]), k


def test_split_d_pass_is_offered_only_where_it_exists():
    """jck_engine_grad_tail: the arena offset PHASE_D_LOSS_A finalises - DCGAN with the batched schedule only; CGAN (gradients keep
    arriving until the penalty's double backward) and the per-pass schedule (batch % 8 != 0) answer -1 and refuse the phases."""
    sys.path.insert(0, os.path.join(ROOT, "jck-generation_amd"))
    import ctypes
    from hipgan import JckError
    from hipgan._lib import lib
    from hipgan.engine import CganEngine, DcganEngine, PHASE_D_LOSS_A, StepInputs
    eng = DcganEngine(batch=16, prec="bf16", device="cuda:0")
    tail = lib.jck_engine_grad_tail(eng._h, 1)
    views = eng.named_views("d", "grads")
    names = list(views.keys())
    first = views[names[names.index("conv4.weight")]]
    assert tail == (first.data_ptr() - eng.arenas["d_grads"].data_ptr()) // 4 and tail > 0
    assert lib.jck_engine_grad_tail(eng._h, 0) == -1                                  # G has no such split
    for other in (DcganEngine(batch=4, prec="bf16", device="cuda:0"), CganEngine(batch=16, prec="bf16", device="cuda:0")):
        assert lib.jck_engine_grad_tail(other._h, 1) == -1
        si = StepInputs()
        si.step = 1
        si.labels = torch.zeros(16, 100, dtype=torch.int64, device="cuda").data_ptr()
        with pytest.raises(JckError):
            lib.jck_engine_phase(other._h, PHASE_D_LOSS_A, ctypes.byref(si), None)


def _guard_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "jck-generation_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ["JCK_BN_RES"] = "0"                 # two ranks share ONE GPU here (see _worker)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hipgan.dist import GradReducer, ReplicaGuard
    from hipgan.engine import DcganEngine
    from oracle.gan_oracle import build_params
    from util import synth_images
    B = 8
    torch.manual_seed(12345)
    g, d = build_params("dcgan")
    eng = DcganEngine(batch=B, prec="f32", device="cuda:0")
    eng.load_state(g, d)
    red = GradReducer(world)
    msgs = []
    guard = ReplicaGuard(eng, world, log=msgs.append)
    imgs = synth_images(B * world * 3)

    def step(s):
        eng.step_async(imgs[(s * world + rank) * B:(s * world + rank + 1) * B].cuda(),
                       {k: v.cuda() for k, v in _noise(B, 500 + 10 * s + rank).items()}, 2e-4,
                       reduce_d=red.start, reduce_g=red.start, grad_scale=1.0 / world)
        torch.cuda.synchronize()
    step(0)
    ok0, over0 = guard.check("after step 0 "), eng.ddp_overlap
    if rank == 1:                                  # one replica drifts (what a misordered collective would do)
        eng.arenas["d_params"][7] += 1e-3
        eng.mark_weights_changed()
    ok1 = guard.check("after the injected divergence ")
    over1, ok2 = eng.ddp_overlap, guard.in_sync()
    step(1)                                        # the plain schedule on re-broadcast state
    ok3 = guard.in_sync()
    q.put((rank, ok0, over0, ok1, over1, ok2, ok3, len(msgs), float(eng.arenas["d_params"].double().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_guard_detects_a_divergence_and_falls_back():
    """hipgan.dist.ReplicaGuard (what both trainers call after the first steps and at every evaluation point): in-sync replicas
    pass and keep the overlapped schedule; once one rank's parameters drift the check fails on EVERY rank, rank 0's state is
    broadcast again, the engines switch to one all-reduce per network (ddp_overlap False), the event is logged once, and the
    replicas are - and stay - identical afterwards."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 77) % 2000)
    procs = [ctx.Process(target=_guard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r = q.get(timeout=240)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank in (0, 1):
        ok0, over0, ok1, over1, ok2, ok3, nmsg, _ = got[rank]
        assert ok0 and over0, (rank, got[rank])
        assert not ok1 and not over1 and ok2 and ok3 and nmsg == 1, (rank, got[rank])
    assert got[0][7] == got[1][7]
