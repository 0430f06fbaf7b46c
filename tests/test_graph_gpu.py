"""hipGraph replay of the step (hipgan/engine.py::_step_graph, jck_engine_capture_*): the replayed step must be the eager
step bit for bit - same kernels, same arguments; only the host side changes (one call per step segment instead of one per
kernel, inputs refilled in fixed-address buffers, Adam's per-step scalars through device memory)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(family, prec, B, steps, graphs, reducers=False, u8=False, pipeline_noise=True):
    import bf16_error as be
    from hipgan.engine import CganEngine, DcganEngine, DeviceBatch
    from oracle.gan_oracle import build_params
    from util import synth_images
    torch.manual_seed(12345)
    g, d = build_params(family)
    eng = (CganEngine if family == "cgan" else DcganEngine)(batch=B, prec=prec)
    eng.graphs = graphs
    eng.load_state(g, d)
    imgs = synth_images(B * steps)
    data = (torch.rand(64, 3, 32, 32, generator=torch.Generator().manual_seed(3)) * 255).to(torch.uint8).cuda()
    calls = []
    red = (lambda flat: calls.append(int(flat.numel())) or None) if reducers else None
    sc = []
    for s in range(steps):
        lab = be.labels_for(B, 5 + s) if family == "cgan" else None
        nz = be.noise_for(family, B, 40 + s, lab)
        nz = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in nz.items()}
        real = DeviceBatch(data, torch.randperm(64, generator=torch.Generator().manual_seed(s))[:B]) if u8 else imgs[s * B:(s + 1) * B].cuda()
        eng.step_async(real, nz, 2e-4 * (1 + s), reduce_d=red, reduce_g=red)        # a different lr every step: not baked
        sc.append(eng.scalars())
    torch.cuda.synchronize()
    n_graphs = len(eng._graph_cache)
    return sc, {k: v.clone() for k, v in eng.arenas.items()}, n_graphs, calls


@pytest.mark.parametrize("family,prec,B,reducers,u8", [("dcgan", "bf16", 16, False, False), ("dcgan", "f32", 8, True, False),
                                                       ("dcgan", "f32", 4, True, False),      # per-pass schedule: penalty pass on its own stream
                                                       ("dcgan", "bf16", 4, False, True), ("cgan", "bf16", 16, False, False),
                                                       ("cgan", "f32", 8, True, False), ("dcgan", "bf16", 256, False, False)])
def test_graph_replay_is_the_eager_step_bit_for_bit(family, prec, B, reducers, u8):
    steps = 5 if B < 256 else 3
    s_e, a_e, n_e, c_e = _run(family, prec, B, steps, False, reducers, u8)
    s_g, a_g, n_g, c_g = _run(family, prec, B, steps, True, reducers, u8)
    assert n_e == 0
    assert n_g == (2 if not reducers else 6), n_g                                     # one graph per (segment, parity)
    assert s_e == s_g, (s_e, s_g)
    # the reduce callbacks ran: eager data-parallel steps hand D's arena over in two pieces (tail first, PHASE_D_LOSS_A / _B), replayed
    # steps in one - either way every element of both arenas exactly once per step
    per_step = (a_e["d_grads"].numel() + a_e["g_grads"].numel()) if reducers else 0
    assert sum(c_e) == sum(c_g) == steps * per_step, (c_e, c_g)
    for k in a_e:
        assert torch.equal(a_e[k], a_g[k]), k


def test_device_rng_steps_replay_with_a_generator():
    """noise=None: the step draws n1 | z | n2, alpha straight into the fixed-address buffers with the caller's generator - the
    same draws as draw_noise() followed by an eager step."""
    from hipgan.engine import DcganEngine
    from oracle.gan_oracle import build_params
    from util import synth_images
    torch.manual_seed(12345)
    g, d = build_params("dcgan")
    B = 16
    imgs = synth_images(B).cuda()
    res = []
    for graphs in (False, True):
        eng = DcganEngine(batch=B, prec="bf16")
        eng.graphs = graphs
        eng.load_state(g, d)
        gen = torch.Generator(device="cuda").manual_seed(9)
        for s in range(4):
            eng.step_async(imgs, None, 2e-4, generator=gen)
        res.append((eng.scalars(), {k: v.clone() for k, v in eng.arenas.items()}))
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


def test_only_linear_graphs_are_instantiated():
    """jck_engine_capture_end refuses a captured graph with parallel branches.  ROCm 7.2's hipGraphLaunch of such a graph
    (hip::GraphExec::Run -> hip::Graph::UpdateStreams) walks the graph's pool of internal streams, skips each one that shares the
    launch stream's hardware queue and never checks the index against the pool's size: with two such streams it reads past the
    end of the vector - the SIGSEGV of round 2, whose faulting instruction is that dereference.  A graph
    of width 1 never enters that code, so the engine captures linearly and makes sure nothing else gets instantiated."""
    import ctypes as C

    from hipgan import JckError, lib
    from hipgan.engine import DcganEngine
    eng = DcganEngine(batch=8, prec="f32")
    st, side = torch.cuda.Stream(), torch.cuda.Stream()
    x, y = torch.zeros(1024, device="cuda"), torch.zeros(1024, device="cuda")
    torch.cuda.synchronize()
    out = C.c_void_p()
    # a linear capture is accepted ...
    lib.jck_engine_capture_begin(eng._h, st.cuda_stream)
    with torch.cuda.stream(st):
        x.add_(1)
        x.mul_(2)
    lib.jck_engine_capture_end(eng._h, st.cuda_stream, C.byref(out))
    lib.jck_graph_launch(out.value, st.cuda_stream)
    st.synchronize()
    assert float(x[0]) == 2.0
    lib.jck_graph_destroy(out.value)
    # ... a fork / join is not
    lib.jck_engine_capture_begin(eng._h, st.cuda_stream)
    with torch.cuda.stream(st):
        x.add_(1)
        side.wait_stream(st)
        with torch.cuda.stream(side):
            y.add_(1)
        x.add_(1)
        st.wait_stream(side)
        x.add_(y)
    with pytest.raises(JckError, match="parallel branches"):
        lib.jck_engine_capture_end(eng._h, st.cuda_stream, C.byref(out))
    torch.cuda.synchronize()
    assert float(x[0]) == 2.0                                         # nothing of the refused capture ran


def test_engine_noise_follows_the_seed():
    """The step's own draws (Philox: z, alpha, instance noise) are keyed by a seed that follows torch's seed unless the caller sets
    one (a one-GPU run must not see the same noise whatever the seed), and by the optimiser step."""
    from hipgan.engine import DcganEngine
    from oracle.gan_oracle import build_params
    from util import synth_images
    imgs = synth_images(8).cuda()
    fakes = []
    for seed in (1, 1, 2):
        torch.manual_seed(12345)
        g, d = build_params("dcgan")
        torch.manual_seed(seed)
        eng = DcganEngine(batch=8, prec="f32")
        eng.load_state(g, d)
        eng.step_async(imgs, None, 2e-4)
        fakes.append(eng.tensor("fake").clone())
    assert torch.equal(fakes[0], fakes[1])
    assert not torch.equal(fakes[0], fakes[2])
    eng.set_noise_seed(77)
    eng.t = 0
    eng.load_state(g, d)
    eng.step_async(imgs, None, 2e-4)
    assert not torch.equal(eng.tensor("fake"), fakes[2])
