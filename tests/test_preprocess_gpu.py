"""Device-resident input pipeline: jck_img_prep_u8 against the oracle transform (bit-exact image, pinned to Pillow by
tests/golden/resize_u8.json) and a full engine step fed by indices against the same step fed by the transformed tensor."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_util
    return gpu_util


def test_img_prep_u8_is_bit_exact(G):
    from make_golden_resize import inputs
    from oracle.preprocess_oracle import transform
    x = inputs(7, 6, 32)
    data = torch.from_numpy(x).cuda()
    idx = torch.tensor([5, 0, 3, 3, 1], dtype=torch.int64, device="cuda")
    b = idx.numel()
    ref = torch.from_numpy(transform(x))[idx.cpu()]
    out = torch.empty(b, 3, 64, 64, device="cuda")
    G.lib.jck_img_prep_u8(G.PREC_F32, data, idx, None, 1.0, 0.0, None, out, b, 32, 32, G.cur_stream())
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref)
    # NHWC4 output with the instance-noise mix (train/dcgan_trainer.py:160), both precisions; idx = NULL takes the first B
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(b, 3, 64, 64, generator=g)
    for prec in (G.PREC_F32, G.PREC_BF16):
        o = torch.full((b, 64, 64, 4), 7.0, dtype=G.DT[prec], device="cuda")
        G.lib.jck_img_prep_u8(prec, data, None, noise.cuda(), 0.9, 0.1, o, None, b, 32, 32, G.cur_stream())
        torch.cuda.synchronize()
        want = 0.9 * torch.from_numpy(transform(x))[:b] + 0.1 * noise
        G.check(G.from_nhwc(o, 3), want, 1e-6 if prec == G.PREC_F32 else 1e-2, "noisy image")
        assert float(o[..., 3].float().abs().max()) == 0.0


def test_step_from_device_batch_equals_step_from_tensor():
    """Same weights, same noise: a step that gathers + transforms its batch from the uint8 dataset in HBM must give the
    scalars and weights of a step that is handed the transformed fp32 tensor."""
    from hipgan.engine import DcganEngine, DeviceBatch
    from model import DCGAN
    B = 16
    g = torch.Generator().manual_seed(11)
    data = (torch.rand(64, 3, 32, 32, generator=g) * 255).to(torch.uint8).cuda()
    idx = torch.randperm(64, generator=g)[:B]
    torch.manual_seed(12345)
    net_g, net_d = DCGAN.Generator(), DCGAN.Discriminator()
    net_g.apply(DCGAN.weights_init)
    net_d.apply(DCGAN.weights_init)
    res = []
    for mode in ("tensor", "indices"):
        eng = DcganEngine(batch=B, prec="f32", device="cuda:0")
        eng.load_state(net_g.state_dict(), net_d.state_dict())
        batch = DeviceBatch(data, idx)
        gen = torch.Generator(device="cuda").manual_seed(5)
        for _ in range(2):
            noise = eng.draw_noise(gen)
            eng.step_async(batch.materialize() if mode == "tensor" else batch, noise, 2e-4)
        sc = eng.scalars()
        gs, ds = eng.state_dicts()
        res.append((sc, gs, ds))
    (s0, g0, d0), (s1, g1, d1) = res
    # the transform is bit-exact and the step has no float atomics: the two runs agree bit for bit
    assert s0 == s1, (s0, s1)
    for a, b in ((g0, g1), (d0, d1)):
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_device_loader_covers_dataset_and_feeds_trainer_batches():
    from preprocess.dcgan_data_preprocessor import DeviceLoader
    data = torch.arange(40, dtype=torch.uint8).view(40, 1, 1, 1).expand(40, 3, 32, 32).contiguous().cuda()
    onehot = torch.nn.functional.one_hot(torch.arange(40) % 100, 100).cuda()
    ld = DeviceLoader(data, 16, onehot=onehot, seed=1)
    assert len(ld) == 3
    seen = []
    for batch in ld:
        img, lab = batch
        t = img.materialize()
        assert t.shape == (img.size(0), 3, 64, 64)
        ids = ((t[:, 0, 0, 0] * 0.5 + 0.5) * 255).round().long().cpu()          # constant images: pixel value = index
        assert torch.equal(lab.argmax(1).cpu(), ids % 100)
        seen += ids.tolist()
    assert sorted(seen) == list(range(40))
    assert [b[0].size(0) for b in ld] == [16, 16, 8]                            # ragged last batch, second epoch reshuffled


def test_resize_norm_matches_aten_bilinear(G):
    """Evaluation branch (train/dcgan_trainer.py:202-206) against the ops the reference calls, run by PyTorch on the CPU."""
    g = torch.Generator().manual_seed(4)
    fake = torch.tanh(torch.randn(5, 3, 64, 64, generator=g))
    mean = torch.tensor([0.485, 0.456, 0.406])
    std = torch.tensor([0.229, 0.224, 0.225])
    ref = torch.nn.functional.interpolate(0.5 * fake + 0.5, size=[299, 299], mode="bilinear", align_corners=False)
    ref = (ref - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1)
    out = torch.empty(5, 3, 299, 299, device="cuda")
    G.lib.jck_resize_norm(fake.cuda(), out, 5, 3, 64, 64, 299, 299, 0.5, 0.5, mean.cuda(), std.cuda(), G.cur_stream())
    torch.cuda.synchronize()
    G.check(out.cpu(), ref, 2e-6, "resize + normalise")
    from train.dcgan_trainer import inception_input
    assert torch.equal(inception_input(fake.cuda()).cpu(), out.cpu())
    # non-square, downscale factor and generic affine
    x = torch.randn(2, 4, 20, 12, generator=g)
    m, s = torch.zeros(4), torch.ones(4)
    ref = torch.nn.functional.interpolate(2.0 * x - 1.0, size=[33, 47], mode="bilinear", align_corners=False)
    out = torch.empty(2, 4, 33, 47, device="cuda")
    G.lib.jck_resize_norm(x.cuda(), out, 2, 4, 20, 12, 33, 47, 2.0, -1.0, m.cuda(), s.cuda(), G.cur_stream())
    torch.cuda.synchronize()
    G.check(out.cpu(), ref, 2e-6, "generic resize")


def test_device_loader_rank_sharding():
    """Under torch.distributed every rank walks a strided share of ONE permutation (DistributedSampler semantics): together
    the ranks cover the dataset, the shares have equal length (wrap-around padding) and a rank's epochs differ."""
    from preprocess.dcgan_data_preprocessor import DeviceLoader
    n, world = 50, 4
    data = torch.arange(n, dtype=torch.uint8).view(n, 1, 1, 1).expand(n, 3, 32, 32).contiguous().cuda()
    seen, lens = [], []
    first_epochs = []
    for rank in range(world):
        ld = DeviceLoader(data, 8, seed=7)
        ld.rank, ld.world = rank, world                    # what __init__ reads from torch.distributed
        ld.n_local = (n + world - 1) // world
        ids = torch.cat([b[0].idx for b in ld]).cpu()
        first_epochs.append(ids)
        assert ids.numel() == 13 and len(ld) == 2
        lens.append(ids.numel())
        seen += ids.tolist()
        ids2 = torch.cat([b[0].idx for b in ld]).cpu()
        assert not torch.equal(ids, ids2)                  # reshuffled next epoch
    assert set(seen) == set(range(n)) and len(seen) == 52  # 2 wrap-around duplicates
    assert len(set(lens)) == 1
