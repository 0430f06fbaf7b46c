"""The metric network on the device (csrc/infer.hip, jck-generation_amd/inception.py) against plain torch on the CPU: the generic
NHWC conv / pool kernels operator by operator, the whole Inception-v3 chain against the CPU restatement
(oracle/inception_oracle.py) on seeded random weights, the fp64 mean / covariance against numpy, and the Metrics class
driving it.  Parity against the REFERENCE's features is unpinned (no torchvision, no fine-tuned weights offline)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,h,w,cin,cout,k,s,p", [(2, 17, 17, 768, 192, (1, 1), (1, 1), (0, 0)), (3, 35, 35, 48, 64, (5, 5), (1, 1), (2, 2)),
                                                  (2, 17, 17, 128, 128, (1, 7), (1, 1), (0, 3)), (2, 17, 17, 128, 192, (7, 1), (1, 1), (3, 0)),
                                                  (2, 35, 35, 288, 384, (3, 3), (2, 2), (0, 0)), (2, 29, 31, 3, 32, (3, 3), (2, 2), (0, 0)),
                                                  (5, 1, 1, 2048, 100, (1, 1), (1, 1), (0, 0))])
def test_conv2d_nhwc(n, h, w, cin, cout, k, s, p):
    from hipgan import lib
    from hipgan._lib import cur_stream
    g = torch.Generator().manual_seed(n * 1000 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k[0], k[1], generator=g) * (2.0 / (cin * k[0] * k[1])) ** 0.5
    sc, sh = 0.5 + torch.rand(cout, generator=g), torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv2d(x, wt, None, s, p) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    oh, ow = ref.shape[2:]
    wk = wt.permute(2, 3, 1, 0).reshape(-1, cout).contiguous().cuda()
    total, off = cout + 24, 8                                       # written into a channel slice of a wider tensor
    out = torch.full((n, oh, ow, total), 7.0, device="cuda")
    lib.jck_conv2d_nhwc_f32(x.permute(0, 2, 3, 1).contiguous().cuda(), wk, sc.cuda(), sh.cuda(), out, n, h, w, cin, k[0], k[1], s[0], s[1],
                            p[0], p[1], cout, total, off, 1, cur_stream())
    got = out[..., off:off + cout].permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert float(out[..., :off].min()) == 7.0 and float(out[..., off + cout:].max()) == 7.0      # neighbours untouched


def test_pools_nhwc():
    from hipgan import lib
    from hipgan._lib import cur_stream
    x = torch.randn(3, 40, 35, 33, generator=torch.Generator().manual_seed(2))
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    for mode, ref in ((0, F.max_pool2d(x, 3, 2)), (1, F.avg_pool2d(x, 3, 1, 1))):
        k, s, p = (3, 2, 0) if mode == 0 else (3, 1, 1)
        out = torch.zeros(3, ref.shape[2], ref.shape[3], 50, device="cuda")
        lib.jck_pool2d_nhwc_f32(xd, out, 3, 35, 33, 40, k, s, p, mode, 50, 10, cur_stream())
        assert (out[..., 10:].permute(0, 3, 1, 2).cpu() - ref).abs().max().item() < 1e-6
    gp = torch.empty(3, 40, device="cuda")
    lib.jck_global_avgpool_nhwc_f32(xd, gp, 3, 35 * 33, 40, cur_stream())
    assert (gp.cpu() - x.mean((2, 3))).abs().max().item() < 1e-6


def test_inception_v3_chain_matches_the_cpu_restatement():
    from inception import InceptionV3Hip
    from oracle.inception_oracle import inception_logits, random_state_dict
    sd = random_state_dict(0)
    x = torch.randn(5, 3, 299, 299, generator=torch.Generator().manual_seed(1))
    ref = inception_logits(sd, x)
    net = InceptionV3Hip(sd, chunk=3)                                # 5 images in chunks of 3 + 2
    got = net(x.cuda()).cpu()
    assert got.shape == (5, 100)
    err = (got - ref).abs().max().item()
    assert err <= 1e-3 * ref.abs().max().item(), (err, ref.abs().max().item())
    with pytest.raises(Exception):
        net(x)                                                       # CPU tensor: no fallback


def test_local_weights_loader(tmp_path):
    from inception import InceptionV3Hip
    from oracle.inception_oracle import random_state_dict
    sd = random_state_dict(3)
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}}, tmp_path / "w.pt")
    a, b = InceptionV3Hip(sd), InceptionV3Hip.from_file(str(tmp_path / "w.pt"))
    x = torch.randn(2, 3, 299, 299, generator=torch.Generator().manual_seed(5)).cuda()
    assert torch.equal(a(x), b(x))


def test_mean_cov_fp64_on_the_device():
    from metrics import fid_from_features, mean_cov
    rng = np.random.default_rng(7)
    x = (1.5 * rng.standard_normal((5000, 100)) + 0.2).astype(np.float32)
    y = rng.standard_normal((1000, 100)).astype(np.float32)
    mu, cov = mean_cov(torch.from_numpy(x).cuda())
    assert np.abs(mu - np.mean(x.astype(np.float64), axis=0)).max() < 1e-12
    assert np.abs(cov - np.cov(x.astype(np.float64), rowvar=False)).max() < 1e-11
    a = fid_from_features(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    b = fid_from_features(x.astype(np.float64), y.astype(np.float64))
    assert abs(a - b) < 1e-8 * abs(b), (a, b)


def test_metrics_with_the_hip_extractor():
    """Metrics(...) as the trainers use it, with the HIP network as the extractor: IS / FID / intra-FID run end to end on device
    features and agree with the same arithmetic applied to the CPU restatement's features."""
    from inception import InceptionV3Hip
    from metrics import Metrics, fid_from_features, inception_score_from_probs
    from oracle.inception_oracle import inception_logits, random_state_dict
    sd = random_state_dict(0)
    net = InceptionV3Hip(sd)
    g = torch.Generator().manual_seed(11)
    real_feats = torch.randn(400, 100, generator=g).numpy() * 2.0

    class Src:
        targets = [i % 100 for i in range(400)]
    m = Metrics(Src(), extractor=net, real_features=real_feats)
    fake = torch.randn(20, 3, 299, 299, generator=g)
    loader = lambda: torch.utils.data.DataLoader(fake, batch_size=8)
    ref = inception_logits(sd, fake)
    is_ref = inception_score_from_probs(torch.softmax(ref, 1).numpy(), splits=2)
    fid_ref = fid_from_features(real_feats, ref.numpy())
    assert abs(m.inception_score(loader(), splits=2) - is_ref) < 2e-3 * is_ref
    assert abs(m.fid(loader()) - fid_ref) < 2e-3 * abs(fid_ref)
