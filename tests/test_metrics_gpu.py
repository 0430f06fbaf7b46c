"""metrics.py end to end on generated tensors (BASELINE.json: "generated tensors, Inception/FID from metrics.py match the
reference CPU run").  The fine-tuned Inception-v3 weights cannot be had offline (features unpinned), so the
pluggable extractor of metrics.Metrics is given a fixed, seeded stand-in network; with it the HIP generator's images must
score like the CPU oracle's images from the same weights and latents: FID(hip, oracle) ~ 0 against FID between two different
generators, equal FID to a common real set, equal Inception Score."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _StandIn(torch.nn.Module):
    """299x299 RGB -> 100 logits; fixed random weights (seed 0), deterministic on the CPU."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.w1 = torch.randn(16, 3, 7, 7, generator=g) * 0.2
        self.w2 = torch.randn(32, 16, 5, 5, generator=g) * 0.1
        self.fc = torch.randn(100, 32, generator=g) * 0.5

    def forward(self, x):
        x = x.float().cpu()
        x = torch.nn.functional.relu(torch.nn.functional.conv2d(x, self.w1, stride=4))
        x = torch.nn.functional.relu(torch.nn.functional.conv2d(x, self.w2, stride=4))
        return x.mean((2, 3)) @ self.fc.t()


@pytest.mark.parametrize("prec,fid_tol,is_tol", [("f32", 1e-6, 1e-5), ("bf16", 2e-3, 2e-3)])
def test_generated_images_score_like_the_oracles(prec, fid_tol, is_tol):
    import metrics
    from hipgan.engine import DcganEngine
    from oracle.gan_oracle import GanOracle
    from train.dcgan_trainer import inception_input
    n = 128
    orc = GanOracle("dcgan", lr=2e-4, seed=12345)
    other = GanOracle("dcgan", lr=2e-4, seed=777)                       # a different generator, for scale
    eng = DcganEngine(batch=n, prec=prec)
    eng.load_state(orc.g, orc.d)
    z = torch.randn(n, 100, 1, 1, generator=torch.Generator().manual_seed(9))
    img_hip = eng.sample(z)
    img_ref = orc.sample(z)
    img_other = other.sample(z)
    net = _StandIn()
    feats = lambda img: net(inception_input(img.cuda())).numpy().astype(np.float64)       # :202-206 through the HIP kernel
    f_hip, f_ref, f_other = feats(img_hip), feats(img_ref), feats(img_other)
    real = net(torch.rand(256, 3, 299, 299, generator=torch.Generator().manual_seed(5))).numpy().astype(np.float64)
    m = metrics.Metrics(None, extractor=net, real_features=real)
    scale = metrics.fid_from_features(f_ref, f_other)
    assert scale > 0
    assert metrics.fid_from_features(f_ref, f_hip) < fid_tol * scale
    fid_hip = m.fid([torch.from_numpy(inception_input(img_hip.cuda()).cpu().numpy())])
    fid_ref = m.fid([torch.from_numpy(inception_input(img_ref.cuda()).cpu().numpy())])
    assert abs(fid_hip - fid_ref) < max(fid_tol * 10, 1e-6) * abs(fid_ref) + 1e-9
    sm = lambda f: torch.softmax(torch.from_numpy(f), 1).numpy()
    is_hip, is_ref = metrics.inception_score_from_probs(sm(f_hip), 4), metrics.inception_score_from_probs(sm(f_ref), 4)
    assert abs(is_hip - is_ref) < is_tol * is_ref
