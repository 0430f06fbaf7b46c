"""Distribution-level gate for the bf16 fast path (BASELINE.json metric: "FID vs CPU ref"; reference metrics.py:113-129 is the
Frechet distance, train/dcgan_trainer.py:198-212 the sampling it scores).  The fine-tuned Inception weights exist nowhere
offline, so the distance is taken in the feature space of the SAME network with seeded random weights
(InceptionV3Hip + jck_mean_cov_f64 + metrics.fid_from_features) - a self-consistent proxy, not the reference's number.

Three trainings of K steps at batch B from ONE initial state and ONE noise / data sequence:
    fp32 oracle with 8 threads,  fp32 oracle with 1 thread (the reference's own run-to-run spread: its summation order
    changes with the thread count and the chaotic GAN map amplifies that - tests/golden/selfdiv.json),  the bf16 HIP engine.
Then N images are sampled from each generator with the same latents (train-mode BatchNorm batches of 100, as the reference
samples) and the three feature clouds compared.  The bf16 path is distributionally equivalent to fp32 training if its cloud is
no further from the fp32 run than the fp32 run is from ITSELF at another thread count (times two, plus a floor of 2 % of how
far training moved the distribution at all)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

K, B, N = 40, 32, 1000


def _noise(seed):
    g = torch.Generator().manual_seed(seed)
    return {"n1": torch.randn(B, 3, 64, 64, generator=g), "z": torch.randn(B, 100, 1, 1, generator=g),
            "n2": torch.randn(B, 3, 64, 64, generator=g), "alpha": torch.rand(B, 1, 1, 1, generator=g)}


def test_bf16_training_stays_within_the_fp32_runs_own_spread_in_feature_space():
    import metrics
    from hipgan.engine import DcganEngine
    from inception import InceptionV3Hip
    from oracle.gan_oracle import GanOracle
    from oracle.inception_oracle import random_state_dict
    from train.dcgan_trainer import inception_input
    from util import synth_images
    pool = synth_images(B * 8)
    threads0 = torch.get_num_threads()
    z = torch.randn(N, 100, 1, 1, generator=torch.Generator().manual_seed(9))
    net = InceptionV3Hip(random_state_dict(0), chunk=50)

    def features(sample):                      # sample(z_batch) -> [n,3,64,64] in [-1,1]
        img = torch.cat([sample(z[i:i + 100]).float().cpu() for i in range(0, N, 100)])
        return torch.cat([net(inception_input(img[i:i + 200].cuda())) for i in range(0, N, 200)]), img

    def train_oracle(nthreads):
        torch.set_num_threads(nthreads)
        orc = GanOracle("dcgan", lr=2e-4, seed=12345)
        for s in range(K):
            orc.step(pool[(s % 8) * B:(s % 8 + 1) * B], None, _noise(1000 + s))
        return orc

    try:
        init = GanOracle("dcgan", lr=2e-4, seed=12345)
        eng = DcganEngine(batch=B, prec="bf16")
        eng.load_state(init.g, init.d)
        for s in range(K):
            eng.step_async(pool[(s % 8) * B:(s % 8 + 1) * B].cuda(), {k: v.cuda() for k, v in _noise(1000 + s).items()}, 2e-4)
        sampler = DcganEngine(batch=100, share=eng)
        f_hip, img_hip = features(lambda zz: sampler.sample(zz.cuda()))
        f_init, _ = features(init.sample)
        o8 = train_oracle(8)
        f_8, img_8 = features(o8.sample)
        o1 = train_oracle(1)
        f_1, _ = features(o1.sample)
    finally:
        torch.set_num_threads(threads0)
    fid = metrics.fid_from_features
    d_self, d_bf16, d_moved = fid(f_8, f_1), fid(f_8, f_hip), fid(f_8, f_init)
    print(f"\nFID proxy after {K} steps at batch {B}, {N} samples: fp32(8 threads) vs fp32(1 thread) {d_self:.6g} | fp32(8 threads) vs bf16 HIP "
          f"{d_bf16:.6g} | fp32 trained vs untrained {d_moved:.6g} | pixel rms bf16 vs fp32 {float((img_hip - img_8).pow(2).mean().sqrt()):.4f}")
    assert np.isfinite([d_self, d_bf16, d_moved]).all() and d_moved > 0
    assert d_bf16 <= 2.0 * d_self + 0.02 * d_moved, (d_bf16, d_self, d_moved)
