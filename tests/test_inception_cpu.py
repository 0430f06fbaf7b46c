"""Host-side checks of the metric network's description (no GPU): the layer table gives the published Inception-v3."""
import torch


def test_layer_table_is_the_published_architecture():
    from inception import conv_specs
    specs = conv_specs()
    assert len(specs) == 94                                            # BasicConv2d modules of torchvision's inception_v3 (no aux head)
    n_conv = sum(cin * cout * k[0] * k[1] for _, cin, cout, k, _, _ in specs)
    n_bn = sum(2 * cout for _, _, cout, _, _, _ in specs)
    # torchvision inception_v3(aux_logits=False): 21 785 568 parameters without the 1000-way fc (2048*1000 + 1000)
    assert n_conv + n_bn == 21_785_568
    names = [s[0] for s in specs]
    assert names[0] == "Conv2d_1a_3x3" and "Mixed_6e.branch7x7dbl_5" in names and names[-1] == "Mixed_7c.branch_pool"
    by = {s[0]: s for s in specs}
    assert by["Mixed_5b.branch1x1"][1] == 192 and by["Mixed_5c.branch1x1"][1] == 256 and by["Mixed_5d.branch1x1"][1] == 288
    assert by["Mixed_6a.branch3x3"][1] == 288 and by["Mixed_6b.branch1x1"][1] == 768 and by["Mixed_7a.branch3x3_1"][1] == 768
    assert by["Mixed_7b.branch1x1"][1] == 1280 and by["Mixed_7c.branch1x1"][1] == 2048
    assert by["Mixed_6b.branch7x7_2"][3:] == ((1, 7), (1, 1), (0, 3)) and by["Mixed_7b.branch3x3_2b"][3:] == ((3, 1), (1, 1), (1, 0))


def test_cpu_restatement_runs_and_is_deterministic():
    from oracle.inception_oracle import inception_logits, random_state_dict
    sd = random_state_dict(0)
    x = torch.randn(1, 3, 299, 299, generator=torch.Generator().manual_seed(1))
    a, b = inception_logits(sd, x), inception_logits(sd, x)
    assert a.shape == (1, 100) and torch.equal(a, b) and bool(torch.isfinite(a).all()) and float(a.abs().max()) > 1e-3
