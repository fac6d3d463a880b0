"""GPU: the fused rollout path of the policy (csrc/mrca_policy.hip + batched fc GEMMs) against the stock PyTorch fp32
layers of the same CNNPolicy (the reference architecture, model/net.py:16-80).  fp32 on both sides: 1e-5."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import util as U  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pol():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.build()
    from mrca.net import CNNPolicy
    torch.manual_seed(3)
    p = CNNPolicy(3, 2).cuda()
    with torch.no_grad():                      # not the symmetric default init: distinct, sizeable biases
        for q in p.parameters():
            q.add_(0.05 * torch.randn_like(q))
    return p


@pytest.mark.parametrize("n", [1, 2, 3, 7, 255, 4096, 5000])
def test_lidar_features_equal_the_pytorch_conv_stack(pol, n):
    from mrca import policy_ops
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.rand(n, 3, 512, device="cuda", generator=g) - 0.5
    rc = pol.refresh_rollout_cache()
    feat = policy_ops.lidar_features(x, rc["w1"], rc["b1"], rc["w2"], rc["b2"])
    assert feat.shape == (2, n, 4096)
    with torch.no_grad():
        for t, tw in enumerate(("act", "crt")):
            h = torch.relu(getattr(pol, f"{tw}_fea_cv1")(x))
            h = torch.relu(getattr(pol, f"{tw}_fea_cv2")(h)).flatten(1)
            err = float((feat[t] - h).abs().max())
            assert err < 1e-5, (tw, n, err)
            assert float(h.abs().max()) > 0.05          # the comparison is not between zeros


def test_lidar_features_asymmetric_probe(pol):
    """One-hot probes: a single non-zero scan sample and a single non-zero weight must land in exactly the outputs
    the convolution arithmetic says (catches transposed / shifted operand layouts that random data could average
    out)."""
    from mrca import policy_ops
    w1 = torch.zeros(2, 32, 3, 5, device="cuda")
    b1 = torch.zeros(2, 32, device="cuda")
    w2 = torch.zeros(2, 32, 32, 3, device="cuda")
    b2 = torch.zeros(2, 32, device="cuda")
    w1[0, 5, 1, 3] = 2.0          # actor: channel 5 <- frame 1, tap 3
    w2[0, 9, 5, 0] = 3.0          # actor: channel 9 <- channel 5, tap 0
    w1[1, 7, 2, 0] = 1.0          # critic: channel 7 <- frame 2, tap 0
    w2[1, 30, 7, 2] = 1.0
    x = torch.zeros(2, 3, 512, device="cuda")
    x[0, 1, 100] = 1.0
    x[1, 2, 301] = 1.0
    got = policy_ops.lidar_features(x, w1, b1, w2, b2)
    for t in range(2):
        h = torch.relu(F.conv1d(x, w1[t], b1[t], stride=2, padding=1))
        want = torch.relu(F.conv1d(h, w2[t], b2[t], stride=2, padding=1)).flatten(1)
        assert torch.equal(got[t], want), t
        assert int((want != 0).sum()) == 1


def test_mean_value_fused_equals_mean_value(pol):
    g = torch.Generator(device="cuda").manual_seed(0)
    n = 4096
    x = torch.rand(n, 3, 512, device="cuda", generator=g) - 0.5
    goal = torch.rand(n, 2, device="cuda", generator=g) * 20 - 10
    speed = torch.rand(n, 2, device="cuda", generator=g)
    pol.refresh_rollout_cache()
    with torch.no_grad():
        m0, v0 = pol.mean_value(x, goal, speed)
    m1, v1 = pol.mean_value_fused(x, goal, speed)
    assert float((m0 - m1).abs().max()) < 1e-5 and float((v0 - v1).abs().max()) < 1e-4 * max(1.0, float(v0.abs().max()))
    # the cache follows the parameters only when refreshed
    with torch.no_grad():
        pol.actor1.bias.add_(0.5)
    m2, _ = pol.mean_value_fused(x, goal, speed)
    assert torch.equal(m1, m2)
    pol.refresh_rollout_cache()
    m3, _ = pol.mean_value_fused(x, goal, speed)
    with torch.no_grad():
        m4, _ = pol.mean_value(x, goal, speed)
        pol.actor1.bias.sub_(0.5)
    assert float((m3 - m4).abs().max()) < 1e-5 and float((m3 - m1).abs().max()) > 1e-3
    pol.refresh_rollout_cache()


def test_lidar_features_rejects_other_geometries(pol):
    from mrca import policy_ops
    rc = pol.refresh_rollout_cache()
    with pytest.raises(ValueError):
        policy_ops.lidar_features(torch.zeros(4, 3, 256, device="cuda"), rc["w1"], rc["b1"], rc["w2"], rc["b2"])
