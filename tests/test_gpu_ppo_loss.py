"""GPU: the fused loss tail of the PPO update (csrc/mrca_ppo_loss.hip, policy_ops.ppo_loss) against the expression the
reference evaluates (model/ppo.py:172-185, model/net.py:72-80, model/utils.py:90-97) written in PyTorch: the five scalars and,
through autograd, the gradients with respect to mean, value and logstd -- including samples whose ratio sits outside the
clip range on either side, exactly on its edge, and with zero advantage (autograd's tie rules for min / clamp)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.build()
    from mrca import net, policy_ops
    return net, policy_ops


def _reference(net, mean, value, logstd, action, old_lp, adv, target, clip, vcoef, cent):
    import torch.nn.functional as F
    ls = logstd.expand_as(mean)
    lp = net.gaussian_logprob(action, mean, ls)
    entropy = (0.5 + net._HALF_LOG_2PI + ls).sum(-1).mean()
    log_ratio = lp - old_lp
    ratio = torch.exp(log_ratio)
    s1 = ratio * adv
    s2 = torch.clamp(ratio, 1 - clip, 1 + clip) * adv
    pl = -torch.min(s1, s2).mean()
    vl = F.mse_loss(value, target)
    return pl + vcoef * vl - cent * entropy, pl, vl, entropy, ((ratio - 1.0) - log_ratio).mean()


@pytest.mark.parametrize("n", [1, 63, 1024, 16384, 70001])
def test_ppo_loss_matches_the_pytorch_expression(ops, n):
    net, P = ops
    g = torch.Generator(device="cuda").manual_seed(n)
    dev = "cuda"
    mean = torch.stack([torch.rand(n, generator=g, device=dev), torch.rand(n, generator=g, device=dev) * 2 - 1], 1)
    logstd = torch.tensor([-0.3, 0.1], device=dev)
    action = mean + torch.exp(logstd) * torch.randn(n, 2, generator=g, device=dev)
    value = torch.randn(n, 1, generator=g, device=dev) * 3
    target = value + torch.randn(n, 1, generator=g, device=dev)
    adv = torch.randn(n, 1, generator=g, device=dev)
    adv[::7] = 0.0                                                    # ties outside the clip range
    # old log-probabilities that put the ratio far inside, near / on the edges and far outside the clip range
    with torch.no_grad():
        lp_now = net.gaussian_logprob(action, mean, logstd.expand_as(mean))
    shift = torch.randn(n, 1, generator=g, device=dev) * 0.15
    shift[::5] = 0.0                                                  # ratio exactly 1
    old_lp = lp_now - shift
    clip, vcoef, cent = 0.1, 20.0, 5e-4
    leaves_a = [t.clone().requires_grad_(True) for t in (mean, value, logstd)]
    ref = _reference(net, *leaves_a, action, old_lp, adv, target, clip, vcoef, cent)
    ref[0].backward()
    leaves_b = [t.clone().requires_grad_(True) for t in (mean, value, logstd)]
    loss, stats = P.ppo_loss(*leaves_b, action, old_lp, adv, target, clip, vcoef, cent)
    loss.backward()
    for k, (want, name) in enumerate(zip(ref, ("loss", "policy loss", "value loss", "entropy", "kl"))):
        assert abs(float(stats[k]) - float(want)) <= 2e-6 * max(1.0, abs(float(want))), (name, float(stats[k]), float(want))
    assert float(loss) == float(stats[0])
    for a, b, name in zip(leaves_a, leaves_b, ("mean", "value", "logstd")):
        scale = float(a.grad.abs().max()) + 1e-12
        # per-sample gradients are the same closed forms: fp32 rounding only; logstd's is a sum over the batch
        tol = 2e-6 if name != "logstd" else 2e-5
        assert float((a.grad - b.grad).abs().max()) <= tol * scale, (name, float((a.grad - b.grad).abs().max()), scale)
    # bit-identical from run to run (fixed summation order), and the scratch is left ready for the next launch
    loss2, stats2 = P.ppo_loss(*[t.detach() for t in leaves_b], action, old_lp, adv, target, clip, vcoef, cent)
    assert torch.equal(stats, stats2)


def test_ppo_loss_clip_edge_cases_route_the_gradient_like_autograd(ops):
    """Ratios constructed to fall exactly on 1 - c, 1 + c and 1 (both surrogates equal) and clearly outside with either
    sign of the advantage: d loss / d mean must equal autograd's, sample by sample."""
    net, P = ops
    dev = "cuda"
    clip = 0.25                                        # 0.75 and 1.25 are exact in fp32
    ratios = torch.tensor([1.0, 0.75, 1.25, 0.5, 2.0, 0.5, 2.0, 1.0, 0.75, 1.25], device=dev)
    adv = torch.tensor([1.0, 1.0, 1.0, 1.0, 1.0, -1.0, -1.0, 0.0, -2.0, -2.0], device=dev).view(-1, 1)
    n = ratios.numel()
    mean = torch.full((n, 2), 0.25, device=dev)
    logstd = torch.zeros(2, device=dev)
    action = mean + torch.tensor([[0.5, -0.25]], device=dev)
    with torch.no_grad():
        lp = net.gaussian_logprob(action, mean, logstd.expand_as(mean))
    old_lp = lp - torch.log(ratios).view(-1, 1)
    value = torch.zeros(n, 1, device=dev)
    target = torch.ones(n, 1, device=dev)
    la = [t.clone().requires_grad_(True) for t in (mean, value, logstd)]
    _reference(net, *la, action, old_lp, adv, target, clip, 20.0, 5e-4)[0].backward()
    lb = [t.clone().requires_grad_(True) for t in (mean, value, logstd)]
    P.ppo_loss(*lb, action, old_lp, adv, target, clip, 20.0, 5e-4)[0].backward()
    # exp(log(r)) may land one ulp off r: samples whose autograd gradient is exactly 0 or full must agree in KIND
    za, zb = la[0].grad.abs().sum(1) == 0, lb[0].grad.abs().sum(1) == 0
    interior = torch.tensor([True, False, False, True, True, True, True, True, False, False], device=dev)   # not on an edge
    assert torch.equal(za[interior], zb[interior])
    assert float((la[0].grad - lb[0].grad)[interior].abs().max()) < 1e-7
    assert float((la[1].grad - lb[1].grad).abs().max()) < 1e-7
