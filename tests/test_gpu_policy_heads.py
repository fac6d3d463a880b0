"""GPU: csrc/mrca_policy_heads.hip (the three output heads of the actor-critic in the PPO update, forward and backward)
against the PyTorch expression of model/net.py:47-55,61-63 evaluated in float64, and CNNPolicy.mean_value's fused_train
path against the stock layers."""
import pytest
import torch

import util as U  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def built():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.build()


def _reference(a, c, w1, b1, w2, b2, wc, bc):
    mean = torch.cat((torch.sigmoid(a @ w1.t() + b1), torch.tanh(a @ w2.t() + b2)), dim=-1)
    return mean, c @ wc.t() + bc


@pytest.mark.parametrize("n", [1, 2, 3, 65, 2048, 16384, 16385, 40001])
def test_policy_heads_forward_and_backward_against_float64(built, n):
    from mrca import policy_ops
    g = torch.Generator(device="cuda").manual_seed(n)
    flat = torch.randn(3 * 128 + 3 + 1, device="cuda", generator=g) * 0.2          # weight rows at odd offsets, as in the
    w1, w2, wc = (flat[1 + 128 * k: 1 + 128 * (k + 1)].view(1, 128) for k in range(3))      # optimiser's flat buffer
    b1, b2, bc = (flat[385 + k: 386 + k] for k in range(3))
    a = torch.relu(torch.randn(n, 128, device="cuda", generator=g))
    c = torch.relu(torch.randn(n, 128, device="cuda", generator=g))
    gm = torch.randn(n, 2, device="cuda", generator=g)
    gv = torch.randn(n, 1, device="cuda", generator=g)
    args = [t.clone().requires_grad_(True) for t in (a, c, w1, b1, w2, b2, wc, bc)]
    mean, value = policy_ops.policy_heads(*args)
    assert mean.shape == (n, 2) and value.shape == (n, 1)
    args64 = [t.detach().double().requires_grad_(True) for t in (a, c, w1, b1, w2, b2, wc, bc)]
    m64, v64 = _reference(*args64)
    assert float((mean.detach().double() - m64.detach()).abs().max()) < 5e-7
    assert float((value.detach().double() - v64.detach()).abs().max()) < 5e-6
    (mean * gm).sum().add((value * gv).sum()).backward()
    ((m64 * gm.double()).sum() + (v64 * gv.double()).sum()).backward()
    names = ("a", "c", "w_actor1", "b_actor1", "w_actor2", "b_actor2", "w_critic", "b_critic")
    for name, t, t64 in zip(names, args, args64):
        assert t.grad.shape == t.shape
        scale = max(1.0, float(t64.grad.abs().max()))
        assert float((t.grad.double() - t64.grad).abs().max()) <= 3e-6 * scale, (name, n)
    # bit-identical from run to run
    again = [t.detach().clone().requires_grad_(True) for t in args]
    m2, v2 = policy_ops.policy_heads(*again)
    (m2 * gm).sum().add((v2 * gv).sum()).backward()
    assert torch.equal(m2, mean) and torch.equal(v2, value)
    assert all(torch.equal(p.grad, q.grad) for p, q in zip(args, again))


def test_only_one_output_used(built):
    """A loss of the value alone (or of the mean alone) hands the backward one gradient of zeros or None."""
    from mrca import policy_ops
    g = torch.Generator(device="cuda").manual_seed(5)
    ws = [torch.randn(1, 128, device="cuda", generator=g).requires_grad_(True) for _ in range(3)]
    bs = [torch.randn(1, device="cuda", generator=g).requires_grad_(True) for _ in range(3)]
    a = torch.randn(100, 128, device="cuda", generator=g).requires_grad_(True)
    c = torch.randn(100, 128, device="cuda", generator=g).requires_grad_(True)
    mean, value = policy_ops.policy_heads(a, c, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2])
    value.sum().backward()
    assert float(a.grad.abs().max()) == 0.0 and float(ws[0].grad.abs().max()) == 0.0 and float(bs[1].grad.abs().max()) == 0.0
    assert torch.allclose(c.grad, ws[2].detach().expand(100, 128)) and abs(float(bs[2].grad) - 100.0) < 1e-4


def test_mean_value_with_the_head_kernels_equals_the_stock_layers(built):
    from mrca.net import CNNPolicy
    torch.manual_seed(9)
    pol = CNNPolicy(3, 2).cuda()
    with torch.no_grad():
        for q in pol.parameters():
            q.add_(0.05 * torch.randn_like(q))
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(300, 3, 512, device="cuda", generator=g) - 0.5
    goal, speed = torch.rand(300, 2, device="cuda", generator=g) * 4 - 2, torch.rand(300, 2, device="cuda", generator=g)
    grads = {}
    for fused in (False, True):
        pol.fused_train = fused
        pol.zero_grad(set_to_none=True)
        mean, value = pol.mean_value(x, goal, speed)
        (mean.sum() * 0.7 + (value ** 2).sum()).backward()
        grads[fused] = (mean.detach(), value.detach(), {k: p.grad.clone() for k, p in pol.named_parameters() if p.grad is not None})
    pol.fused_train = False
    assert float((grads[True][0] - grads[False][0]).abs().max()) < 1e-5 and float((grads[True][1] - grads[False][1]).abs().max()) < 1e-4
    for k, gref in grads[False][2].items():
        gf = grads[True][2][k]
        assert float((gf - gref).norm()) <= 2e-3 * float(gref.norm()) + 1e-6, k


@pytest.mark.parametrize("n", [1, 3, 257, 16384])
def test_heads_that_apply_fc2s_relu_themselves(built, n):
    """relu_inputs = 1: policy_heads(z_a, z_c) == policy_heads(relu(z_a), relu(z_c)) forward, and backward the gradients with
    respect to the PRE-activations (the ReLU's mask applied inside), bit for bit against the two-step form."""
    from mrca import policy_ops
    g = torch.Generator(device="cuda").manual_seed(7 * n)
    ws = [torch.randn(1, 128, device="cuda", generator=g) * 0.2 for _ in range(3)]
    bs = [torch.randn(1, device="cuda", generator=g) for _ in range(3)]
    za, zc = torch.randn(n, 128, device="cuda", generator=g), torch.randn(n, 128, device="cuda", generator=g)
    za[0, :4] = 0.0                                                  # relu'(0) = 0, as threshold_backward has it
    gm, gv = torch.randn(n, 2, device="cuda", generator=g), torch.randn(n, 1, device="cuda", generator=g)
    res = []
    for inside in (True, False):
        leaves = [t.clone().requires_grad_(True) for t in (za, zc, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2])]
        a, c = (leaves[0], leaves[1]) if inside else (torch.relu(leaves[0]), torch.relu(leaves[1]))
        mean, value = policy_ops.policy_heads(a, c, *leaves[2:], relu_inputs=inside)
        ((mean * gm).sum() + (value * gv).sum()).backward()
        res.append((mean.detach(), value.detach(), [t.grad for t in leaves]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for p, q in zip(res[0][2], res[1][2]):
        assert torch.equal(p, q)


@pytest.mark.parametrize("n", [1, 5, 16384])
def test_relu_cat_equals_relu_then_cat(built, n):
    from mrca import policy_ops
    g = torch.Generator(device="cuda").manual_seed(n)
    h1 = torch.randn(n, 256, device="cuda", generator=g)
    h1[0, :3] = 0.0
    goal, speed = torch.randn(n, 2, device="cuda", generator=g), torch.rand(n, 2, device="cuda", generator=g)
    gout = torch.randn(n, 260, device="cuda", generator=g)
    a = h1.clone().requires_grad_(True)
    b = h1.clone().requires_grad_(True)
    out = policy_ops.relu_cat(a, goal, speed)
    ref = torch.cat((torch.relu(b), goal, speed), dim=-1)
    assert out.shape == (n, 260) and torch.equal(out, ref)
    (out * gout).sum().backward()
    (ref * gout).sum().backward()
    assert torch.equal(a.grad, b.grad)
    with pytest.raises(ValueError):
        policy_ops.relu_cat(h1[:, :128], goal, speed)


@pytest.mark.parametrize("n", [1, 5, 1000, 16384])
def test_bias_gradients_formed_by_the_row_kernels(built, n):
    """relu_cat(h1_bias=...) and policy_heads(z_bias=...): the biases of the layers in front receive the column sums of dh1 / dz
    as their gradient -- equal to the float64 sums of the gradients the same kernels write (the kernels add fp32 per-thread partials,
    then float64), and everything else unchanged to the bit."""
    from mrca import policy_ops
    g = torch.Generator(device="cuda").manual_seed(31 * n)
    h1 = torch.randn(n, 256, device="cuda", generator=g)
    goal, speed = torch.randn(n, 2, device="cuda", generator=g), torch.rand(n, 2, device="cuda", generator=g)
    gout = torch.randn(n, 260, device="cuda", generator=g)
    a, b = h1.clone().requires_grad_(True), h1.clone().requires_grad_(True)
    bias = torch.zeros(256, device="cuda", requires_grad=True)
    (policy_ops.relu_cat(a, goal, speed, h1_bias=bias) * gout).sum().backward()
    (policy_ops.relu_cat(b, goal, speed) * gout).sum().backward()
    assert torch.equal(a.grad, b.grad)
    want = b.grad.double().sum(0)
    tol = 1e-6 * float(b.grad.abs().double().sum(0).max()) + 1e-7
    assert float((bias.grad.double() - want).abs().max()) <= tol

    ws = [torch.randn(1, 128, device="cuda", generator=g) * 0.2 for _ in range(3)]
    bs = [torch.randn(1, device="cuda", generator=g) for _ in range(3)]
    za, zc = torch.randn(n, 128, device="cuda", generator=g), torch.randn(n, 128, device="cuda", generator=g)
    gm, gv = torch.randn(n, 2, device="cuda", generator=g), torch.randn(n, 1, device="cuda", generator=g)
    res = []
    for with_bias in (True, False):
        leaves = [t.clone().requires_grad_(True) for t in (za, zc, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2])]
        zb = (torch.zeros(128, device="cuda", requires_grad=True), torch.zeros(128, device="cuda", requires_grad=True))
        mean, value = policy_ops.policy_heads(*leaves, relu_inputs=True, z_bias=zb if with_bias else None)
        ((mean * gm).sum() + (value * gv).sum()).backward()
        res.append(([t.grad for t in leaves], zb))
    for p, q in zip(res[0][0], res[1][0]):
        assert torch.equal(p, q)
    assert res[1][1][0].grad is None
    for k in range(2):
        dz = res[0][0][k]
        want = dz.double().sum(0)
        tol = 1e-6 * float(dz.abs().double().sum(0).max()) + 1e-7
        assert float((res[0][1][k].grad.double() - want).abs().max()) <= tol, k
