"""GPU: csrc/mrca_adam.hip (one Adam step on flat buffers, one launch) against torch.optim.Adam's single-tensor form, and
the learner's optimiser (ppo.FlatAdam) against torch.optim.Adam on the policy."""
import copy

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


@pytest.mark.parametrize("n", [4, 5, 7, 1024, 1027, 2172101])
def test_adam_step_kernel_equals_torch_adam(built, n):
    from mrca import policy_ops
    g = torch.Generator(device="cuda").manual_seed(n)
    p0 = torch.randn(n, device="cuda", generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=5e-5, foreach=False, fused=False)
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    lr = 5e-5
    for step in range(1, 8):
        grad = torch.randn(n, device="cuda", generator=g) * (10.0 ** ((step % 3) - 2))
        if step == 4:
            lr = 3e-4
            opt.param_groups[0]["lr"] = lr
        ref.grad = grad.clone()
        opt.step()
        policy_ops.adam_step(p, grad, m, v, lr, 0.9, 0.999, 1e-8, step)
        st = opt.state[ref]
        # (PyTorch's element-wise kernels are compiled with fp contraction, this library without: a running average a few
        # ulps apart after a few steps; m is a signed sum: measured against the largest entry)
        assert float((m - st["exp_avg"]).abs().max()) <= 1e-6 * float(st["exp_avg"].abs().max()), step
        assert torch.allclose(v, st["exp_avg_sq"], rtol=2e-6, atol=1e-6 * float(st["exp_avg_sq"].max())), step
        # a parameter of size ~1 moves by ~lr per step and is rounded to ITS ulp each time: agreement to a few ulps of the
        # parameter, and the accumulated movement to 1 %
        assert float((p - ref.detach()).abs().max()) <= 6e-7 * float(p.abs().max()), step
        assert float(((p - p0) - (ref.detach() - p0)).abs().max()) <= 0.01 * float((p - p0).abs().max()) + 1e-6, step
    with pytest.raises(ValueError):
        policy_ops.adam_step(p, grad[:-1], m, v, lr, 0.9, 0.999, 1e-8, 1)
    with pytest.raises(RuntimeError):
        policy_ops.adam_step(p, grad, m, v, lr, 0.9, 0.999, 1e-8, 0)          # steps count from 1


def test_flat_adam_on_the_policy_equals_torch_adam(built):
    from mrca import ppo
    from mrca.net import CNNPolicy
    torch.manual_seed(4)
    a = CNNPolicy(3, 2).cuda()
    b = copy.deepcopy(a)
    opt_a = torch.optim.Adam(a.parameters(), lr=5e-5, foreach=False, fused=False)
    fg = ppo.FlatGrads(b.parameters())
    opt_b = ppo.FlatAdam(fg, lr=5e-5)
    g = torch.Generator(device="cuda").manual_seed(0)
    start = [p.detach().clone() for p in a.parameters()]
    for step in range(5):
        x = torch.rand(64, 3, 512, device="cuda", generator=g) - 0.5
        goal, speed, action = (torch.rand(64, 2, device="cuda", generator=g) for _ in range(3))

        def loss_of(pol):
            v, lp, ent = pol.evaluate_actions(x, goal, speed, action)
            return (v ** 2).mean() - lp.mean() - 0.01 * ent
        opt_a.zero_grad()
        loss_of(a).backward()
        opt_a.step()
        fg.backward(loss_of(b))
        opt_b.step()
    for p, q, s in zip(a.parameters(), b.parameters(), start):
        moved = float((p.detach() - s).abs().max())
        assert moved > 0 and float((p.detach() - q.detach()).abs().max()) <= 0.02 * moved + 1e-7     # (the two nets see each other's rounding from step 2 on)
    # the rollout path's derived copies follow the re-seated parameters
    b.refresh_rollout_cache()
    assert torch.equal(b._rollout_cache()["fc1_b"].view(2, 256)[0], b.act_fc1.bias.detach())
