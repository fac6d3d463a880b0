"""CPU: ppo.FlatAdam (the learner's optimiser on flat buffers) against torch.optim.Adam, the reference's optimiser
(ppo_stage1.py:176), and FlatGrads.backward against loss.backward()."""
import copy

import pytest
import torch

import util as U  # noqa: F401
from mrca import ppo
from mrca.net import CNNPolicy


def _small_policy(seed):
    torch.manual_seed(seed)
    p = CNNPolicy(3, 2)
    with torch.no_grad():
        for q in p.parameters():
            q.add_(0.05 * torch.randn_like(q))
    return p


def _loss(policy, x, goal, speed, action):
    v, lp, ent = policy.evaluate_actions(x, goal, speed, action)
    return (v ** 2).mean() - lp.mean() - 0.01 * ent


def test_flat_adam_equals_torch_adam_over_many_steps():
    a = _small_policy(1)
    b = copy.deepcopy(a)
    g = torch.Generator().manual_seed(0)
    opt_a = torch.optim.Adam(a.parameters(), lr=5e-5)
    fg = ppo.FlatGrads(b.parameters())
    opt_b = ppo.FlatAdam(fg, lr=5e-5)
    before = [p.detach().clone() for p in b.parameters()]
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))         # re-seating keeps the values
    for step in range(6):
        x = torch.rand(16, 3, 512, generator=g) - 0.5
        goal, speed, action = torch.rand(16, 2, generator=g), torch.rand(16, 2, generator=g), torch.rand(16, 2, generator=g)
        if step == 3:                                 # the KL controller / --lr write the group's rate
            for o in (opt_a, opt_b):
                o.param_groups[0]["lr"] = 2e-4
        opt_a.zero_grad()
        _loss(a, x, goal, speed, action).backward()
        opt_a.step()
        fg.backward(_loss(b, x, goal, speed, action))
        for p, q in zip(a.parameters(), b.parameters()):
            assert torch.equal(p.grad, q.grad)        # autograd.grad + cat == backward() into fresh .grad tensors
        opt_b.step()
        for p, q in zip(a.parameters(), b.parameters()):
            assert torch.allclose(p, q, rtol=1e-6, atol=1e-9), step
    assert any(not torch.equal(p, q) for p, q in zip(before, b.parameters()))
    # every parameter is a slice of the optimiser's buffer, in the bucket's order
    off = 0
    for p in b.parameters():
        assert p.data_ptr() == opt_b.flat.data_ptr() + 4 * off
        off += p.numel()
    assert off == opt_b.flat.numel() == fg.flat.numel() == 2172101        # (the all-reduce bucket)


def test_flat_adam_state_dict_is_torch_adams():
    a = _small_policy(2)
    b = copy.deepcopy(a)
    c = copy.deepcopy(a)
    g = torch.Generator().manual_seed(1)
    batch = lambda: (torch.rand(8, 3, 512, generator=g) - 0.5, torch.rand(8, 2, generator=g), torch.rand(8, 2, generator=g),  # noqa: E731
                     torch.rand(8, 2, generator=g))
    fg = ppo.FlatGrads(b.parameters())
    flat = ppo.FlatAdam(fg, lr=1e-4)
    stock = torch.optim.Adam(a.parameters(), lr=1e-4)
    for _ in range(3):
        d = batch()
        fg.backward(_loss(b, *d))
        flat.step()
        stock.zero_grad()
        _loss(a, *d).backward()
        stock.step()
    # FlatAdam's state loads into torch.optim.Adam ...
    sd = flat.state_dict()
    other = torch.optim.Adam(c.parameters(), lr=123.0)
    other.load_state_dict(sd)
    assert other.param_groups[0]["lr"] == 1e-4
    c.load_state_dict(b.state_dict())
    # ... and torch.optim.Adam's into a fresh FlatAdam: both continue identically
    e = copy.deepcopy(a)
    fg2 = ppo.FlatGrads(e.parameters())
    flat2 = ppo.FlatAdam(fg2, lr=9.0)
    flat2.load_state_dict(stock.state_dict())
    assert flat2.steps == 3 and flat2.param_groups[0]["lr"] == 1e-4
    d = batch()
    other.zero_grad()
    _loss(c, *d).backward()
    other.step()
    fg.backward(_loss(b, *d))
    flat.step()
    fg2.backward(_loss(e, *d))
    flat2.step()
    stock.zero_grad()
    _loss(a, *d).backward()
    stock.step()
    for p, q, r, s in zip(b.parameters(), c.parameters(), e.parameters(), a.parameters()):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-9) and torch.allclose(r, s, rtol=1e-6, atol=1e-9)
        assert torch.allclose(p, s, rtol=1e-5, atol=1e-8)
    # a state of another shape is refused, and so is a parameter that left the buffer
    with pytest.raises(ValueError):
        flat.load_state_dict({"state": {}, "param_groups": [{"params": [0, 1]}]})
    next(b.parameters()).data = next(b.parameters()).data.clone()
    with pytest.raises(RuntimeError):
        flat.step()


def test_flat_grads_backward_fills_unused_parameters_with_zeros():
    lin = torch.nn.Linear(4, 3)
    unused = torch.nn.Parameter(torch.ones(5))
    fg = ppo.FlatGrads(list(lin.parameters()) + [unused])
    fg.flat.fill_(7.0)
    fg.backward(lin(torch.ones(2, 4)).sum())
    assert torch.equal(unused.grad, torch.zeros(5)) and torch.equal(lin.bias.grad, torch.full((3,), 2.0))
    assert torch.equal(lin.weight.grad, torch.full((3, 4), 2.0))
