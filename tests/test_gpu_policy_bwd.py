"""GPU: the backward kernel of the policy's conv front end (csrc/mrca_policy_bwd.hip) against torch.autograd through
the stock Conv1d -> ReLU -> Conv1d -> ReLU layers of the same CNNPolicy (model/net.py:19-25,37-49).

Tolerance: a weight gradient is a sum of N x 128 (conv2) / N x 255 (conv1) fp32 products; the kernel and MIOpen add them
in different orders.  Small batches are compared with a float64 evaluation on the CPU to 2e-5 of the gradient's largest
entry; the 4096- and 16 384-sample batches with the fp32 device result (MIOpen, another summation order: a sum of up to
4.2 million products in fp32) to 1e-3 of it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import util as U  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.build()


def _weights(seed):
    g = torch.Generator().manual_seed(seed)
    w1 = torch.randn(2, 32, 3, 5, generator=g) * 0.3
    b1 = torch.randn(2, 32, generator=g) * 0.1
    w2 = torch.randn(2, 32, 32, 3, generator=g) * 0.1
    b2 = torch.randn(2, 32, generator=g) * 0.1
    return w1, b1, w2, b2


def _autograd(x, w1, b1, w2, b2, gfeat, dtype, device):
    x, gfeat = x.to(device, dtype), gfeat.to(device, dtype)
    ps = [p.to(device, dtype).requires_grad_(True) for p in (w1, b1, w2, b2)]
    feats = []
    for t in range(2):
        h = torch.relu(F.conv1d(x, ps[0][t], ps[1][t], stride=2, padding=1))
        feats.append(torch.relu(F.conv1d(h, ps[2][t], ps[3][t], stride=2, padding=1)).flatten(1))
    feat = torch.stack(feats)
    feat.backward(gfeat)
    return feat.detach(), [p.grad for p in ps]


@pytest.mark.parametrize("n", [1, 2, 5, 64, 513])
def test_backward_equals_float64_autograd(n):
    from mrca import policy_ops
    w1, b1, w2, b2 = _weights(n)
    g = torch.Generator().manual_seed(100 + n)
    x = torch.rand(n, 3, 512, generator=g) - 0.5
    gfeat = torch.randn(2, n, 4096, generator=g)
    _, want = _autograd(x, w1, b1, w2, b2, gfeat, torch.float64, "cpu")
    c = lambda t: t.cuda().contiguous()          # noqa: E731
    feat = policy_ops.lidar_features(c(x), c(w1), c(b1), c(w2), c(b2))
    got = policy_ops.lidar_features_backward(c(x), c(w1), c(b1), c(w2), feat, c(gfeat[0]), c(gfeat[1]))
    for name, a, b in zip(("dw1", "db1", "dw2", "db2"), got, want):
        scale = float(b.abs().max())
        err = float((a.cpu().double() - b).abs().max())
        assert scale > 0.1 and err < 2e-5 * scale, (name, n, err, scale)


@pytest.mark.parametrize("n", [4096, 16384])
def test_backward_equals_the_device_autograd_at_minibatch_size(n):
    from mrca import policy_ops
    w1, b1, w2, b2 = _weights(7)
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.rand(n, 3, 512, generator=g, device="cuda") - 0.5
    gfeat = torch.randn(2, n, 4096, generator=g, device="cuda") / n
    feat_ref, want = _autograd(x, w1, b1, w2, b2, gfeat, torch.float32, "cuda")
    c = lambda t: t.cuda().contiguous()          # noqa: E731
    feat = policy_ops.lidar_features(x, c(w1), c(b1), c(w2), c(b2))
    assert float((feat - feat_ref).abs().max()) < 1e-5
    got = policy_ops.lidar_features_backward(x, c(w1), c(b1), c(w2), feat, gfeat[0], gfeat[1])
    again = policy_ops.lidar_features_backward(x, c(w1), c(b1), c(w2), feat, gfeat[0], gfeat[1])
    for name, a, a2, b in zip(("dw1", "db1", "dw2", "db2"), got, again, want):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) < 1e-3 * scale, (name, n, float((a - b).abs().max()), scale)
        assert torch.equal(a, a2), name          # fixed-order reduction: bit-identical from run to run


def test_backward_one_hot_probes():
    """A single non-zero upstream gradient, scan sample and weight per tower must land in exactly the weight-gradient
    entries the convolution arithmetic says."""
    from mrca import policy_ops
    w1 = torch.zeros(2, 32, 3, 5)
    b1 = torch.zeros(2, 32)
    w2 = torch.zeros(2, 32, 32, 3)
    b2 = torch.zeros(2, 32)
    w1[0, 5, 1, 3], w2[0, 9, 5, 0] = 2.0, 3.0        # actor: ch 5 <- frame 1 tap 3 ; ch 9 <- ch 5 tap 0
    w1[1, 7, 2, 0], w2[1, 30, 7, 2] = 1.0, 1.0       # critic
    b2[:] = 0.5                                      # keep the second ReLU open everywhere
    b1[:] = 0.25
    x = torch.zeros(3, 3, 512)
    x[0, 1, 100], x[1, 2, 301], x[2, 0, 7] = 1.0, 1.0, -2.0
    gfeat = torch.zeros(2, 3, 4096)
    gfeat[0, 0, 9 * 128 + 26], gfeat[1, 1, 30 * 128 + 75], gfeat[0, 2, 3 * 128 + 127] = 1.0, -1.5, 2.0
    _, want = _autograd(x, w1, b1, w2, b2, gfeat, torch.float64, "cpu")
    c = lambda t: t.cuda().contiguous()          # noqa: E731
    feat = policy_ops.lidar_features(c(x), c(w1), c(b1), c(w2), c(b2))
    got = policy_ops.lidar_features_backward(c(x), c(w1), c(b1), c(w2), feat, c(gfeat[0]), c(gfeat[1]))
    for name, a, b in zip(("dw1", "db1", "dw2", "db2"), got, want):
        assert float((a.cpu().double() - b).abs().max()) < 1e-6, name
        assert int((b != 0).sum()) > 0, name


def test_policy_gradients_with_the_hip_front_end_equal_the_stock_gradients():
    """evaluate_actions -> PPO-like loss -> backward, through the stock layers (MIOpen), through lidar_features_fn, and
    in float64 on the CPU: the two fp32 device results differ from each other by summation order only.  The conv
    gradients of a real loss cancel heavily (sums of 2048 x 255 signed products whose total is ~1e-3 of the sum of their
    magnitudes), so neither fp32 result is better than ~1e-4 .. 1e-3 of the tensor: the kernel accumulates one sequential
    fp32 MFMA chain per wave over its items' 255 positions where MIOpen reduces in blocks (measured on conv1's weights,
    the worst tensor: 1.2e-3 vs MIOpen's 2.8e-4 of the largest entry).  Bars: see the loop.  The reference-run learner
    goldens (tests/test_golden_learner.py, 'cuda-fused' legs) bound what that means for the parameters after an update:
    1e-5."""
    import copy
    from mrca.net import CNNPolicy
    torch.manual_seed(5)
    pol = CNNPolicy(3, 2).cuda()
    with torch.no_grad():
        for q in pol.parameters():
            q.add_(0.05 * torch.randn_like(q))
    n = 2048
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(n, 3, 512, device="cuda", generator=g) - 0.5
    goal = torch.rand(n, 2, device="cuda", generator=g) * 10 - 5
    speed = torch.rand(n, 2, device="cuda", generator=g)
    act = torch.rand(n, 2, device="cuda", generator=g)
    adv = torch.randn(n, 1, device="cuda", generator=g)
    tgt = torch.randn(n, 1, device="cuda", generator=g)

    def grads_of(p, args):
        p.zero_grad()
        v, lp, ent = p.evaluate_actions(*args[:4])
        loss = -(torch.exp(lp) * args[4]).mean() + 20.0 * F.mse_loss(v, args[5]) - 5e-4 * ent
        loss.backward()
        return {k: q.grad.detach().double().cpu() for k, q in p.named_parameters()}

    args = (x, goal, speed, act, adv, tgt)
    pol.fused_train = False
    stock = grads_of(pol, args)
    pol.fused_train = True
    fused = grads_of(pol, args)
    ref_pol = copy.deepcopy(pol).cpu().double()
    ref_pol.fused_train = False
    ref = grads_of(ref_pol, tuple(a.cpu().double() for a in args))
    worst = {}
    for k in ref:
        # relative L2 error of the whole tensor, and its largest entry error relative to the largest entry.  The latter
        # is NOT an arithmetic bound: a pre-activation of fc1 / fc2 within rounding distance of 0 flips its ReLU between
        # two evaluations that differ by 1e-6 .. 1e-5, and ONE flipped (sample, unit) moves that unit's row of the weight
        # gradient by ~1/sqrt(N) of its magnitude (measured: 6e-3 on act_fc1.weight with N = 2048) -- in either fp32
        # path.  The L2 error averages such rows out.
        n2 = float(ref[k].norm())
        scale = float(ref[k].abs().max())
        l2_f, l2_s = float((fused[k] - ref[k]).norm()) / n2, float((stock[k] - ref[k]).norm()) / n2
        mx_f, mx_s = float((fused[k] - ref[k]).abs().max()) / scale, float((stock[k] - ref[k]).abs().max()) / scale
        worst[k] = (l2_f, l2_s, mx_f, mx_s)
    print("relative L2 error vs float64 (fused, stock), max-entry error (fused, stock):")
    for k, w in worst.items():
        print(f"  {k:22s} {w[0]:.1e} {w[1]:.1e}   {w[2]:.1e} {w[3]:.1e}")
    for k, w in worst.items():
        assert w[1] <= 1e-3 and w[0] <= 2e-3, (k, w)
        assert w[3] <= 3e-2 and w[2] <= 3e-2, (k, w)
    assert float(fused["act_fea_cv1.weight"].abs().max()) > 0 and float(fused["crt_fea_cv2.bias"].abs().max()) > 0


@pytest.mark.parametrize("n", [1, 7, 300, 4097])
def test_row_table_form_equals_the_gathered_form_bit_for_bit(n):
    """mrca_lidar_features_rows / _backward_rows read the stacks out of a frame store through a row table; the kernels do the
    same arithmetic on the same numbers as with the gathered [n, 3, 512] copy: features and gradients identical to the bit."""
    from mrca import policy_ops
    w1, b1, w2, b2 = (t.cuda().contiguous() for t in _weights(7 + n))
    g = torch.Generator().manual_seed(500 + n)
    store = (torch.rand(3 * n + 11, 512, generator=g) - 0.5).cuda()
    rows = torch.randint(0, store.shape[0], (n, 3), generator=g, dtype=torch.int32).cuda()        # any rows, repeats included
    table = policy_ops.FrameTable(store, rows)
    x = table.gather()
    assert x.shape == (n, 3, 512)
    feat_t = policy_ops.lidar_features(table, w1, b1, w2, b2)
    feat_x = policy_ops.lidar_features(x.contiguous(), w1, b1, w2, b2)
    assert torch.equal(feat_t, feat_x)
    ga, gc = torch.randn(n, 4096, generator=g).cuda(), torch.randn(n, 4096, generator=g).cuda()
    got_t = policy_ops.lidar_features_backward(table, w1, b1, w2, feat_t, ga, gc)
    got_x = policy_ops.lidar_features_backward(x.contiguous(), w1, b1, w2, feat_x, ga, gc)
    for name, a, b in zip(("dw1", "db1", "dw2", "db2"), got_t, got_x):
        assert torch.equal(a, b), name


def test_update_through_the_frame_store_equals_the_update_through_gathered_stacks():
    """ppo._ppo_epochs with a one-frame-per-tick buffer: the fused front end reads the minibatch's stacks through a FrameTable
    (FrameRows.lazy); the parameters after an update equal, bit for bit, those of the same update fed gathered stacks."""
    from mrca import net, ppo
    T, N = 6, 64
    g = torch.Generator().manual_seed(3)
    frames = (torch.rand(T + 2, N, 512, generator=g) - 0.5).cuda()
    fidx = torch.stack([torch.stack([torch.arange(t, t + 3) for _ in range(N)]) for t in range(T)]).cuda()      # [T, N, 3]
    fidx[2, 5] = torch.tensor([4, 4, 4], device="cuda")          # a robot that restarted: its stack repeats one frame
    rows = ppo.FrameRows(frames, fidx)
    goals, speeds = (torch.randn(T, N, 2, generator=g).cuda() for _ in range(2))
    actions = torch.rand(T, N, 2, generator=g).cuda()
    logprobs = (torch.randn(T, N, 1, generator=g) * 0.1 - 2.0).cuda()
    targets, advs = (torch.randn(T, N, 1, generator=g).cuda() for _ in range(2))
    batches = lambda n: list(torch.arange(n, device="cuda").flip(0).split(128))      # noqa: E731  (a fixed order, three minibatches)
    outs = []
    for lazy_allowed in (True, False):
        torch.manual_seed(11)
        policy = net.CNNPolicy(frames=3, action_space=2).cuda()
        policy.fused_train = True
        opt = torch.optim.Adam(policy.parameters(), lr=1e-3)
        obss = rows if lazy_allowed else rows.materialise()            # FrameRows | [T, N, 3, 512]
        memory = (obss, goals, speeds, actions, logprobs, targets, None, None, advs)
        ppo.ppo_update_stage1(policy, opt, 128, memory, epoch=2, num_step=T, num_env=N, frames=3, obs_size=512, act_size=2,
                              index_batches=batches)
        outs.append([p.detach().clone() for p in policy.parameters()])
        if lazy_allowed:
            assert rows.lazy, "the fused update did not switch the frame store to row tables"
    for a, b in zip(*outs):
        assert torch.equal(a, b)
