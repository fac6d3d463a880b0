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
    # the derived copies follow the parameters whoever changes them (stamped with the parameters' versions, rebuilt lazily
    # and in place): no explicit refresh between the in-place change and the next fused call
    addr = pol._rc["head_b"].data_ptr()
    with torch.no_grad():
        pol.actor1.bias.add_(0.5)
    m3, _ = pol.mean_value_fused(x, goal, speed)
    with torch.no_grad():
        m4, _ = pol.mean_value(x, goal, speed)
        pol.actor1.bias.sub_(0.5)
    assert float((m3 - m4).abs().max()) < 1e-5 and float((m3 - m1).abs().max()) > 1e-3
    assert pol._rc["head_b"].data_ptr() == addr
    m5, _ = pol.mean_value_fused(x, goal, speed)
    assert float((m5 - m1).abs().max()) < 1e-6          # (b + 0.5) - 0.5 is b up to one rounding


def test_lidar_features_rejects_other_geometries(pol):
    from mrca import policy_ops
    rc = pol.refresh_rollout_cache()
    with pytest.raises(ValueError):
        policy_ops.lidar_features(torch.zeros(4, 3, 256, device="cuda"), rc["w1"], rc["b1"], rc["w2"], rc["b2"])


def test_bf16_inference_is_bounded_against_fp32():
    """bf16 autocast inference is an opt-in of the rollout (`--bf16-inference`, bench `--policy-dtype bf16`; fp32 is the
    default and what every other test uses).  With the trained checkpoint on real circle-test observations: the action
    mean and the value must stay within bf16's three significant digits of the fp32 forward, and the circle test under
    bf16 inference is run and reported (the scenario is chaotic in the last bits, so its success rate is printed, not
    asserted beyond "most robots do not crash")."""
    import os
    from mrca import evaluate, ppo
    from mrca import scenario as S
    from mrca.net import CNNPolicy
    from mrca.vec_env import VecStageWorld
    ck = os.path.join(U.ROOT, "rl-collision-avoidance_amd", "mrca", "data", "policy_r02_stage2_circles.pth")
    pol = CNNPolicy(3, 2).cuda()
    pol.load_state_dict(torch.load(ck, map_location="cuda"))
    env = VecStageWorld(S.circle(num_worlds=20, seed=0))
    env.reset()
    worst_m, worst_v, sum_m, cnt = 0.0, 0.0, 0.0, 0
    for k in range(240):
        with torch.no_grad():
            m32, v32 = pol.mean_value(env.obs, env.local_goal, env.speed)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                m16, v16 = pol.mean_value(env.obs, env.local_goal, env.speed)
        worst_m = max(worst_m, float((m32 - m16.float()).abs().max()))
        sum_m += float((m32 - m16.float()).abs().mean())
        cnt += 1
        worst_v = max(worst_v, float(((v32 - v16.float()).abs() / (1.0 + v32.abs())).max()))
        lo, hi = ppo._bounds(evaluate.ACTION_BOUND, m32.device, m32.dtype)
        env.step(torch.minimum(torch.maximum(m32, lo), hi).contiguous())
    print(f"bf16 vs fp32 inference over 240 circle ticks x 1000 robots: max |d mean| {worst_m:.4f}, "
          f"(mean {sum_m / cnt:.5f}), max |d value| / (1 + |value|) {worst_v:.4f}")
    # measured: max 0.104, mean 0.0066, value 0.14 over 240 000 evaluations; circle SR under bf16 inference 1.00 -- a saturating sigmoid / tanh head amplifies bf16's 3 digits where
    # its input is large; the typical deviation is two orders of magnitude smaller
    assert worst_m < 0.25 and sum_m / cnt < 0.02 and worst_v < 0.3

    def bf16_policy(obs, goal, speed):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            mean, _v = pol.mean_value(obs, goal, speed)
        lo, hi = ppo._bounds(evaluate.ACTION_BOUND, obs.device, torch.float32)
        return torch.minimum(torch.maximum(mean.float(), lo), hi)
    env1 = VecStageWorld(S.circle(num_worlds=1, seed=0))
    m = evaluate.circle_test(env1, bf16_policy, max_ticks=1500)
    print("circle test under bf16 inference:", m)
    assert m["crash_rate"] < 0.5
    env.close()
    env1.close()


def test_front_end_reads_the_envs_frame_ring_in_place(pol):
    """The env keeps the last three scans of every robot as a ring of RAW ranges (one row written per tick); the kernel
    forms x / 6 - 0.5 and the deque order while it stages and must see exactly the stacks the materialised copy
    (env.obs, mrca_materialize) holds -- bit for bit, at every phase of the ring and across restarts (a restarted robot
    has all its slots rewritten).  The rollout buffer's row (mrca_newest_obs) and the newest-scan view likewise."""
    from mrca import policy_ops
    from mrca.vec_env import VecStageWorld
    from util import S
    env = VecStageWorld(S.stage1(num_worlds=8, robots_per_world=24, seed=2), device="cuda:0")
    env.reset()
    rc = pol.refresh_rollout_cache()
    g = torch.Generator(device="cuda").manual_seed(0)
    seen_heads = set()
    for k in range(12):
        a = torch.stack([torch.rand(env.N, generator=g, device="cuda"), torch.rand(env.N, generator=g, device="cuda") * 2 - 1], 1)
        env.step(a.contiguous())
        ring, head = env.policy_obs()
        assert head.raw and ring.data_ptr() == env.scan_ring.data_ptr()
        seen_heads |= set(head.slots.unique().tolist())
        via_ring = policy_ops.lidar_features(ring, rc["w1"], rc["b1"], rc["w2"], rc["b2"], head=head)
        via_copy = policy_ops.lidar_features(env.obs, rc["w1"], rc["b1"], rc["w2"], rc["b2"])
        assert torch.equal(via_ring, via_copy), k
        assert torch.equal(env.newest_frame(), env.obs[:, -1])
        # the views against the ring itself: newest scan, and torch's own x / 6 - 0.5 (the IEEE quotient minus one half)
        ar = torch.arange(env.N, device="cuda")
        assert torch.equal(env.scan, ring[ar, head.slots.long()])             # (ABI 6: the ring holds plain ranges)
        assert (ring >= 0).all() and not torch.signbit(ring).any()
        assert torch.equal(env.obs[:, -1], policy_ops.normalize_scans(env.scan))
        # ... which is the correctly rounded x / 6 minus one half (numpy on the host; torch's GPU `x / 6.0` multiplies by
        # RN(1/6) instead and differs in the last bit for one value in a few hundred)
        host = env.scan.cpu().numpy()
        assert np.array_equal(env.obs[:, -1].cpu().numpy(), host / np.float32(6.0) - np.float32(0.5))
        # a ring of NORMALISED frames (what a caller of ABI 3 hands over) still works: plain u8 heads
        via_norm = policy_ops.lidar_features(policy_ops.normalize_scans(ring), rc["w1"], rc["b1"], rc["w2"], rc["b2"], head=head.slots)
        assert torch.equal(via_norm, via_copy), k
        m0, v0 = pol.mean_value_fused(ring, env.local_goal, env.speed, head=head)
        m1, v1 = pol.mean_value_fused(env.obs, env.local_goal, env.speed)
        assert torch.equal(m0, m1) and torch.equal(v0, v1)
    assert seen_heads == {0, 1, 2}
    env.close()


@pytest.mark.parametrize("n", [1, 31, 33, 4096])
def test_act_fused_equals_the_stock_generate_action(pol, n):
    """The three-launch rollout inference (conv front end, batched fc1 GEMM, mrca_policy_tail) against the stock PyTorch
    layers with the SAME noise draws: value, unclipped action, log-probability and clipped action to fp32 summation
    order; and the deterministic mean action of generate_action_no_sampling."""
    from mrca import ppo
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.rand(n, 3, 512, device="cuda", generator=g) - 0.5
    goal = torch.rand(n, 2, device="cuda", generator=g) * 20 - 10
    speed = torch.rand(n, 2, device="cuda", generator=g)
    bound = ((0.0, -1.0), (1.0, 1.0))
    with torch.no_grad():
        pol.logstd.copy_(torch.tensor([-0.3, 0.2], device="cuda"))
    pol.refresh_rollout_cache()
    g0 = torch.Generator(device="cuda").manual_seed(77)
    g1 = torch.Generator(device="cuda").manual_seed(77)
    v0, a0, lp0, s0 = ppo.generate_action(pol, x, goal, speed, bound, g0, fused=False)
    v1, a1, lp1, s1 = ppo.generate_action(pol, x, goal, speed, bound, g1, fused=True)
    assert v1.shape == v0.shape == (n, 1) and a1.shape == (n, 2) and lp1.shape == lp0.shape == (n, 1) and s1.shape == (n, 2)
    assert float((v0 - v1).abs().max()) < 1e-4 * max(1.0, float(v0.abs().max()))
    assert float((a0 - a1).abs().max()) < 1e-5
    assert float((lp0 - lp1).abs().max()) < 1e-4
    assert float((s0 - s1).abs().max()) < 1e-5
    assert float(s1[:, 0].min()) >= 0.0 and float(s1[:, 0].max()) <= 1.0 and float(s1[:, 1].abs().max()) <= 1.0
    m0, sc0 = ppo.generate_action_no_sampling(pol, x, goal, speed, bound, fused=False)
    m1, sc1 = ppo.generate_action_no_sampling(pol, x, goal, speed, bound, fused=True)
    assert float((m0 - m1).abs().max()) < 1e-5 and float((sc0 - sc1).abs().max()) < 1e-5
    with torch.no_grad():
        pol.logstd.zero_()
    pol.refresh_rollout_cache()


@pytest.mark.parametrize("n", [1, 33, 4096])
def test_policy_tail_adds_the_fc1_bias_itself(pol, n):
    """mrca_policy_tail(h1 without bias, fc1_b) == mrca_policy_tail(h1 + bias, NULL), every output, bit for bit: the kernel
    adds the bias to the fp32 value the GEMM stored, which is what the separate add does."""
    from mrca import policy_ops
    g = torch.Generator(device="cuda").manual_seed(100 + n)
    rc = pol.refresh_rollout_cache()
    h1 = torch.randn(2, n, 256, device="cuda", generator=g)
    goal = torch.rand(n, 2, device="cuda", generator=g) * 20 - 10
    speed = torch.rand(n, 2, device="cuda", generator=g)
    noise = torch.randn(n, 2, device="cuda", generator=g)
    lo, hi = torch.tensor([0.0, -1.0], device="cuda"), torch.tensor([1.0, 1.0], device="cuda")
    rest = (goal, speed, rc["fc2_w"], rc["fc2_b"], rc["head_w"], rc["head_b"], rc["critic_w"], rc["critic_b"], rc["logstd"], noise, lo, hi)
    assert float(rc["fc1_b"].abs().min()) >= 0.0 and float(rc["fc1_b"].abs().max()) > 0.01
    with_bias = policy_ops.policy_tail((h1 + rc["fc1_b"]).contiguous(), *rest)
    in_kernel = policy_ops.policy_tail(h1, *rest, fc1_b=rc["fc1_b"])
    for a, b in zip(with_bias, in_kernel):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        policy_ops.policy_tail(h1, *rest, fc1_b=rc["fc1_b"][:, :, :128].contiguous())
