"""Learner-side parity, pinned by golden vectors produced by RUNNING the reference's own
model/ppo.py, model/net.py, model/utils.py (tools/make_golden.py; tests/golden/*.npz).

Tolerances: GAE in float64 is exact to 1e-12; the policy forward is fp32 on both sides -> 1e-5;
after the 6-minibatch Adam update (lr 5e-5) parameters agree to 1e-6 absolute on the CPU.  Every learner test
also runs ON THE DEVICE (``cuda`` legs, marked gpu): same goldens, policy forward / evaluate to 1e-5, parameters
after the replayed minibatches to 1e-5 (MIOpen / rocBLAS sum in a different order than the CPU kernels the
goldens were produced with; Adam at lr 5e-5 keeps a sign flip of a ~0 gradient below 1e-4 -- see _check_update),
filter index exact, GAE kernel vs the reference's targets to 1e-4 (fp32 scan vs float64)."""
import os

import numpy as np
import pytest
import torch

DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]
# "cuda-fused": the same goldens with the conv front end evaluated AND differentiated by the hand-written HIP kernels
# (net.CNNPolicy.fused_train: csrc/mrca_policy.hip forward, csrc/mrca_policy_bwd.hip backward) instead of MIOpen
POLICY_DEVICES = DEVICES + [pytest.param("cuda-fused", marks=pytest.mark.gpu)]


def _need(device):
    if device.startswith("cuda") and not torch.cuda.is_available():
        pytest.skip("no GPU")


def _place(pol, device):
    """-> the torch device; switches the policy's HIP front end on for the 'cuda-fused' leg."""
    if device == "cuda-fused":
        import __graft_entry__ as g
        g.build()
        pol.fused_train = True
        device = "cuda"
    pol.to(device)
    return device

import util as U
from util import O

GOLD = os.path.join(U.ROOT, "tests", "golden")


def formula_state_dict(keys, shapes):
    sd = {}
    for k, (name, shape) in enumerate(zip(keys, shapes)):
        n = int(np.prod(shape))
        w = 0.05 * np.sin(0.37 * np.arange(n, dtype=np.float64) + 1.3 * k)
        sd[str(name)] = torch.from_numpy(w.reshape(shape).astype(np.float32))
    return sd


def _policy():
    from mrca.net import CNNPolicy
    g = np.load(os.path.join(GOLD, "net.npz"))
    keys = [str(k) for k in g["keys"]]
    shapes = [eval(str(s)) for s in g["shapes"]]
    pol = CNNPolicy(frames=3, action_space=2)
    own = {k: tuple(v.shape) for k, v in pol.state_dict().items()}
    assert own == dict(zip(keys, shapes)), "state_dict keys/shapes must equal the reference's (checkpoint compat)"
    sd = formula_state_dict(keys, shapes)
    sd["logstd"] = torch.from_numpy(g["logstd"])
    pol.load_state_dict(sd)
    return pol, g, keys, shapes, sd


def test_gae_matches_reference_generate_train_data():
    g = np.load(os.path.join(GOLD, "gae.npz"))
    T, N = g["rewards"].shape
    t, a = O.gae(g["rewards"], g["values"].reshape(T, N), g["last_value"].reshape(N), g["dones"],
                 float(g["gamma"]), float(g["lam"]), np.float64)
    assert np.abs(t - g["targets"]).max() < 1e-12 and np.abs(a - g["advs"]).max() < 1e-12
    t32, a32 = O.gae(g["rewards"], g["values"].reshape(T, N), g["last_value"].reshape(N), g["dones"],
                     float(g["gamma"]), float(g["lam"]), np.float32)
    assert np.abs(t32 - g["targets"]).max() < 1e-4  # fp32 mode (what the HIP kernel is bit-compared to)


@pytest.mark.gpu
def test_gae_kernel_matches_reference_generate_train_data():
    """The HIP GAE kernel against the targets / advantages the reference's generate_train_data produced."""
    _need("cuda")
    from mrca import ppo
    g = np.load(os.path.join(GOLD, "gae.npz"))
    T, N = g["rewards"].shape
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()  # noqa: E731
    t, a = ppo.generate_train_data(dev(g["rewards"], torch.float32), float(g["gamma"]),
                                   dev(g["values"].reshape(T, N), torch.float32),
                                   dev(g["last_value"].reshape(N), torch.float32), dev(g["dones"], torch.uint8),
                                   float(g["lam"]))
    assert np.abs(t.cpu().numpy() - g["targets"]).max() < 1e-4
    assert np.abs(a.cpu().numpy() - g["advs"]).max() < 1e-4


@pytest.mark.parametrize("device", DEVICES)
def test_filter_index_matches_reference_quirk_included(device):
    _need(device)
    from mrca import ppo
    g = np.load(os.path.join(GOLD, "filter.npz"))
    assert O.filter_index(g["dones"]) == list(g["index"])
    got = ppo.get_filter_index(torch.from_numpy(g["dones"].astype(np.uint8)).to(device))
    assert sorted(got.tolist()) == sorted(g["index"].tolist())


@pytest.mark.parametrize("device", POLICY_DEVICES)
def test_policy_forward_and_evaluate_match_reference(device):
    _need(device)
    pol, g, *_ = _policy()
    device = _place(pol, device)
    x, goal, speed, action = (torch.from_numpy(g[k]).to(device) for k in ("x", "goal", "speed", "action"))
    with torch.no_grad():
        v, logprob, entropy = pol.evaluate_actions(x, goal, speed, action)
        mean, v2 = pol.mean_value(x, goal, speed)
    assert np.abs(v.cpu().numpy() - g["value"]).max() < 1e-5
    assert np.abs(v2.cpu().numpy() - g["value"]).max() < 1e-5
    assert np.abs(mean.cpu().numpy() - g["mean"]).max() < 1e-5
    assert np.abs(logprob.cpu().numpy() - g["logprob"]).max() < 1e-5
    assert abs(float(entropy) - float(g["entropy"])) < 1e-6
    vs, a, lp, m = pol(x, goal, speed, generator=torch.Generator(device=device).manual_seed(0))
    assert a.shape == (6, 2) and lp.shape == (6, 1) and vs.shape == (6, 1)
    from mrca.net import gaussian_logprob
    assert torch.allclose(lp, gaussian_logprob(a, m, pol.logstd.expand_as(m)))


def _check_update(prefix, stage2, device="cpu"):
    """Replays the minibatch index lists the reference's sampler drew through our update and compares with what the
    reference's own ppo_update_stage{1,2} left behind.  CPU: 1e-6.  Device: 1e-5 on parameter heads / sums -- the
    first Adam step moves every parameter by lr * sign(g) whatever |g| is, so a parameter whose gradient is ~0
    (|g| below the reduction-order noise, ~1e-9) may move the other way on the device: at most 2 * lr per step,
    6 steps -> 6e-4 in the worst case for such a parameter; they are excluded by the |golden delta| test below,
    which bounds the aggregate movement to 2 % instead."""
    from mrca import ppo
    tol = 1e-6 if device == "cpu" else 1e-5
    pol, _g, keys, shapes, sd0 = _policy()
    device = _place(pol, device)
    g = np.load(os.path.join(GOLD, "ppo_update.npz"), allow_pickle=True)
    P = lambda k: g[f"{prefix}_{k}"]  # noqa: E731
    opt = torch.optim.Adam(pol.parameters(), lr=5e-5)
    T, N = P("obss").shape[:2]
    mem = tuple(torch.from_numpy(np.asarray(P(k), dtype=np.float32)).to(device) for k in
                ("obss", "goals", "speeds", "actions", "logprobs", "targets"))
    advs = torch.from_numpy(P("advs").astype(np.float32)).to(device)
    memory = (*mem, None, None, advs)
    batches = [torch.as_tensor(np.asarray(b, dtype=np.int64), device=device) for b in P("batches")]
    per_epoch = len(batches) // 2
    it = iter([batches[:per_epoch], batches[per_epoch:]])
    log = []
    kw = dict(policy=pol, optimizer=opt, memory=memory, epoch=2, coeff_entropy=5e-4, clip_value=0.1, num_step=T,
              num_env=N, frames=3, obs_size=512, act_size=2, index_batches=lambda n: next(it), log=log)
    if stage2:
        ppo.ppo_update_stage2(batch_size=8, filter_index=P("filter_index"), **kw)
    else:
        ppo.ppo_update_stage1(batch_size=16, **kw)
    losses = np.array([[float(x) for x in row] for row in log])
    assert np.abs(losses - P("losses")).max() < 2e-5, (losses, P("losses"))
    new = {k: v.cpu() for k, v in pol.state_dict().items()}
    for i, k in enumerate(keys):
        assert abs(float(new[k].double().sum()) - P("param_sum")[i]) < tol * max(1.0, new[k].numel() ** 0.5), k
        head = new[k].reshape(-1)[:4].double().numpy()
        assert np.abs(head - P("param_head")[i][: head.size]).max() < tol, k
        delta = float((new[k].double() - sd0[k].double()).abs().sum()) if k != "logstd" else None
        if delta is not None:
            want = P("param_delta_abs_sum")[i]
            assert abs(delta - want) <= 0.02 * want + 1e-7, (k, delta, want)


@pytest.mark.parametrize("device", POLICY_DEVICES)
def test_ppo_update_stage1_matches_reference(device):
    _need(device)
    _check_update("s1", False, device)


@pytest.mark.parametrize("device", POLICY_DEVICES)
def test_ppo_update_stage2_matches_reference(device):
    _need(device)
    _check_update("s2", True, device)
