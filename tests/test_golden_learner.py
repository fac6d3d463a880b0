"""Learner-side parity, pinned by golden vectors produced by RUNNING the reference's own
model/ppo.py, model/net.py, model/utils.py (tools/make_golden.py; tests/golden/*.npz).

Tolerances: GAE in float64 is exact to 1e-12; the policy forward is fp32 on both sides -> 1e-5;
after the 6-minibatch Adam update (lr 5e-5) parameters agree to 1e-6 absolute."""
import os

import numpy as np
import torch

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


def test_filter_index_matches_reference_quirk_included():
    from mrca import ppo
    g = np.load(os.path.join(GOLD, "filter.npz"))
    assert O.filter_index(g["dones"]) == list(g["index"])
    got = ppo.get_filter_index(torch.from_numpy(g["dones"].astype(np.uint8)))
    assert sorted(got.tolist()) == sorted(g["index"].tolist())


def test_policy_forward_and_evaluate_match_reference():
    pol, g, *_ = _policy()
    x, goal, speed, action = (torch.from_numpy(g[k]) for k in ("x", "goal", "speed", "action"))
    with torch.no_grad():
        v, logprob, entropy = pol.evaluate_actions(x, goal, speed, action)
        mean, v2 = pol.mean_value(x, goal, speed)
    assert np.abs(v.numpy() - g["value"]).max() < 1e-5
    assert np.abs(v2.numpy() - g["value"]).max() < 1e-5
    assert np.abs(mean.numpy() - g["mean"]).max() < 1e-5
    assert np.abs(logprob.numpy() - g["logprob"]).max() < 1e-5
    assert abs(float(entropy) - float(g["entropy"])) < 1e-6
    vs, a, lp, m = pol(x, goal, speed, generator=torch.Generator().manual_seed(0))
    assert a.shape == (6, 2) and lp.shape == (6, 1) and vs.shape == (6, 1)
    from mrca.net import gaussian_logprob
    assert torch.allclose(lp, gaussian_logprob(a, m, pol.logstd.expand_as(m)))


def _check_update(prefix, stage2):
    from mrca import ppo
    pol, _g, keys, shapes, sd0 = _policy()
    g = np.load(os.path.join(GOLD, "ppo_update.npz"), allow_pickle=True)
    P = lambda k: g[f"{prefix}_{k}"]  # noqa: E731
    opt = torch.optim.Adam(pol.parameters(), lr=5e-5)
    T, N = P("obss").shape[:2]
    mem = tuple(torch.from_numpy(np.asarray(P(k), dtype=np.float32)) for k in
                ("obss", "goals", "speeds", "actions", "logprobs", "targets"))
    advs = torch.from_numpy(P("advs").astype(np.float32))
    memory = (*mem, None, None, advs)
    batches = [torch.as_tensor(np.asarray(b, dtype=np.int64)) for b in P("batches")]
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
    new = pol.state_dict()
    for i, k in enumerate(keys):
        assert abs(float(new[k].double().sum()) - P("param_sum")[i]) < 1e-6 * max(1.0, new[k].numel() ** 0.5), k
        head = new[k].reshape(-1)[:4].double().numpy()
        assert np.abs(head - P("param_head")[i][: head.size]).max() < 1e-6, k
        delta = float((new[k].double() - sd0[k].double()).abs().sum()) if k != "logstd" else None
        if delta is not None:
            want = P("param_delta_abs_sum")[i]
            assert abs(delta - want) <= 0.02 * want + 1e-7, (k, delta, want)


def test_ppo_update_stage1_matches_reference():
    _check_update("s1", False)


def test_ppo_update_stage2_matches_reference():
    _check_update("s2", True)
