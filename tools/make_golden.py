#!/usr/bin/env python3
"""Generate golden vectors by RUNNING the reference's own importable Python (model/net.py,
model/ppo.py, model/utils.py) in this container.  Output: tests/golden/*.npz (small, committed).
The GPU box has no /root/reference, so tests only ever read the committed files.

Pins (SURVEY 8c "what CAN be imported as a live oracle"):
  gae.npz          generate_train_data            model/ppo.py:122-139
  filter.npz       get_filter_index               model/utils.py:65-78
  net.npz          CNNPolicy.forward / evaluate_actions with formula weights   model/net.py:16-80
  ppo_update.npz   ppo_update_stage1 / ppo_update_stage2 (CPU, .cuda() patched to identity), with
                   the minibatch index lists the reference's sampler drew   model/ppo.py:143-259
"""
import logging
import os
import sys
import tempfile

import numpy as np
import torch

REF = os.environ.get("MRCA_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def formula_state_dict(shapes):
    """Deterministic weights: tensor k, flat index i -> 0.05*sin(0.37*i + 1.3*k) (+ small logstd)."""
    sd = {}
    for k, (name, shape) in enumerate(shapes):
        n = int(np.prod(shape))
        w = 0.05 * np.sin(0.37 * np.arange(n, dtype=np.float64) + 1.3 * k)
        sd[name] = torch.from_numpy(w.reshape(shape).astype(np.float32))
    return sd


def main():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, REF)
    os.chdir(tempfile.mkdtemp())  # model/ppo.py:10-19 creates ./log/<host>/ppo.log on import
    import functools
    import builtins
    builtins.reduce = functools.reduce  # model/utils.py:85 uses the py2 builtin
    from model import net as rnet, ppo as rppo, utils as rutils

    rng = np.random.default_rng(2024)

    # ---- GAE
    T, N = 16, 5
    rewards = rng.normal(size=(T, N))
    values = rng.normal(size=(T, N, 1))
    last_v = rng.normal(size=(N, 1))
    dones = (rng.uniform(size=(T, N)) < 0.2)
    targets, advs = rppo.generate_train_data(rewards=rewards, gamma=0.99, values=values, last_value=last_v,
                                             dones=dones, lam=0.95)
    np.savez(os.path.join(OUT, "gae.npz"), rewards=rewards, values=values, last_value=last_v, dones=dones,
             targets=targets, advs=advs, gamma=0.99, lam=0.95)

    # ---- filter index
    d = rng.uniform(size=(12, 6)) < 0.5
    d[:, 3] = True
    d[-1, 1] = True
    d[0, 2] = True
    idx = np.asarray(rutils.get_filter_index(d), dtype=np.int64)
    np.savez(os.path.join(OUT, "filter.npz"), dones=d, index=idx)

    # ---- network
    pol = rnet.CNNPolicy(frames=3, action_space=2)
    shapes = [(k, tuple(v.shape)) for k, v in pol.state_dict().items()]
    sd = formula_state_dict(shapes)
    sd["logstd"] = torch.tensor([-0.3, 0.2])
    pol.load_state_dict(sd)
    B = 6
    x = torch.from_numpy(rng.uniform(-0.5, 0.5, size=(B, 3, 512)).astype(np.float32))
    goal = torch.from_numpy(rng.uniform(-5, 5, size=(B, 2)).astype(np.float32))
    speed = torch.from_numpy(rng.uniform(-1, 1, size=(B, 2)).astype(np.float32))
    action = torch.from_numpy(rng.uniform(-1, 1, size=(B, 2)).astype(np.float32))
    with torch.no_grad():
        v, logprob, entropy = pol.evaluate_actions(x, goal, speed, action)
        _v2, _a, _lp, mean = pol(x, goal, speed)
    np.savez(os.path.join(OUT, "net.npz"), keys=np.array([k for k, _ in shapes]),
             shapes=np.array([str(s) for _, s in shapes]), logstd=sd["logstd"].numpy(), x=x.numpy(),
             goal=goal.numpy(), speed=speed.numpy(), action=action.numpy(), value=v.numpy(),
             logprob=logprob.numpy(), entropy=entropy.numpy(), mean=mean.numpy())

    # ---- PPO updates on CPU: .cuda() -> identity, record the sampler's batches and the logged losses
    torch.Tensor.cuda = lambda self, *a, **k: self
    recorded = []

    class RecSampler(rppo.BatchSampler):
        def __iter__(self):
            for b in super().__iter__():
                recorded.append(list(b))
                yield b

    rppo.BatchSampler = RecSampler
    losses = []

    class H(logging.Handler):
        def emit(self, rec):
            losses.append([float(t) for t in rec.getMessage().split(",")])

    rppo.logger_ppo.addHandler(H())

    def run_update(stage2):
        recorded.clear()
        losses.clear()
        torch.manual_seed(7)
        pol = rnet.CNNPolicy(frames=3, action_space=2)
        sd = formula_state_dict(shapes)
        sd["logstd"] = torch.tensor([-0.3, 0.2])
        pol.load_state_dict(sd)
        opt = torch.optim.Adam(pol.parameters(), lr=5e-5)
        T, N = 8, 6
        r2 = np.random.default_rng(5 + stage2)
        obss = r2.uniform(-0.5, 0.5, size=(T, N, 3, 512))
        goals = r2.uniform(-5, 5, size=(T, N, 2))
        speeds = r2.uniform(-1, 1, size=(T, N, 2))
        actions = r2.uniform(-1, 1, size=(T, N, 2))
        logprobs = r2.normal(size=(T, N, 1)) * 0.1 - 2.0
        values = r2.normal(size=(T, N, 1))
        rewards = r2.normal(size=(T, N))
        dones = r2.uniform(size=(T, N)) < (0.5 if stage2 else 0.1)
        targets, advs = rppo.generate_train_data(rewards=rewards, gamma=0.99, values=values,
                                                 last_value=r2.normal(size=(N, 1)), dones=dones, lam=0.95)
        memory = (obss, goals, speeds, actions, logprobs, targets, values, rewards, advs)
        out = dict(obss=obss.astype(np.float32), goals=goals.astype(np.float32), speeds=speeds.astype(np.float32),
                   actions=actions.astype(np.float32), logprobs=logprobs.astype(np.float32), targets=targets,
                   advs=advs, dones=dones)
        if stage2:
            fidx = rutils.get_filter_index(dones)
            rppo.ppo_update_stage2(policy=pol, optimizer=opt, batch_size=8, memory=memory, filter_index=fidx,
                                   epoch=2, coeff_entropy=5e-4, clip_value=0.1, num_step=T, num_env=N, frames=3,
                                   obs_size=512, act_size=2)
            out["filter_index"] = np.asarray(fidx, np.int64)
        else:
            rppo.ppo_update_stage1(policy=pol, optimizer=opt, batch_size=16, memory=memory, epoch=2,
                                   coeff_entropy=5e-4, clip_value=0.1, num_step=T, num_env=N, frames=3,
                                   obs_size=512, act_size=2)
        out["batches"] = np.array([np.asarray(b, np.int64) for b in recorded], dtype=object)
        out["losses"] = np.asarray(losses, np.float64)
        new = pol.state_dict()
        out["param_sum"] = np.array([float(new[k].double().sum()) for k, _ in shapes])
        out["param_head"] = np.stack([np.pad(new[k].reshape(-1)[:4].double().numpy(), (0, max(0, 4 - new[k].numel())))
                                      for k, _ in shapes])
        out["param_delta_abs_sum"] = np.array([float((new[k].double() - sd[k].double()).abs().sum())
                                               for k, _ in shapes])
        return out

    s1 = run_update(False)
    s2 = run_update(True)
    np.savez_compressed(os.path.join(OUT, "ppo_update.npz"), **{f"s1_{k}": v for k, v in s1.items()},
                        **{f"s2_{k}": v for k, v in s2.items()}, allow_pickle=True)
    print("golden vectors written to", OUT, "| stage1 minibatches:", len(s1["batches"]), "| stage2 minibatches:",
          len(s2["batches"]), "filtered:", len(s2["filter_index"]))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
