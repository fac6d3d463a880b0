#!/usr/bin/env python3
"""Test-side research harness (NOT product, never shipped; lives under tests/ because it drives the oracle): run
mrca.trainer.Stage1Trainer on the CPU against the C restatement of the oracle, to study the learning behaviour of the
PPO recipe without spending MI355X minutes.

The env is oracle/libmrca_oracle_c.so behind a tiny object with the VecStageWorld surface (numpy state viewed as
torch CPU tensors); the GAE kernel is replaced by the torch loop of tests/test_golden_learner.  Nothing under
rl-collision-avoidance_amd/ imports this file.

    python tests/cpu_train.py --stage 1 --worlds 1 --robots 24 --updates 300 --lr 5e-5
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402  (adds the package + oracle to sys.path)
from mrca import ppo, scenario, vec_env  # noqa: E402
from mrca.trainer import HParams, Stage1Trainer  # noqa: E402


class CpuEnv:
    """COracleEnv with torch views (the VecStageWorld surface the trainer uses)."""

    def __init__(self, sc):
        self.scenario = sc
        self.env = util.COracleEnv(sc)
        self.device = torch.device("cpu")
        self.N, self.R, self.W = sc.num_robots, sc.robots_per_world, sc.num_worlds
        for k in util.STATE_FIELDS + ["fresh"]:
            setattr(self, k, torch.from_numpy(getattr(self.env, k)))
        # the VecStageWorld surface keeps the stacks as a ring; the oracle's deque-ordered stacks ARE a ring whose newest
        # frame sits in the last slot
        self.scan_ring = self.obs                 # (normalised frames, not raw ranges:)
        self.ring_is_raw = False
        self.ring_head = torch.full((self.N,), sc.frames - 1, dtype=torch.uint8)

    def reset(self, mask=None, poses=None, goals=None):
        self.env.reset(None if mask is None else mask.numpy(), None if poses is None else poses.numpy(),
                       None if goals is None else goals.numpy())
        return self.obs, self.local_goal, self.speed

    def step(self, actions):
        self.env.step(actions.detach().numpy())
        return self.obs, self.local_goal, self.speed, self.reward, self.done, self.result


def _gae_cpu(rewards, values, last_value, dones, gamma, lam):
    T, N = rewards.shape
    v = torch.cat([values, last_value.view(1, N)], 0)
    nd = 1.0 - dones.float()
    targets = torch.zeros_like(rewards)
    gae = torch.zeros(N)
    for t in range(T - 1, -1, -1):
        delta = rewards[t] + gamma * v[t + 1] * nd[t] - v[t]
        gae = delta + gamma * lam * nd[t] * gae
        targets[t] = gae + v[t]
    return targets, targets - values


def install_cpu_gae():
    """Replace the HIP GAE kernel's binding by the torch loop above (this process only; tests use monkeypatch instead so
    that nothing leaks into GPU tests collected in the same session)."""
    vec_env.gae = _gae_cpu


def circle_eval(policy, circles=1, max_ticks=1500):
    from mrca import evaluate
    env = CpuEnv(scenario.circle(num_worlds=circles))
    return evaluate.circle_test(env, evaluate.cnn_policy_fn(policy), max_ticks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", type=int, default=1)
    ap.add_argument("--worlds", type=int, default=1)
    ap.add_argument("--robots", type=int, default=24)
    ap.add_argument("--updates", type=int, default=100)
    ap.add_argument("--lr", type=float, default=5e-5)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--epoch", type=int, default=None)
    ap.add_argument("--horizon", type=int, default=128)
    ap.add_argument("--kl-target", type=float, default=0.0)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--load", default=None)
    ap.add_argument("--save", default=None)
    ap.add_argument("--circle-every", type=int, default=0)
    ap.add_argument("--logstd-min", type=float, default=None)
    ap.add_argument("--kl-stop", type=float, default=0.0)
    ap.add_argument("--lr-max", type=float, default=1e-3)
    ap.add_argument("--max-grad-norm", type=float, default=0.0)
    a = ap.parse_args()
    install_cpu_gae()
    torch.set_num_threads(a.threads)
    os.environ.setdefault("OMP_NUM_THREADS", str(a.threads))
    if a.stage == 1:
        sc = scenario.stage1(num_worlds=a.worlds, robots_per_world=a.robots, seed=a.seed)
        hp = HParams()
    else:
        sc = scenario.stage2(num_worlds=a.worlds, seed=a.seed)
        hp = HParams(batch_size=512, epoch=4)
    hp.learning_rate, hp.horizon = a.lr, a.horizon
    if a.batch_size:
        hp.batch_size = a.batch_size
    if a.epoch:
        hp.epoch = a.epoch
    hp.kl_target, hp.kl_stop, hp.lr_max, hp.max_grad_norm = a.kl_target, a.kl_stop, a.lr_max, a.max_grad_norm
    hp.single_frame_buffer = True
    if a.logstd_min is not None:
        hp.logstd_min = a.logstd_min
    env = CpuEnv(sc)
    tr = Stage1Trainer(env, hp=hp, seed=a.seed, stage2=(a.stage == 2))
    if a.load:
        tr.policy.load_state_dict(torch.load(a.load))
    tr.start()
    for u in range(a.updates):
        t0 = time.perf_counter()
        cnt = np.zeros(4)
        for _ in range(hp.horizon):
            was_live = env.live.bool().clone()
            tr.tick()
            d = env.done.bool() & was_live
            r = env.result[d]
            for k in (1, 2, 3):
                cnt[k] += int((r == k).sum())
        tot = max(cnt.sum(), 1)
        extra = ""
        if getattr(tr, "last_kl", None) is not None:
            extra = f"  kl {tr.last_kl:.4f}  lr {tr.optimizer.param_groups[0]['lr']:.2e}"
        print(f"update {u + 1:4d}  {time.perf_counter() - t0:5.1f}s  episodes {int(tot):5d}  reach {cnt[1] / tot:.3f}  "
              f"crash {cnt[2] / tot:.3f}  timeout {cnt[3] / tot:.3f}  logstd {tr.policy.logstd.data.tolist()}{extra}",
              flush=True)
        if a.circle_every and (u + 1) % a.circle_every == 0:
            m = circle_eval(tr.policy)
            print(f"   circle: success {m['success_rate']:.2f} crash {m['crash_rate']:.2f} unfinished "
                  f"{m['unfinished_rate']:.2f} ticks {m['ticks_run']}", flush=True)
        if a.save and ((u + 1) % 20 == 0 or u + 1 == a.updates):
            torch.save(tr.policy.state_dict(), a.save)


if __name__ == "__main__":
    main()
