"""CPU gate: the product's per-lane arithmetic (csrc/mrca_device.h, the functions the gfx950
kernels inline) driven by host loops must agree BIT-FOR-BIT with the oracle's fp32 mode, flags
included, over whole episodes (resets, crashes, time-outs, group-synchronous episodes)."""
import numpy as np
import pytest

import util as U
from util import S


def _run(sc, steps, seed, check_every=1):
    o = U.oracle_env(sc)
    e = U.EmulEnv(sc)
    o.reset()
    e.reset()
    U.assert_state_equal(e, o, what=f"{sc.name} reset")
    rng = np.random.default_rng(seed)
    for k in range(steps):
        a = U.random_actions(rng, sc.num_robots)
        o.step(a)
        e.step(a)
        if k % check_every == 0 or k == steps - 1:
            U.assert_state_equal(e, o, what=f"{sc.name} step {k}")
    return o


def test_stage1_bit_exact():
    o = _run(S.stage1(num_worlds=2, robots_per_world=8, seed=11), 120, 3)
    assert o.episode.max() >= 2  # auto resets happened


def test_stage2_bit_exact_group_episodes():
    o = _run(S.stage2(num_worlds=1, seed=5), 215, 3, check_every=5)
    assert o.episode.max() >= 2  # at least one group restarted (timeout 200)


def test_circle_bit_exact():
    _run(S.circle(num_worlds=1, seed=2), 25, 4)


def test_explicit_reset_mask_and_overrides():
    sc = S.stage1(num_worlds=1, robots_per_world=6, seed=9)
    o = U.oracle_env(sc)
    e = U.EmulEnv(sc)
    o.reset()
    e.reset()
    mask = np.array([1, 0, 1, 0, 0, 1], np.uint8)
    poses = np.zeros((6, 3), np.float32)
    poses[:, 0] = np.arange(6) - 2.5
    poses[:, 2] = 0.3
    goals = np.tile(np.array([[4.0, 4.0]], np.float32), (6, 1))
    o.reset(mask, poses, goals)
    e.reset(mask, poses, goals)
    U.assert_state_equal(e, o, what="masked reset")
    assert (o.pose[[0, 2, 5], 0] == poses[[0, 2, 5], 0]).all()
