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


def test_stage2_hold_velocity_bit_exact():
    """Fidelity switch: Stage keeps the last SetSpeed -- dead robots keep driving, the speed input survives the restart."""
    o = _run(S.stage2(num_worlds=1, seed=6, hold_velocity=True), 90, 2, check_every=5)      # first group restart: step 46
    assert o.episode.max() >= 2 and (o.live == 0).any()


def test_stage1_hold_velocity_bit_exact():
    _run(S.stage1(num_worlds=2, robots_per_world=8, seed=11, hold_velocity=True), 120, 3, check_every=3)


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


def test_dense_world_beam_culling_is_conservative():
    """64 robots in a 14 m arena: many neighbours at every range.  The product culls robot-robot
    lidar tests by bearing interval, the oracle tests every pair: results must stay bit-identical."""
    g = U.small_grid(cell=0.05, size=14.0, blocks=[(-0.6, -0.6, 0.6, 0.6)])
    sc = S.stage1(num_worlds=1, robots_per_world=64, seed=31, grid=g)
    o = U.oracle_env(sc)
    e = U.EmulEnv(sc)
    rng = np.random.default_rng(5)
    poses = np.stack([rng.uniform(-6, 6, 64), rng.uniform(-6, 6, 64), rng.uniform(-np.pi, np.pi, 64)], 1).astype(np.float32)
    poses[:8, :2] = [[2.0 + 0.5 * k, 2.0] for k in range(8)]          # a tight row: near-touching neighbours
    goals = rng.uniform(-6, 6, (64, 2)).astype(np.float32)
    o.reset(None, poses, goals)
    e.reset(None, poses, goals)
    U.assert_state_equal(e, o, what="dense reset")
    for k in range(12):
        a = U.random_actions(rng, 64)
        o.step(a)
        e.step(a)
        U.assert_state_equal(e, o, what=f"dense step {k}")
    assert (o.scan < 1.0).mean() > 0.01


def test_non_finite_actions_idle_the_robot():
    sc = S.stage1(num_worlds=1, robots_per_world=4, seed=2)
    o, e, c = U.oracle_env(sc), U.EmulEnv(sc), U.COracleEnv(sc)
    for env in (o, e, c):
        env.reset()
    a = np.array([[np.nan, 0.5], [0.5, np.inf], [-np.inf, np.nan], [0.3, 0.2]], np.float32)
    before = o.pose.copy()
    for env in (o, e, c):
        env.step(a)
    U.assert_state_equal(e, o, what="nan actions (emul)")
    U.assert_state_equal(c, o, what="nan actions (C oracle)")
    assert np.isfinite(o.pose).all() and np.isfinite(o.scan).all() and np.isfinite(o.reward).all()
    assert (o.pose[2] == before[2]).all()       # both components non-finite -> (0, 0): no motion
