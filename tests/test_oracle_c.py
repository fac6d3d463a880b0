"""The plain-C restatement of the oracle (oracle/mrca_oracle_c.c: CPU baseline + fast checker) must
agree bit-for-bit with the NumPy oracle's fp32 mode on every field."""
import numpy as np

import util as U
from util import S


def _run(sc, steps, seed, every=1):
    o = U.oracle_env(sc)
    c = U.COracleEnv(sc)
    o.reset()
    c.reset()
    U.assert_state_equal(c, o, what=f"{sc.name} reset")
    rng = np.random.default_rng(seed)
    for k in range(steps):
        a = U.random_actions(rng, sc.num_robots)
        o.step(a)
        c.step(a)
        if k % every == 0 or k == steps - 1:
            U.assert_state_equal(c, o, what=f"{sc.name} step {k}")


def test_c_oracle_stage1():
    _run(S.stage1(num_worlds=3, robots_per_world=8, seed=4), 100, 1)


def test_c_oracle_stage2_groups():
    _run(S.stage2(num_worlds=1, seed=6), 210, 2, every=7)


def test_c_oracle_stage2_hold_velocity():
    _run(S.stage2(num_worlds=1, seed=6, hold_velocity=True), 90, 2, every=6)      # first group restart at step 46


def test_c_oracle_fidelity_mode_with_the_raster_lidar():
    """Stage's own resolutions: 0.2 m map cells, robots collide when their outlines share a 0.2 m raster cell AND are seen
    by each other's lidar through that raster (the C oracle marks a window of cells per robot, the NumPy oracle walks the
    full set: bit-identical)."""
    _run(S.stage1(num_worlds=2, robots_per_world=12, seed=3, stage_resolution=True), 40, 5, every=4)
    _run(S.stage2(num_worlds=1, seed=2, stage_resolution=True), 24, 6, every=6)


def test_c_oracle_circle():
    _run(S.circle(num_worlds=1, seed=1), 15, 3)


def test_c_oracle_world_slice_matches_full_batch():
    sc = S.stage1(num_worlds=4, robots_per_world=6, seed=9)
    full = U.COracleEnv(sc)
    sl_sc = S.stage1(num_worlds=1, robots_per_world=6, seed=9)
    sl = U.COracleEnv(sl_sc, first_world=3)
    full.reset()
    sl.reset()
    rng = np.random.default_rng(0)
    for _ in range(20):
        a = U.random_actions(rng, sc.num_robots)
        full.step(a)
        sl.step(a[18:])
    assert (full.pose[18:].view(np.uint32) == sl.pose.view(np.uint32)).all()
    assert (full.scan[18:].view(np.uint32) == sl.scan.view(np.uint32)).all()
