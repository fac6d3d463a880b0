"""Behavioural pins of the oracle (and, through the host harness, of the product arithmetic).

The reference's only tests of this path are the stage_ros rostests; their assertions are restated
here as unit tests (stage_ros-add_pose_and_crash/test/cmdpose_tests.py:87-203, hztest.xml:14,18):
  * cmd_vel linear.x moves the robot along its heading only      (test_cmdvel_x)
  * cmd_vel angular.z changes yaw only                             (test_cmdvel_yaw)
  * cmd_pose teleports x, y, yaw exactly                           (test_pose)
  * one tick = 0.1 s of simulated time (10 Hz)                     (hztest)
plus hand-computed vectors for every rule in stage_world1.py:122-211."""
import numpy as np
import pytest

import util as U
from util import S, O


def _open_world(R=1, **kw):
    g = U.small_grid(cell=0.05, size=20.0)  # 20 x 20 m box, walls on the border only
    sc = S.stage1(num_worlds=1, robots_per_world=R, seed=1, grid=g)
    for k, v in kw.items():
        setattr(sc, k, v)
    return sc


def _both(sc):
    return [U.oracle_env(sc, np.float32), U.oracle_env(sc, np.float64), U.EmulEnv(sc)]


def _place(env, poses, goals):
    env.reset(None, np.asarray(poses, np.float32), np.asarray(goals, np.float32))


@pytest.mark.parametrize("which", [0, 1, 2])
def test_cmdvel_x_moves_along_heading_only(which):
    env = _both(_open_world())[which]
    _place(env, [[1.0, 2.0, 0.0]], [[8.0, 2.0]])
    for _ in range(30):                      # 3 s at v = 1 (cmdpose_tests.py:87-108)
        env.step(np.array([[1.0, 0.0]], np.float32))
    assert abs(env.pose[0, 0] - 4.0) < 1e-5 and env.pose[0, 1] == 2.0 and env.pose[0, 2] == 0.0
    assert env.speed_gt[0, 0] == 1.0 and env.speed_gt[0, 1] == 0.0


@pytest.mark.parametrize("which", [0, 1, 2])
def test_cmdvel_yaw_rotates_in_place(which):
    env = _both(_open_world())[which]
    _place(env, [[1.0, 2.0, 0.0]], [[8.0, 2.0]])
    for _ in range(30):                      # cmdpose_tests.py:112-133, angular.z = 0.25
        env.step(np.array([[0.0, 0.25]], np.float32))
    assert env.pose[0, 0] == 1.0 and env.pose[0, 1] == 2.0
    assert abs(env.pose[0, 2] - 0.75) < 1e-5


@pytest.mark.parametrize("which", [0, 1, 2])
def test_cmd_pose_teleports_exactly(which):
    env = _both(_open_world())[which]
    _place(env, [[-3.25, 4.5, 1.25]], [[0.0, 0.0]])
    assert tuple(env.pose[0]) == (np.float32(-3.25), np.float32(4.5), np.float32(1.25))
    assert env.t[0] == 1 and env.crashed[0] == 0


def test_heading_wraps_to_minus_pi_pi():
    for env in _both(_open_world()):
        _place(env, [[0.0, 0.0, 3.1]], [[5.0, 5.0]])
        env.step(np.array([[0.0, 1.0]], np.float32))     # 3.1 + 0.1 > pi
        assert -np.pi < env.pose[0, 2] < -3.0


def test_wall_range_and_normalisation():
    """Robot at the origin of a 20 m box facing +x: the forward beam enters the wall cell whose
    left edge is at x = 10 - 0.05 = 9.95 -> beyond range -> 6.0; move to x = 5: range 4.95."""
    for env in _both(_open_world()):
        _place(env, [[0.0, 0.0, 0.0]], [[1.0, 0.0]])
        assert env.scan.min() == 6.0 and np.all(env.obs == 0.5)          # scan/6 - 0.5
        _place(env, [[5.0, 0.0, 0.0]], [[1.0, 0.0]])
        mid = 255  # bearing -pi/2 + 255*pi/511 = -0.00307 rad (just right of straight ahead)
        assert abs(env.scan[0, mid] - 4.95 / np.cos(0.5 * np.pi / 511)) < 2e-4
        assert abs(env.scan[0, 0] - 6.0) < 1e-6           # beam 0 looks along -y: wall 9.95 m away
        assert np.allclose(env.obs[0, 2], env.scan[0] / 6.0 - 0.5, atol=1e-7)
        assert (env.obs[0, 0] == env.obs[0, 2]).all() and (env.obs[0, 1] == env.obs[0, 2]).all()


def test_beam_order_is_right_to_left():
    """bearing_i = -pi/2 + i*pi/511 (stageros.cpp:495-497): beam 0 points to the robot's right."""
    g = U.small_grid(cell=0.05, size=20.0, blocks=[(-0.5, -3.0, 0.5, -2.0)])   # box on the -y side
    sc = S.stage1(num_worlds=1, robots_per_world=1, grid=g)
    for env in (U.oracle_env(sc, np.float64), U.EmulEnv(sc)):
        _place(env, [[0.0, 0.0, 0.0]], [[1.0, 0.0]])
        assert abs(env.scan[0, 0] - 2.0) < 0.051 and env.scan[0, 511] == 6.0


def test_lidar_sees_other_robots_but_not_self():
    for env in _both(_open_world(R=2)):
        _place(env, [[0.0, 0.0, 0.0], [3.0, 0.0, 0.0]], [[0, 5], [3, 5]])
        fwd = env.scan[0, 255]
        assert abs(fwd - (3.0 - 0.22)) < 1e-3           # rear face of robot 1 (half length 0.22)
        assert env.scan[1].min() == 6.0                 # robot 1 looks away from robot 0


def test_fidelity_mode_lidar_sees_other_robots_through_the_raster():
    """collision_raster = 0.2 m (worlds/stage1.world:3): the lidar walks the raster the robots are mapped into -- the range
    of a beam that meets another robot is the ENTRY distance of the first 0.2 m cell holding a piece of its outline
    (quantised to the raster like Stage's ranges, up to one cell diagonal short of the analytic distance, never beyond
    it), a robot sharing the lidar's own cell reads 0, beams that miss every outline cell keep their map range."""
    sc = _open_world(R=3, collision_raster=0.2)
    exact = _open_world(R=3)
    for f in (np.float32, np.float64):
        env, ref = U.oracle_env(sc, f), U.oracle_env(exact, f)
        poses = [[0.03, 0.04, 0.0], [3.03, 0.04, 0.0], [-5.0, -7.0, 0.3]]
        goals = [[0, 5], [3, 5], [0, 0]]
        _place(env, poses, goals)
        _place(ref, poses, goals)
        fwd, ana = float(env.scan[0, 255]), float(ref.scan[0, 255])
        assert abs(ana - (3.0 - 0.22)) < 1e-3
        # the rear outline of robot 1 (x = 2.81) lies in the raster column [2.8, 3.0): entered at x = 2.8
        assert abs(fwd - (2.8 - 0.03)) < 2e-3 and ana - 0.2 * 1.4143 <= fwd <= ana + 1e-6
        hit_r, hit_a = env.scan[0] < 6.0, ref.scan[0] < 6.0
        assert hit_a.sum() >= 20 and (hit_r | ~hit_a).all()       # every analytic hit is a raster hit (the cells cover the outline)
        assert hit_r.sum() <= hit_a.sum() + 40                    # ... and the raster adds at most a cell's worth of beams
        assert (env.scan[0][hit_r & hit_a] <= ref.scan[0][hit_r & hit_a] + 1e-6).all()
        assert env.scan[1].min() == 6.0 and np.array_equal(env.scan[2], ref.scan[2])   # looking away / nobody in range: the map only
        # two robots whose outlines share the lidar's own cell: every beam the cull keeps starts inside a marked cell
        near = [[0.03, 0.04, 0.0], [0.41, 0.04, 0.0], [-5.0, -7.0, 0.3]]      # rear face at x = 0.19: the lidar's own column
        _place(env, near, goals)
        assert env.scan[0, 255] == 0.0


def test_two_robots_head_on_collide_in_robot_order():
    """Gap between footprints 0.06 m, both advance 0.05 m: robot 0 moves first (free), robot 1
    then overlaps robot 0's NEW pose -> reverts and stalls; next tick robot 0 stalls too."""
    for env in _both(_open_world(R=2)):
        _place(env, [[0.0, 0.0, 0.0], [0.5, 0.0, np.pi]], [[9, 0], [-9, 0]])
        a = np.array([[0.5, 0.0], [0.5, 0.0]], np.float32)
        env.step(a)
        assert env.crashed.tolist() == [0, 1] or env.first_result.tolist() == [0, 2]
        assert env.first_result[1] == O.RESULT_CRASH and env.first_result[0] == 0


def test_wall_hit_reverts_pose_and_stalls():
    sc = _open_world(auto_reset=S.AUTO_NONE)
    for env in _both(sc):
        _place(env, [[9.70, 0.0, 0.0]], [[0.0, 0.0]])    # nose at 9.92, wall cells start at 9.95
        env.step(np.array([[1.0, 0.0]], np.float32))     # nose would reach 10.02 -> hit
        assert env.crashed[0] == 1 and abs(env.pose[0, 0] - 9.70) < 1e-6
        assert env.speed_gt[0, 0] == 0.0                 # finite-difference GT velocity of a stalled robot
        assert env.result[0] == O.RESULT_CRASH and abs(env.reward[0] - (-15.0)) < 1e-6


def test_reward_rules_stage1():
    sc = _open_world(auto_reset=S.AUTO_NONE)
    for env in _both(sc):
        _place(env, [[0.0, 0.0, 0.0]], [[5.0, 0.0]])
        env.step(np.array([[1.0, 0.0]], np.float32))
        assert abs(env.reward[0] - 0.25) < 1e-5          # 2.5 * 0.1 m progress (stage_world1.py:187)
        env.step(np.array([[0.0, 1.06]], np.float32))    # |w| > 1.05 -> -0.1*|w| (stage_world1.py:203-204)
        assert abs(env.reward[0] - (-0.106)) < 1e-5
        env.step(np.array([[0.0, 1.05]], np.float32))
        assert abs(env.reward[0]) < 1e-6
        _place(env, [[4.45, 0.0, 0.0]], [[5.0, 0.0]])
        env.step(np.array([[1.0, 0.0]], np.float32))     # dist 0.45 < 0.5 -> +15, Reach Goal
        assert env.done[0] == 1 and env.result[0] == O.RESULT_REACH and abs(env.reward[0] - 15.0) < 1e-6


def test_timeout_at_step_151_stage1():
    sc = _open_world(auto_reset=S.AUTO_NONE)
    env = U.oracle_env(sc, np.float32)
    em = U.EmulEnv(sc)
    for e in (env, em):
        _place(e, [[0.0, 0.0, 0.0]], [[9.0, 9.0]])
        for k in range(151):
            e.step(np.array([[0.0, 0.1]], np.float32))
            assert (e.done[0] == 1) == (k == 150)        # t > 150 first true on the 151st call
        assert e.result[0] == O.RESULT_TIMEOUT


def test_stage2_quirk_first_step_pre_distance_zero():
    sc = _open_world(pre_dist_zero=True, auto_reset=S.AUTO_NONE)
    for env in _both(sc):
        _place(env, [[0.0, 0.0, 0.0]], [[4.0, 3.0]])
        env.step(np.array([[0.0, 0.5]], np.float32))     # no translation: r = 2.5*(0 - 5) (stage_world2.py:170-171)
        assert abs(env.reward[0] - (-12.5)) < 1e-5


def test_goal_and_crash_same_tick_both_apply():
    """stage_world1.py:193-201: +15 and -15 add up; 'Crashed' overrides 'Reach Goal'."""
    sc = _open_world(auto_reset=S.AUTO_NONE)
    for env in _both(sc):
        _place(env, [[9.70, 0.0, 0.0]], [[9.9, 0.0]])
        env.step(np.array([[1.0, 0.0]], np.float32))
        assert env.result[0] == O.RESULT_CRASH and abs(env.reward[0]) < 1e-6 and env.done[0] == 1


@pytest.mark.parametrize("hold", [False, True])
def test_finished_robot_waiting_for_its_group(hold):
    """ppo_stage2.py:72-74 sends no cmd_vel for a robot whose episode is over; stageros keeps the last SetSpeed
    (stageros.cpp:272-280) and its watchdog is global (:466-471), so under Stage that robot DRIVES ON.  Default: it
    idles.  hold_velocity: it keeps its last command, stays an obstacle on the move, its stale
    (reward, done) keep being reported -- and the odom twist survives the restart's teleport."""
    sc = _open_world(R=2, auto_reset=S.AUTO_GROUP, hold_velocity=hold, pre_dist_zero=True)
    for env in _both(sc):
        _place(env, [[0.0, 0.0, 0.0], [0.0, 5.0, 0.0]], [[0.55, 0.0], [8.0, 5.0]])
        env.step(np.array([[1.0, 0.25], [1.0, 0.0]], np.float32))          # robot 0: 0.45 m from its goal -> Reach Goal
        assert env.done[0] == 1 and env.result[0] == O.RESULT_REACH and env.live[0] == 0 and env.live[1] == 1
        r0, x0 = float(env.reward[0]), float(env.pose[0, 0])
        for _ in range(5):
            env.step(np.array([[0.3, -0.9], [1.0, 0.0]], np.float32))      # whatever the policy says for robot 0 is ignored
        assert env.live[0] == 0 and env.done[0] == 1 and float(env.reward[0]) == r0
        if hold:
            assert float(env.pose[0, 0]) > x0 + 0.4 and abs(float(env.pose[0, 2]) - 6 * 0.025) < 1e-5
            assert np.allclose(env.speed[0], [1.0, 0.25]) and np.allclose(env.speed_gt[0], [1.0, 0.25])
        else:
            assert float(env.pose[0, 0]) == x0 and np.allclose(env.speed[0], [0.0, 0.0])
        # robot 1 reaches its goal too: the group restarts, both robots begin episode 2
        ep = int(env.episode[0])
        for _ in range(80):
            env.step(np.array([[0.0, 0.0], [1.0, 0.0]], np.float32))
            if env.episode[0] > ep:
                break
        assert env.episode[0] == ep + 1 and env.episode[1] == ep + 1 and env.live.all() and env.t[0] == 1
        if hold:
            assert np.allclose(env.speed[0], [1.0, 0.25]) and np.allclose(env.speed[1], [1.0, 0.0])
        else:
            assert np.allclose(env.speed, 0.0)


def test_local_goal_transform():
    for env in _both(_open_world()):
        _place(env, [[1.0, 1.0, np.pi / 2]], [[1.0, 4.0]])
        assert abs(env.local_goal[0, 0] - 3.0) < 1e-5 and abs(env.local_goal[0, 1]) < 1e-5


def test_reset_distributions_stage1():
    sc = S.stage1(num_worlds=16, robots_per_world=32, seed=3)
    env = U.oracle_env(sc, np.float32)
    env.reset()
    r = np.hypot(env.pose[:, 0], env.pose[:, 1])
    assert r.max() <= 9.0 and (np.abs(env.pose[:, 2]) <= np.pi + 1e-6).all()   # stage_world1.py:251-260
    dg = np.hypot(env.goal[:, 0] - env.pose[:, 0], env.goal[:, 1] - env.pose[:, 1])
    assert dg.min() >= 8.0 - 1e-5 and dg.max() <= 10.0 + 1e-5                  # stage_world1.py:262-274
    assert np.hypot(env.goal[:, 0], env.goal[:, 1]).max() <= 9.0
    assert np.allclose(env.prev_dist, dg, atol=1e-6)
    # uniform in the disc: mean radius 6, area fraction inside r<4.5 is 1/4
    assert abs(r.mean() - 6.0) < 0.3 and abs((r < 4.5).mean() - 0.25) < 0.06


def test_reset_tables_stage2_and_circle():
    sc = S.stage2(num_worlds=1, seed=1)
    env = U.oracle_env(sc, np.float32)
    env.reset()
    assert np.allclose(env.pose[0], [-7.0, 11.5, np.pi], atol=1e-6)            # model/utils.py:42
    assert np.allclose(env.goal[0], [-18.0, 11.5])                             # model/utils.py:55
    assert (env.pose[34:, 0] >= 9).all() and (env.pose[34:, 0] <= 19).all()    # stage_world2.py:252
    y = env.pose[34:, 1]
    assert (((y <= -1) & (y >= -5)) | ((y <= -13) & (y >= -19))).all()         # stage_world2.py:253-257
    assert (env.prev_dist == 0).all()
    scc = S.circle(num_worlds=1)
    ec = U.oracle_env(scc, np.float32)
    ec.reset()
    assert np.allclose(ec.goal, -ec.pose[:, :2], atol=1e-6)                    # antipodal (model/utils.py:6-38)
    assert np.allclose(np.hypot(ec.pose[:, 0], ec.pose[:, 1]), 25.0, atol=0.01)


def test_region_sampled_start_keeps_away_from_the_world_file_pose_in_the_first_episode():
    """stage_world2.py:250-268: a start sampled in Stage-2's region is re-drawn until it is 7 m from the robot's CURRENT
    position -- in the first episode that is its world-file pose (worlds/stage2.world; the table's rows 34..43), which
    lies INSIDE the region.  (Rounds 1-2 started every robot at the origin, 9 m from the region: the first draw always
    stood.)"""
    sc = S.stage2(num_worlds=12, seed=3)
    table = np.asarray(sc.init_table, np.float64)
    for env in (U.oracle_env(sc, np.float32), U.EmulEnv(sc)):
        assert np.allclose(np.asarray(env.pose, np.float64).reshape(12, 44, 3)[:, 34:44], table[34:44], atol=1e-6)
        env.reset()
        p = np.asarray(env.pose, np.float64).reshape(12, 44, 3)
        d = np.hypot(p[:, 34:44, 0] - table[34:44, 0], p[:, 34:44, 1] - table[34:44, 1])
        assert d.min() >= 7.0 - 1e-4
        # 120 draws, half of the region is within 7 m of such a pose: without the rule some would be closer
        assert np.allclose(p[:, :34, :2], table[:34, :2], atol=1e-6)
