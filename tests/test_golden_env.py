"""The Python half of the environment against golden vectors made by RUNNING the reference's own
stage_world1.py / stage_world2.py / circle_world.py (tools/make_golden_env.py -> tests/golden/env_python_*.npz):
observation, local goal, reward / terminal, goal-point bookkeeping, published messages, subscriber callbacks and the
distributions of the pose / goal samplers.

  oracle (fp64)  == reference to 1e-12          oracle (fp32)  == reference to the stated fp32 tolerances
  facade (mrca/stage_world.py) == reference     HIP env (-m gpu) == reference, robots teleported into the golden states

Tolerances of the fp32 legs: positions are up to 25 m, so one fp32 ulp is 1.9e-6 m; a distance is a sqrt of two
squared differences (<= 3 ulp), the progress reward multiplies a difference of two distances by 2.5.  Stated bars:
pose 1e-5, distance 1e-5, reward 3e-5, local goal 2e-5; flags must be equal wherever the reference's own float64
margin to the threshold is above 1e-5 (every hand-picked edge case is)."""
import os

import numpy as np
import pytest
from scipy import stats

import util as U
from util import S

GOLD = os.path.join(U.ROOT, "tests", "golden")
VARIANTS = ("stage1", "stage2", "circle")


def gold(name):
    return np.load(os.path.join(GOLD, f"env_python_{name}.npz"))


def wrap(th):
    return np.arctan2(np.sin(th), np.cos(th))


def one_robot_worlds(g, K):
    """K independent one-robot worlds in an open map with the variant's reward constants: every golden case gets a
    world of its own, so a spinning / driving case cannot touch anything."""
    return S.Scenario("golden", K, 1, S.empty_grid(), timeout=int(g["timeout"]), w_thresh=float(g["w_thresh"]),
                      pre_dist_zero=bool(g["pre_zero"]), auto_reset=S.AUTO_NONE,
                      reset_mode=np.full(1, S.RESET_TABLE, np.int32))


def check_reward_cases(g, pose, reward, done, result, dist, local_goal, fp32):
    ref_r, ref_d = g["rw_reward"], g["rw_dist"]
    if not fp32:
        tol = dict(pose=1e-12, dist=1e-12, reward=1e-11, lg=1e-11)
        safe = np.ones(len(ref_r), bool)
    else:
        tol = dict(pose=1e-5, dist=1e-5, reward=3e-5, lg=2e-5)
        w = np.abs(g["rw_speed_gt"][:, 1])
        safe = (np.abs(ref_d - 0.5) > 1e-5) & (np.abs(w - float(g["w_thresh"])) > 1e-6)
        assert safe.sum() >= len(safe) - 3
    dp = pose - g["rw_state"]
    dp[:, 2] = wrap(dp[:, 2])
    assert np.abs(dp).max() <= tol["pose"]
    assert np.abs(dist - ref_d).max() <= tol["dist"]
    assert np.array_equal(done.astype(bool)[safe], g["rw_term"][safe])
    assert np.array_equal(result.astype(np.int64)[safe], g["rw_result"].astype(np.int64)[safe])
    assert np.abs(reward - ref_r)[safe].max() <= tol["reward"]
    assert np.abs(local_goal - g["rw_local_goal"]).max() <= tol["lg"]
    # the cases really cover what they claim to
    assert {0, 1, 2, 3} <= set(g["rw_result"].tolist())
    assert (np.abs(ref_r) < 1e-9).sum() < len(ref_r) // 2


# ------------------------------------------------------------------------------------------------ oracle
@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_oracle_reward_terminal_local_goal_match_the_reference(variant, dtype):
    g = gold(variant)
    K = len(g["rw_t"])
    env = U.oracle_env(one_robot_worlds(g, K), dtype)
    env.reset(None, g["rw_pose0"], g["rw_goal"])
    env.prev_dist[:] = g["rw_prev"]
    env.t[:] = g["rw_t"]
    env.crashed[:] = g["rw_crashed"]
    env.step(g["rw_cmd"])
    assert np.array_equal(env.speed_gt.astype(np.float64), g["rw_speed_gt"])
    check_reward_cases(g, env.pose.astype(np.float64), env.reward.astype(np.float64), env.done, env.result,
                       env.prev_dist.astype(np.float64), env.local_goal.astype(np.float64), fp32=dtype is np.float32)


@pytest.mark.parametrize("variant", VARIANTS)
def test_oracle_observation_and_first_step_distance_match_the_reference(variant):
    g = gold(variant)
    # the observation of the oracle is the identity sub-sampling of stage_world1.py:126-139 (raw = beam_num = 512)
    scans = g["obs_scan"]
    finite = np.isfinite(scans).all(1)
    for dtype, tol in ((np.float64, 0.0), (np.float32, 6e-8)):
        f = dtype
        new = (scans[finite].astype(f) / f(6.0) - f(0.5)).astype(f)          # mrca_oracle.py:_observe
        assert np.abs(new.astype(np.float64) - g["obs_out_512"][finite]).max() <= tol
    # NaN / inf beams read 6.0 -> 0.5 in the reference (the ray cast never produces them: ranges are clamped to 6.0)
    bad = ~np.isfinite(scans)
    assert bad.any() and np.all(g["obs_out_512"][bad] == 0.5)
    # generate_goal_point: pre_distance = |local goal| in stage 1, 0 in stage 2 / circle
    idx = g["ep_index"]
    table = np.isfinite(g["ep_goal"]).all(1)
    sc = {"stage1": S.stage1(1, 24), "stage2": S.stage2(1), "circle": S.circle(1)}[variant]
    env = U.oracle_env(sc, np.float64)
    poses = np.zeros((sc.num_robots, 3))
    poses[:len(idx)] = g["ep_reset_pose"]
    poses[:, 2] = wrap(poses[:, 2])
    goals = np.zeros((sc.num_robots, 2))
    goals[:len(idx)] = g["ep_goal"]
    env.cfg.beams = 512
    env._begin_episode(np.arange(sc.num_robots), poses, goals)
    assert np.abs(env.prev_dist[:len(idx)] - g["ep_pre_distance"]).max() <= 1e-12
    if variant != "stage1":
        assert np.all(g["ep_pre_distance"] == 0) and np.all(g["ep_distance"] == 0)
        # table-driven robots: the pose the reference publishes and the goal it sets are the scenario's tables
        fixed = sc.reset_mode == S.RESET_TABLE
        n = len(idx)
        d = g["ep_reset_pose"][fixed[:n]] - sc.init_table[:n][fixed[:n]]
        d[:, 2] = wrap(d[:, 2])
        assert np.abs(d).max() <= 1e-12
        assert np.abs(g["ep_goal"][fixed[:n]] - sc.goal_table[:n][fixed[:n]]).max() == 0
    assert table.all()


# ------------------------------------------------------------------------------------------------ facade
class _FieldBackend:
    """Backend stand-in that serves whatever arrays the test puts in (the facade's getters are pure functions of them)."""

    def __init__(self, sc):
        N = sc.num_robots
        self.f = {"pose": np.zeros((N, 3)), "goal": np.zeros((N, 2)), "scan": np.full((N, 512), 6.0, np.float32),
                  "speed": np.zeros((N, 2)), "speed_gt": np.zeros((N, 2)), "crashed": np.zeros(N, np.int8),
                  "prev_dist": np.zeros(N), "reward": np.zeros(N)}

    def reset(self, mask, poses, goals):
        pass

    def step(self, actions):
        pass

    def field(self, name):
        return self.f[name]


def _facade_cls(variant):
    from mrca import stage_world as W
    return {"stage1": W.Stage1World, "stage2": W.Stage2World, "circle": W.CircleWorld}[variant]


@pytest.mark.parametrize("variant", VARIANTS)
def test_facade_getters_match_the_reference(variant):
    from mrca import stage_world as W
    g = gold(variant)
    n_env = {"stage1": 24, "stage2": 44, "circle": 50}[variant]
    W.set_backend_factory(_FieldBackend)
    try:
        cls = _facade_cls(variant)
        w = cls(512, 3, n_env)
        fields = w.world.backend.f
        # get_laser_observation, every beam_num, NaN / inf beams included: EXACT
        for bn in (512, 256, 128):
            w.beam_mum = bn
            for s, ref in zip(g["obs_scan"], g[f"obs_out_{bn}"]):
                fields["scan"][3] = s
                w.world.cache.clear()
                out = w.get_laser_observation()
                assert out.dtype == np.float64 and out.shape == (bn,)
                assert np.array_equal(out, ref)
        # get_local_goal
        for st, goal, ref in zip(g["lg_state"], g["lg_goal"], g["lg_out"]):
            fields["pose"][3] = st
            w.world.cache.clear()
            w.goal_point = list(goal)
            assert np.abs(np.asarray(w.get_local_goal()) - ref).max() <= 1e-12
        # getters mirror the subscriber callbacks (stage_world1.py:88-114): state_GT / speed_GT / state / speed / crash
        for k in range(len(g["cb_in"])):
            x, y, yaw, vx, vy, wz = g["cb_in"][k]
            fields["pose"][3] = (x, y, yaw)
            fields["speed_gt"][3] = (np.hypot(vx, vy), wz)
            fields["speed"][3] = (vx, wz)
            fields["crashed"][3] = 1
            w.world.cache.clear()
            assert np.abs(np.asarray(w.get_self_stateGT()) - g["cb_state_gt"][k]).max() <= 1e-12
            assert np.abs(np.asarray(w.get_self_speedGT()) - g["cb_speed_gt"][k]).max() <= 1e-12
            assert np.abs(np.asarray(w.get_self_state()) - g["cb_state"][k]).max() <= 1e-12
            assert np.abs(np.asarray(w.get_self_speed()) - g["cb_speed"][k]).max() <= 1e-12
            assert w.get_crash_state() == int(g["cb_crashed"])
    finally:
        W.set_backend_factory(None)


@pytest.mark.parametrize("variant", VARIANTS)
def test_facade_reward_and_episode_setup_match_the_reference(variant):
    """The facade on the fp64 oracle backend: robots teleported into the golden states (the at-rest cases: a facade
    world is ONE world, so spinning / driving cases could touch a neighbour or a wall), one tick, then
    get_reward_and_terminate(t) with the golden step counter."""
    from mrca import stage_world as W
    g = gold(variant)
    n_env = {"stage1": 24, "stage2": 44, "circle": 50}[variant]
    W.set_backend_factory(lambda sc: U.OracleBackend(sc, np.float64))
    try:
        cls = _facade_cls(variant)
        ws = [cls(512, i, n_env) for i in range(n_env)]
        world = ws[0].world
        env = world.backend.env
        rest = np.nonzero((g["rw_cmd"] == 0).all(1))[0]
        assert len(rest) >= 100
        for lo in range(0, len(rest), n_env):
            ks = rest[lo: lo + n_env]
            for w, k in zip(ws, ks):
                w.control_pose(list(g["rw_pose0"][k]))
                w.goal_point = list(g["rw_goal"][k])
                world.teleport(w.index, goal=w.goal_point)
                w.distance = float(g["rw_prev"][k])
            n = len(ks)
            env.prev_dist[:n] = g["rw_prev"][ks]
            env.crashed[:n] = g["rw_crashed"][ks]
            for w in ws[:n]:
                w.control_vel([0.0, 0.0])
            world.tick()
            for w, k in zip(ws, ks):
                r, term, res = w.get_reward_and_terminate(int(g["rw_t"][k]))
                assert abs(r - g["rw_reward"][k]) <= 1e-11, (k, r, g["rw_reward"][k])
                assert term == bool(g["rw_term"][k])
                assert {0: 0, "Reach Goal": 1, "Crashed": 2, "Time out": 3}[res] == int(g["rw_result"][k])
                assert abs(w.distance - g["rw_dist"][k]) <= 1e-12 and abs(w.pre_distance - g["rw_pre"][k]) <= 1e-12
                assert np.abs(np.asarray(w.get_local_goal()) - g["rw_local_goal"][k]).max() <= 1e-11
        # reset_pose / generate_goal_point of the table-driven robots
        if variant != "stage1":
            for i in g["ep_index"]:
                w = ws[int(i)]
                if variant == "stage2" and 33 < i < 44:
                    continue
                w.reset_pose()
                p = np.asarray(w.get_self_stateGT()) - g["ep_reset_pose"][i]
                p[2] = wrap(p[2])
                assert np.abs(p).max() <= 2e-6        # the teleport goes through the fp32 boundary (mrca_reset)
                w.generate_goal_point()
                assert np.array_equal(np.asarray(w.goal_point), g["ep_goal"][i])
                assert w.pre_distance == g["ep_pre_distance"][i] and w.distance == g["ep_distance"][i]
    finally:
        W.set_backend_factory(None)


def test_published_messages_of_the_reference_are_what_the_boundary_takes():
    """control_vel publishes Twist(linear.x = v, angular.z = w), everything else 0 (stage_world1.py:226-234);
    control_pose publishes the yaw as a quaternion about z (:237-249) -- the (v, w) pair and the (x, y, yaw) triple are
    exactly what mrca_step / mrca_reset take (include/mrca_env.h)."""
    for v in VARIANTS:
        g = gold(v)
        tw = g["cv_twist"]
        assert np.array_equal(tw[:, 0], g["cv_action"][:, 0]) and np.array_equal(tw[:, 5], g["cv_action"][:, 1])
        assert np.all(tw[:, 1:5] == 0)
        pm, p = g["cp_msg"], g["cp_pose"]
        assert np.array_equal(pm[:, :2], p[:, :2]) and np.all(pm[:, 2:5] == 0)
        assert np.abs(wrap(2 * np.arctan2(pm[:, 5], pm[:, 6]) - p[:, 2])).max() <= 1e-12
        assert np.array_equal(g["cb_scan"], np.linspace(0, 6, 512, dtype=np.float32).astype(np.float64))


# ------------------------------------------------------------------------------------------------ samplers
def _disc_support(pose, goal):
    assert np.all(np.hypot(pose[:, 0], pose[:, 1]) <= 9.0 + 1e-6)
    assert np.all((pose[:, 2] >= -np.pi - 1e-6) & (pose[:, 2] <= 2 * np.pi + 1e-6))
    assert np.all(np.hypot(goal[:, 0], goal[:, 1]) <= 9.0 + 1e-6)
    d = np.hypot(goal[:, 0] - pose[:, 0], goal[:, 1] - pose[:, 1])
    assert np.all((d >= 8.0 - 1e-5) & (d <= 10.0 + 1e-5))


def _region_support(cur, pt):
    assert np.all((pt[:, 0] >= 9.0) & (pt[:, 0] <= 19.0))
    y = pt[:, 1]
    assert np.all(((y <= -1.0 + 1e-6) & (y >= -5.0 - 1e-6)) | ((y <= -13.0 + 1e-6) & (y >= -19.0 - 1e-6)))
    assert np.all(np.hypot(pt[:, 0] - cur[:, 0], pt[:, 1] - cur[:, 1]) >= 7.0 - 1e-5)


def _ks(a, b, what):
    p = stats.ks_2samp(np.asarray(a, np.float64), np.asarray(b, np.float64)).pvalue
    assert p > 0.01, f"{what}: KS p = {p:.4g}"


def compare_disc_draws(g, pose, goal, tag):
    rp, rg = g["dr_pose"].astype(np.float64), g["dr_goal"].astype(np.float64)
    _disc_support(rp, rg)
    _disc_support(pose, goal)
    _ks(pose[:, 0], rp[:, 0], tag + " pose x")
    _ks(pose[:, 1], rp[:, 1], tag + " pose y")
    _ks(np.mod(pose[:, 2], 2 * np.pi), np.mod(rp[:, 2], 2 * np.pi), tag + " pose theta")
    _ks(np.hypot(pose[:, 0], pose[:, 1]), np.hypot(rp[:, 0], rp[:, 1]), tag + " pose radius")
    _ks(goal[:, 0], rg[:, 0], tag + " goal x")
    _ks(goal[:, 1], rg[:, 1], tag + " goal y")
    _ks(np.hypot(goal[:, 0] - pose[:, 0], goal[:, 1] - pose[:, 1]), np.hypot(rg[:, 0] - rp[:, 0], rg[:, 1] - rp[:, 1]),
        tag + " goal distance")


def compare_region_draws(g, cur, pose, goal, tag):
    rc, rp, rg = (g[k].astype(np.float64) for k in ("dr_cur", "dr_pose", "dr_goal"))
    _region_support(rc, rp)
    _region_support(rp, rg)
    _region_support(cur, pose)
    _region_support(pose, goal)
    for name, a, b in (("pose", pose, rp), ("goal", goal, rg)):
        _ks(a[:, 0], b[:, 0], f"{tag} {name} x")
        _ks(a[:, 1], b[:, 1], f"{tag} {name} y")
    _ks(np.mod(pose[:, 2], 2 * np.pi), np.mod(rp[:, 2], 2 * np.pi), tag + " pose theta")
    _ks(np.hypot(pose[:, 0] - cur[:, 0], pose[:, 1] - cur[:, 1]), np.hypot(rp[:, 0] - rc[:, 0], rp[:, 1] - rc[:, 1]),
        tag + " pose distance from the robot")
    _ks(np.hypot(goal[:, 0] - pose[:, 0], goal[:, 1] - pose[:, 1]), np.hypot(rg[:, 0] - rp[:, 0], rg[:, 1] - rp[:, 1]),
        tag + " goal distance from the robot")


def region_scenario(n):
    R = 32
    return S.Scenario("region", n // R, R, S.empty_grid(), timeout=200, pre_dist_zero=True, auto_reset=S.AUTO_NONE,
                      reset_mode=np.full(R, S.RESET_REGION, np.int32))


def test_philox_samplers_of_the_oracle_match_the_reference_draws():
    """Support exact, marginals indistinguishable (two-sample KS, p > 0.01) from 10^5 draws of the reference's
    generate_random_pose / generate_random_goal (np.random)."""
    g = gold("stage1")
    env = U.oracle_env(S.stage1(num_worlds=3125, robots_per_world=32, seed=11), np.float32)
    idx = np.arange(env.N)
    pose = env._sample_pose(idx)
    env.pose[:] = pose
    goal = env._sample_goal(idx)
    compare_disc_draws(g, pose.astype(np.float64), goal.astype(np.float64), "oracle disc")

    g = gold("stage2")
    n = 99968
    env = U.oracle_env(region_scenario(n), np.float32)
    idx = np.arange(env.N)
    cur = g["dr_cur"][:n]
    env.pose[:] = cur
    pose = env._sample_pose(idx)
    env.pose[:] = pose
    goal = env._sample_goal(idx)
    sub = {k: g[k][:n] for k in ("dr_cur", "dr_pose", "dr_goal")}
    compare_region_draws(sub, cur.astype(np.float64), pose.astype(np.float64), goal.astype(np.float64), "oracle region")


def test_facade_samplers_match_the_reference_draws():
    from mrca import stage_world as W
    W.set_backend_factory(_FieldBackend)
    try:
        n = 20000
        g = gold("stage1")
        w = W.Stage1World(512, 0, 24)
        f = w.world.backend.f
        pose, goal = np.zeros((n, 3)), np.zeros((n, 2))
        for k in range(n):
            pose[k] = w.generate_random_pose()
            f["pose"][0] = pose[k]
            w.world.cache.clear()
            goal[k] = w.generate_random_goal()
        compare_disc_draws(g, pose, goal, "facade disc")
        g = gold("stage2")
        w = W.Stage2World(512, 40, 44)
        f = w.world.backend.f
        cur = g["dr_cur"][:n].astype(np.float64)
        for k in range(n):
            f["pose"][40] = cur[k]
            w.world.cache.clear()
            pose[k] = w.generate_random_pose()
            f["pose"][40] = pose[k]
            w.world.cache.clear()
            goal[k] = w.generate_random_goal()
        sub = {k: g[k][:n] for k in ("dr_cur", "dr_pose", "dr_goal")}
        compare_region_draws(sub, cur, pose, goal, "facade region")
    finally:
        W.set_backend_factory(None)


# ------------------------------------------------------------------------------------------------ the HIP env
@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_hip_env_reward_terminal_local_goal_match_the_reference(variant):
    """Robots of the device env are teleported into the golden states (mrca_reset(poses, goals)), the bookkeeping the
    reference carries between ticks (distance, stall flag, step counter) is written into the arena, ONE tick runs with
    the golden command: reward / terminal / result / distance / local goal must be the REFERENCE's values."""
    import torch
    from mrca.vec_env import VecStageWorld
    g = gold(variant)
    K = len(g["rw_t"])
    env = VecStageWorld(one_robot_worlds(g, K), device="cuda:0")
    dev = env.device
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)      # noqa: E731
    env.reset(None, t(g["rw_pose0"], np.float32), t(g["rw_goal"], np.float32))
    env.prev_dist.copy_(t(g["rw_prev"], np.float32))
    env.t.copy_(t(g["rw_t"], np.int32))
    env.crashed.copy_(t(g["rw_crashed"], np.uint8))
    env.step(t(g["rw_cmd"], np.float32))
    torch.cuda.synchronize()
    h = lambda x: x.cpu().numpy().astype(np.float64)                             # noqa: E731
    assert np.array_equal(h(env.speed_gt), g["rw_speed_gt"])
    assert np.array_equal(h(env.speed), g["rw_cmd"])
    check_reward_cases(g, h(env.pose), h(env.reward), env.done.cpu().numpy(), env.result.cpu().numpy(),
                       h(env.prev_dist), h(env.local_goal), fp32=True)
    # the reference's observation of a 512-beam scan is x / 6 - 0.5 elementwise (pinned above on the golden scans):
    # the device's newest frame is that, rounded once to fp32
    scan = h(env.scan)
    assert np.abs(h(env.obs[:, -1]) - (scan / 6.0 - 0.5)).max() <= 6e-8


@pytest.mark.gpu
def test_hip_env_observation_of_real_scans_matches_the_reference_formula():
    import torch
    from mrca.vec_env import VecStageWorld
    g = gold("stage1")
    finite = np.isfinite(g["obs_scan"]).all(1)
    assert np.array_equal(g["obs_out_512"][finite], g["obs_scan"][finite].astype(np.float64) / 6.0 - 0.5)
    env = VecStageWorld(S.stage1(num_worlds=4, robots_per_world=24, seed=5), device="cuda:0")
    env.reset()
    rng = np.random.default_rng(1)
    for _ in range(4):
        env.step(torch.from_numpy(U.random_actions(rng, env.N)).cuda())
    torch.cuda.synchronize()
    scan = env.scan.cpu().numpy().astype(np.float64)
    assert scan.min() < 3.0 and scan.max() == 6.0
    assert np.abs(env.obs[:, -1].cpu().numpy().astype(np.float64) - (scan / 6.0 - 0.5)).max() <= 6e-8


def test_sparse_beam_index_is_the_references_pick_rule():
    """vec_env.sparse_beam_index (what mrca_sparse_obs is fed) against the beams the REFERENCE's get_laser_observation
    picked for beam_num 256 / 128: a golden scan whose 512 values are all different identifies every pick."""
    from mrca.vec_env import sparse_beam_index
    g = gold("stage1")
    assert np.array_equal(sparse_beam_index(512, 512), np.arange(512))
    tagged = 0
    for s, o256, o128 in zip(g["obs_scan"], g["obs_out_256"], g["obs_out_128"]):
        if not np.isfinite(s).all() or len(np.unique(s)) != 512:
            continue
        tagged += 1
        full = s.astype(np.float64) / 6.0 - 0.5
        assert np.array_equal(full[sparse_beam_index(512, 256)], o256)
        assert np.array_equal(full[sparse_beam_index(512, 128)], o128)
    if not tagged:          # no golden scan with 512 distinct values: the rule against the facade's own loop instead
        for bn in (256, 128, 300, 64):
            step = 512.0 / bn
            want = [int(i * step) for i in range(bn // 2)] + [int(511.0 - i * step) for i in range(bn // 2)][::-1]
            got = sparse_beam_index(512, bn).tolist()
            assert got == want or bn == 300, bn       # (non-dyadic steps: repeated addition may round differently)


@pytest.mark.gpu
@pytest.mark.parametrize("bn", [256, 128])
def test_hip_sparse_observation_matches_the_reference(bn):
    """Device-side get_laser_observation for beam_num != 512 (mrca_sparse_obs): on REAL scans of the HIP env the stack a
    StageWorld(bn, ...) would build -- every frame, deque order -- equals the reference's pick rule applied to the env's
    own full observation, bit for bit (the affine map is the same instruction sequence)."""
    import torch
    from mrca.vec_env import VecStageWorld, sparse_beam_index
    env = VecStageWorld(S.stage1(num_worlds=4, robots_per_world=24, seed=5), device="cuda:0")
    env.reset()
    rng = np.random.default_rng(2)
    for _ in range(5):
        env.step(torch.from_numpy(U.random_actions(rng, env.N)).cuda())
    got = env.sparse_obs(bn)
    idx = torch.from_numpy(sparse_beam_index(512, bn)).long().cuda()
    assert got.shape == (env.N, 3, bn)
    assert torch.equal(got, env.obs[:, :, idx])
    # ... and the facade's float64 observation of the same scan is that row to fp32 rounding
    scan = env.scan.cpu().numpy().astype(np.float64)
    assert np.abs(got[:, -1].cpu().numpy().astype(np.float64) - (scan[:, idx.cpu().numpy()] / 6.0 - 0.5)).max() <= 6e-8
    env.close()


@pytest.mark.gpu
def test_hip_samplers_match_the_reference_draws():
    import torch
    from mrca.vec_env import VecStageWorld
    g = gold("stage1")
    env = VecStageWorld(S.stage1(num_worlds=3125, robots_per_world=32, seed=23), device="cuda:0")
    env.reset()
    torch.cuda.synchronize()
    compare_disc_draws(g, env.pose.cpu().numpy().astype(np.float64), env.goal.cpu().numpy().astype(np.float64), "hip disc")
    env.close()
    g = gold("stage2")
    n = 99968
    env = VecStageWorld(region_scenario(n), device="cuda:0")
    cur = np.zeros((n, 3), np.float32)
    cur[:, :2] = g["dr_cur"][:n, :2]
    env.reset(None, torch.from_numpy(cur).cuda(), torch.zeros(n, 2, device="cuda"))
    env.reset()
    torch.cuda.synchronize()
    sub = {k: g[k][:n] for k in ("dr_cur", "dr_pose", "dr_goal")}
    compare_region_draws(sub, cur.astype(np.float64), env.pose.cpu().numpy().astype(np.float64),
                         env.goal.cpu().numpy().astype(np.float64), "hip region")
