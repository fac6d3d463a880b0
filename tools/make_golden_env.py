#!/usr/bin/env python3
"""Golden vectors of the PYTHON half of the environment, made by RUNNING the reference's own
``stage_world1.py`` / ``stage_world2.py`` / ``circle_world.py`` in this container.

The classes import ROS (rospy, tf, six message packages) at module level and use Python 2's ``xrange``; none
of that is needed by the methods on the hot path, so this script registers empty stand-ins in ``sys.modules``
(publishers RECORD what is published, so the outgoing messages are pinned too), injects ``builtins.xrange``,
imports the three modules from ``$MRCA_REFERENCE`` (default /root/reference), builds instances with
``object.__new__`` (the constructors busy-wait for ROS callbacks) and records what the reference's methods return:

  get_laser_observation   stage_world1.py:122-140   NaN / inf beams, beam_num 512 / 256 / 128
  get_local_goal          stage_world1.py:155-160
  get_reward_and_terminate stage_world1.py:180-211, stage_world2.py:175-208, circle_world.py:171-203
                          goal reach, crash, both on one tick, |w| either side of the threshold, t either side of the
                          time-out, the pre_distance = 0 first step of stage2 / circle, random states
  generate_goal_point     stage_world1.py:171-177, stage_world2.py:164-171, circle_world.py:164-167
  reset_pose / control_pose / control_vel  -> the published Pose / Twist messages (stage_world1.py:213-249, ...)
  ground_truth_callback / odometry_callback / laser_scan_callback / crash_callback   stage_world1.py:88-114
  generate_random_pose / generate_random_goal   10^5 draws each   stage_world1.py:251-274, stage_world2.py:250-287

Output: tests/golden/env_python_{stage1,stage2,circle}.npz (committed; the GPU box has no /root/reference).
Every state input is a float32-representable number stored as float64, so the fp32 device env can be put into exactly
the state the reference was evaluated in.  The only arithmetic in here that is not the reference's is (a) the stand-in
``tf.transformations`` (yaw <-> quaternion about z, the textbook formulas) and (b) the explicit-Euler pose of the
"moved" reward cases (float64, heading at tick start) -- both labelled in the files.
"""
import builtins
import os
import sys
import tempfile
import types

import numpy as np

REF = os.environ.get("MRCA_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
RESULT_CODE = {0: 0, "Reach Goal": 1, "Crashed": 2, "Time out": 3}
N_DRAWS = 100_000


# ----------------------------------------------------------------------------------------------- ROS stand-ins
class _Msg:
    """Attribute bag: msg.linear.x = ... creates the path on the fly (Twist, Pose, Odometry ...)."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        v = _Msg()
        object.__setattr__(self, k, v)
        return v


class _Publisher:
    def __init__(self, *a, **k):
        self.sent = []

    def publish(self, m):
        self.sent.append(m)


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("rospy", init_node=lambda *a, **k: None, Publisher=_Publisher, Subscriber=lambda *a, **k: None,
        ServiceProxy=lambda *a, **k: (lambda: None), sleep=lambda *a, **k: None, is_shutdown=lambda: False)

    def quaternion_from_euler(ai, aj, ak, axes="sxyz"):      # only ever called as (0, 0, yaw, 'rxyz')
        assert ai == 0 and aj == 0
        return np.array([0.0, 0.0, np.sin(ak / 2.0), np.cos(ak / 2.0)])

    def euler_from_quaternion(q, axes="sxyz"):
        x, y, z, w = q
        return (np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y)),
                np.arcsin(np.clip(2 * (w * y - z * x), -1, 1)),
                np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z)))

    tr = mod("tf.transformations", quaternion_from_euler=quaternion_from_euler,
             euler_from_quaternion=euler_from_quaternion)
    mod("tf", transformations=tr)
    for pkg, names in (("geometry_msgs", ("Twist", "Pose")), ("nav_msgs", ("Odometry",)),
                       ("sensor_msgs", ("LaserScan",)), ("rosgraph_msgs", ("Clock",)), ("std_msgs", ("Int8",))):
        mod(pkg)
        mod(pkg + ".msg", **{n: _Msg for n in names})
    mod("std_srvs")
    mod("std_srvs.srv", Empty=_Msg)
    builtins.xrange = range


def f32(x):
    """float32-representable values, carried as float64."""
    return np.asarray(x, np.float32).astype(np.float64)


def make(cls, index, beam_num=512):
    w = object.__new__(cls)
    w.index, w.num_env, w.beam_mum, w.goal_size = index, 64, beam_num, 0.5
    w.cmd_vel, w.cmd_pose = _Publisher(), _Publisher()
    w.laser_cb_num = 0
    return w


# ----------------------------------------------------------------------------------------------- recorders
def rec_observation(cls, rng):
    S = 24
    scans = rng.uniform(0.0, 6.0, (S, 512)).astype(np.float32)
    scans[0] = 6.0
    scans[1] = 0.0
    scans[2] = np.linspace(0, 6, 512, dtype=np.float32)
    scans[3, ::7] = np.nan
    scans[4, ::5] = np.inf
    scans[5, 3::11] = -np.inf
    scans[6, [0, 255, 256, 511]] = [np.nan, np.inf, np.nan, np.inf]
    out = {"obs_scan": scans}
    for bn in (512, 256, 128):
        w = make(cls, 0, bn)
        rows = []
        for s in scans:
            # laser_scan_callback stores np.array(scan.ranges): rospy hands float32[] over as a tuple of Python floats,
            # so the array is float64 (stage_world1.py:100)
            w.scan = np.array(tuple(float(v) for v in s))
            rows.append(np.asarray(w.get_laser_observation(), np.float64))
        out[f"obs_out_{bn}"] = np.stack(rows)
    return out


def rec_local_goal(cls, rng, extent):
    M = 400
    st = np.stack([rng.uniform(-extent, extent, M), rng.uniform(-extent, extent, M), rng.uniform(-np.pi, np.pi, M)], 1)
    st[0] = (0, 0, 0)
    st[1] = (1, 2, np.pi)
    st[2] = (-3, 4, -np.pi / 2)
    st[3] = (5, -5, np.pi / 2)
    goal = rng.uniform(-extent, extent, (M, 2))
    goal[4] = st[4, :2]                      # standing on the goal
    st, goal = f32(st), f32(goal)
    w = make(cls, 0)
    out = np.zeros((M, 2))
    for k in range(M):
        w.state_GT, w.goal_point = list(st[k]), list(goal[k])
        out[k] = w.get_local_goal()
    return {"lg_state": st, "lg_goal": goal, "lg_out": out}


def rec_reward(cls, rng, extent, w_thresh, timeout, pre_zero):
    """Cases are built as (pose before the tick, command) -> what the ground-truth topics would then show.  A robot
    that is stalled does not move, so a crashed case carries speed_GT = (0, 0) (stageros.cpp:585-590)."""
    cases = []          # pose0, cmd, crashed_in, goal, prev, t

    def add(pose0, cmd, crashed, goal, prev, t):
        cases.append((pose0, cmd, crashed, goal, prev, t))

    eps = 1e-3
    # -- hand-picked
    for d in (0.3, 0.49, 0.4999, 0.5, 0.5001, 0.51, 1.0, 9.5):              # goal radius
        add((1.0, -2.0, 0.3), (0, 0), 0, (1.0 + d, -2.0), d + 0.07, 5)
    add((0.5, 0.5, 1.0), (0, 0), 1, (6.0, 0.5), 5.6, 9)                      # crash
    add((0.5, 0.5, 1.0), (0, 0), 1, (0.7, 0.5), 0.3, 9)                      # crash + goal on one tick: +15 - 15
    add((0.5, 0.5, 1.0), (0, 0), 1, (0.7, 0.5), 0.3, timeout + 1)            # ... + time-out: result precedence
    for w in (w_thresh - eps, w_thresh, w_thresh + eps, -(w_thresh - eps), -w_thresh, -(w_thresh + eps), 1.5, -1.5,
              0.7 - eps, 0.7 + eps, 1.05 - eps, 1.05 + eps):                 # the omega penalty
        add((2.0, 2.0, -0.4), (0, w), 0, (-4.0, 3.0), 6.2, 20)
    for t in (1, timeout - 1, timeout, timeout + 1, timeout + 2, 150, 151, 200, 201, 10000, 10001):
        add((-3.0, 1.0, 2.0), (0, 0), 0, (3.0, 1.0), 6.05, t)                # the time-out
    if pre_zero:                                                             # generate_goal_point leaves distance = 0
        for d in (0.3, 3.0, 12.0, 50.0):
            add((0.0, 0.0, 0.0), (0, 0), 0, (d, 0.0), 0.0, 1)
    # -- random, at rest / spinning / driving
    for k in range(300):
        p = (rng.uniform(-extent, extent), rng.uniform(-extent, extent), rng.uniform(-np.pi, np.pi))
        ang, d = rng.uniform(0, 2 * np.pi), rng.choice([rng.uniform(0.05, 0.9), rng.uniform(0.9, 12.0)])
        g = (p[0] + d * np.cos(ang), p[1] + d * np.sin(ang))
        kind = k % 3
        cmd = (0.0, 0.0) if kind == 0 else (0.0, rng.uniform(-1.5, 1.5)) if kind == 1 else \
            (rng.uniform(0, 1), rng.uniform(-1, 1))
        crashed = int(kind == 0 and rng.uniform() < 0.3)
        prev = 0.0 if (pre_zero and rng.uniform() < 0.1) else d + rng.uniform(-0.12, 0.12)
        t = int(rng.choice([rng.integers(1, timeout), timeout + rng.integers(-1, 3)]))
        add(p, cmd, crashed, g, max(prev, 0.0), t)

    K = len(cases)
    pose0 = f32([c[0] for c in cases])
    cmd = f32([c[1] for c in cases])
    crashed = np.array([c[2] for c in cases], np.int8)
    goal = f32([c[3] for c in cases])
    prev = f32([c[4] for c in cases])
    t = np.array([c[5] for c in cases], np.int64)
    # what the GT topics show after the tick (explicit Euler, heading at tick start, dt 0.1; float64 -- labelled above)
    moved = (crashed == 0) & ((cmd[:, 0] != 0) | (cmd[:, 1] != 0))
    dt = 0.1
    state = pose0.copy()
    state[:, 0] += np.where(moved, cmd[:, 0] * dt * np.cos(pose0[:, 2]), 0.0)
    state[:, 1] += np.where(moved, cmd[:, 0] * dt * np.sin(pose0[:, 2]), 0.0)
    th = pose0[:, 2] + np.where(moved, cmd[:, 1] * dt, 0.0)
    state[:, 2] = np.arctan2(np.sin(th), np.cos(th))
    speed_gt = np.where(moved[:, None], np.stack([np.abs(cmd[:, 0]), cmd[:, 1]], 1), 0.0)

    w = make(cls, 0)
    w.scan = np.full(512, 6.0)
    reward, term, result, dist, pre = (np.zeros(K), np.zeros(K, bool), np.zeros(K, np.int8), np.zeros(K), np.zeros(K))
    lgoal = np.zeros((K, 2))
    for k in range(K):
        w.state_GT, w.speed_GT = list(state[k]), list(speed_gt[k])
        w.is_crashed, w.goal_point = int(crashed[k]), list(goal[k])
        w.distance = float(prev[k])
        r, te, res = w.get_reward_and_terminate(int(t[k]))
        reward[k], term[k], result[k], dist[k], pre[k] = r, te, RESULT_CODE[res], w.distance, w.pre_distance
        lgoal[k] = w.get_local_goal()
    return {"rw_pose0": pose0, "rw_cmd": cmd, "rw_crashed": crashed, "rw_goal": goal, "rw_prev": prev, "rw_t": t,
            "rw_moved": moved, "rw_state": state, "rw_speed_gt": speed_gt,
            "rw_reward": reward, "rw_term": term, "rw_result": result, "rw_dist": dist, "rw_pre": pre,
            "rw_local_goal": lgoal}


def rec_messages(cls, rng, n_index):
    """control_vel / control_pose / reset_pose -> published messages; the four subscriber callbacks."""
    w = make(cls, 0)
    acts = f32(np.stack([rng.uniform(0, 1, 16), rng.uniform(-1, 1, 16)], 1))
    tw = np.zeros((16, 6))
    for k, a in enumerate(acts):
        w.control_vel(a)
        m = w.cmd_vel.sent[-1]
        tw[k] = (m.linear.x, m.linear.y, m.linear.z, m.angular.x, m.angular.y, m.angular.z)
    poses = f32(np.stack([rng.uniform(-9, 9, 16), rng.uniform(-9, 9, 16), rng.uniform(-2 * np.pi, 2 * np.pi, 16)], 1))
    pm = np.zeros((16, 7))
    for k, p in enumerate(poses):
        w.control_pose(list(p))
        m = w.cmd_pose.sent[-1]
        pm[k] = (m.position.x, m.position.y, m.position.z, m.orientation.x, m.orientation.y, m.orientation.z,
                 m.orientation.w)
    # callbacks
    G = 32
    gt_in = f32(np.stack([rng.uniform(-9, 9, G), rng.uniform(-9, 9, G), rng.uniform(-np.pi, np.pi, G),
                          rng.uniform(-1, 1, G), rng.uniform(-1, 1, G), rng.uniform(-1.5, 1.5, G)], 1))
    gt_state, gt_speed, od_state, od_speed = np.zeros((G, 3)), np.zeros((G, 2)), np.zeros((G, 3)), np.zeros((G, 2))
    for k, (x, y, yaw, vx, vy, wz) in enumerate(gt_in):
        m = _Msg()
        m.pose.pose.position.x, m.pose.pose.position.y = x, y
        o = m.pose.pose.orientation
        o.x, o.y, o.z, o.w = 0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)
        m.twist.twist.linear.x, m.twist.twist.linear.y, m.twist.twist.angular.z = vx, vy, wz
        w.ground_truth_callback(m)
        w.odometry_callback(m)
        gt_state[k], gt_speed[k], od_state[k], od_speed[k] = w.state_GT, w.speed_GT, w.state, w.speed
    m = _Msg()
    m.ranges = tuple(float(v) for v in np.linspace(0, 6, 512, dtype=np.float32))
    w.laser_scan_callback(m)
    flag = _Msg()
    flag.data = 1
    w.crash_callback(flag)
    out = {"cv_action": acts, "cv_twist": tw, "cp_pose": poses, "cp_msg": pm, "cb_in": gt_in, "cb_state_gt": gt_state,
           "cb_speed_gt": gt_speed, "cb_state": od_state, "cb_speed": od_speed,
           "cb_scan": np.asarray(w.scan, np.float64), "cb_crashed": np.int64(w.get_crash_state())}
    # reset_pose / generate_goal_point per index (table-driven variants publish the table pose)
    idx = np.arange(n_index)
    rp = np.full((n_index, 3), np.nan)          # x, y, yaw of the FIRST published cmd_pose
    gp = np.full((n_index, 2), np.nan)
    gp_pre, gp_dist = np.zeros(n_index), np.zeros(n_index)
    cur = f32(np.stack([rng.uniform(9, 19, n_index), rng.uniform(-19, -1, n_index), rng.uniform(-3, 3, n_index)], 1)) \
        if cls.__module__ != "stage_world1" else \
        f32(np.stack([rng.uniform(-6, 6, n_index), rng.uniform(-6, 6, n_index), rng.uniform(-3, 3, n_index)], 1))
    for i in idx:
        w = make(cls, int(i))
        w.state_GT = list(cur[i])
        orig = w.control_pose

        def control_pose(pose, w=w, orig=orig):
            orig(pose)
            w.state_GT = [pose[0], pose[1], float(np.arctan2(np.sin(pose[2]), np.cos(pose[2])))]
        w.control_pose = control_pose      # state_GT follows the teleport, so reset_pose's wait loop ends
        np.random.seed(1000 + int(i))
        w.reset_pose()
        m = w.cmd_pose.sent[0]
        rp[i] = (m.position.x, m.position.y, 2 * np.arctan2(m.orientation.z, m.orientation.w))
        np.random.seed(2000 + int(i))
        w.generate_goal_point()
        gp[i], gp_pre[i], gp_dist[i] = w.goal_point, w.pre_distance, w.distance
    out.update({"ep_index": idx, "ep_cur": cur, "ep_reset_pose": rp, "ep_goal": gp, "ep_pre_distance": gp_pre,
                "ep_distance": gp_dist})
    return out


def rec_draws(cls, rng, region):
    """10^5 draws of each sampler from the reference's process-global np.random (seeded here)."""
    w = make(cls, 40)
    if region:
        # the robot's position when the sampler runs: the world-file starts of the random robots (indices 34..43) and,
        # from the second episode on, a previous draw of the same sampler
        from model.utils import get_init_pose
        starts = np.array([get_init_pose(i) for i in range(34, 44)])
        cur = np.zeros((N_DRAWS, 3))
        cur[: N_DRAWS // 2] = starts[rng.integers(0, 10, N_DRAWS // 2)]
    poses = np.zeros((N_DRAWS, 3))
    goals = np.zeros((N_DRAWS, 2))
    np.random.seed(12345)
    for k in range(N_DRAWS):
        if region:
            if k >= N_DRAWS // 2:
                cur[k] = poses[k - N_DRAWS // 2]
            w.state_GT = list(f32(cur[k]))
        poses[k] = w.generate_random_pose()
    poses32 = f32(poses)
    np.random.seed(54321)
    for k in range(N_DRAWS):
        w.state_GT = list(poses32[k])       # the goal is drawn right after the teleport to the new pose
        goals[k] = w.generate_random_goal()
    out = {"dr_pose": poses.astype(np.float32), "dr_goal": goals.astype(np.float32)}
    if region:
        out["dr_cur"] = cur.astype(np.float32)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    sys.path.insert(0, REF)
    os.chdir(tempfile.mkdtemp())
    import stage_world1
    import stage_world2
    import circle_world
    variants = (("stage1", stage_world1.StageWorld, 9.0, 1.05, 150, False, 24, False),
                ("stage2", stage_world2.StageWorld, 19.0, 1.05, 200, True, 44, True),
                ("circle", circle_world.StageWorld, 25.0, 0.7, 10000, True, 50, True))
    for name, cls, extent, w_thresh, timeout, pre_zero, n_index, region in variants:
        rng = np.random.default_rng({"stage1": 101, "stage2": 202, "circle": 303}[name])
        data = {"variant": name, "w_thresh": w_thresh, "timeout": timeout, "pre_zero": pre_zero}
        data.update(rec_observation(cls, rng))
        data.update(rec_local_goal(cls, rng, extent))
        data.update(rec_reward(cls, rng, extent, w_thresh, timeout, pre_zero))
        data.update(rec_messages(cls, rng, n_index))
        if name != "circle":                 # circle_world.py:236-273 is byte-identical to stage_world2.py:250-287
            data.update(rec_draws(cls, rng, region))
        path = os.path.join(OUT, f"env_python_{name}.npz")
        np.savez_compressed(path, **data)
        print(name, "->", path, f"{os.path.getsize(path) / 1e6:.2f} MB;", len(data["rw_t"]), "reward cases")


if __name__ == "__main__":
    main()
