"""The ROS-free restatement of the stageros topic contract (mrca/bridge.py) on the oracle backend:
every published field against what stage_ros-add_pose_and_crash/src/stageros.cpp puts there, the inbound topics and the
service, Stage's persistent commanded velocity with the GLOBAL watchdog -- and the messages fed into the REFERENCE's own
subscriber callbacks (stage_world1.py:88-114, under the ROS stand-ins of tools/make_golden_env.py; skipped where the
reference checkout is absent): what the reference's StageWorld then believes must be the env's state."""
import math
import os
import sys

import numpy as np
import pytest

import util as U
from util import S

REF = U.reference_dir() or "/nonexistent"


_BACKEND = "oracle"      # "hip": the product backend (the -m gpu leg below re-runs this module's tests on it)


def make(R=24, seed=4):
    from mrca import bridge
    sc = S.stage1(num_worlds=1, robots_per_world=R, seed=seed)
    sc.auto_reset = S.AUTO_NONE
    if _BACKEND == "hip":
        from mrca import stage_world
        backend = stage_world.HipBackend(sc)
    else:
        backend = U.OracleBackend(sc, np.float32)
    backend.reset(np.ones(R, np.uint8), None, None)
    init = S.load_tables()["stage1"]["world_agents"][:R]
    return bridge, bridge.StageBridge(backend, R, init), backend


def twist(bridge, v, w):
    return bridge.Twist(bridge.Vector3(v, 0.0, 0.0), bridge.Vector3(0.0, 0.0, w))


def test_published_fields_follow_stageros():
    bridge, br, backend = make()
    first = br.publish()
    for r in range(24):
        br.cmd_vel(r, twist(bridge, 0.5 + 0.01 * r, 0.4 - 0.03 * r))
    before = backend.field("pose").astype(np.float64).copy()
    out = br.update_world()
    pose = backend.field("pose").astype(np.float64)
    assert set(out) == {f"/robot_{r}/{n}" for r in range(24)
                        for n in ("base_scan", "odom", "is_crashed", "base_pose_ground_truth")} | {"/clock"}
    for r in (0, 7, 23):
        ls = out[f"/robot_{r}/base_scan"]                              # stageros.cpp:493-513
        assert ls.angle_min == -math.pi / 2 and ls.angle_max == math.pi / 2
        assert ls.angle_increment == math.pi / 511                     # fov / (sample_count - 1)
        assert (ls.range_min, ls.range_max) == (0.0, 6.0)              # worlds/stage1.world:13
        assert ls.ranges.dtype == np.float32 and ls.ranges.shape == (512,)      # float32[] on the wire (:505)
        assert np.array_equal(ls.ranges, backend.field("scan")[r]) and float(ls.ranges.max()) <= 6.0
        # intensities: 1 where the beam returned from the floorplan, 0 for another robot (ranger_return 0.5 cast to uint8,
        # stageros.cpp:506) and for a miss
        assert ls.intensities.shape == (512,) and ls.intensities.dtype == np.float32
        rob = np.asarray(backend.field("hit_robot"))[r].astype(bool)
        assert np.array_equal(ls.intensities, ((ls.ranges < 6.0) & ~rob).astype(np.float32))
        assert set(np.unique(ls.intensities)) <= {0.0, 1.0}
        assert np.asarray(backend.field("hit_robot")).any() and any(
            (out[f"/robot_{q}/base_scan"].intensities == 1.0).any() for q in range(24))   # both kinds of return occur
        assert ls.header.frame_id == f"/robot_{r}/base_laser_link" and ls.header.stamp == pytest.approx(0.1)
        od = out[f"/robot_{r}/odom"]                                   # :543-558
        assert (od.pose.position.x, od.pose.position.y, od.pose.position.z) == (pose[r, 0], pose[r, 1], 0.0)
        assert od.pose.orientation.x == 0 and od.pose.orientation.y == 0
        assert od.pose.orientation.z == pytest.approx(math.sin(pose[r, 2] / 2)) and \
            od.pose.orientation.w == pytest.approx(math.cos(pose[r, 2] / 2))
        assert od.twist.linear.x == pytest.approx(0.5 + 0.01 * r, abs=1e-6) and od.twist.linear.y == 0.0   # GetVelocity()
        assert od.twist.angular.z == pytest.approx(0.4 - 0.03 * r, abs=1e-6)
        assert od.header.frame_id == f"/robot_{r}/odom"
        gt = out[f"/robot_{r}/base_pose_ground_truth"]                 # :575-611: finite differences, world frame
        assert gt.twist.linear.x == pytest.approx((pose[r, 0] - before[r, 0]) / 0.1, abs=1e-12)
        assert gt.twist.linear.y == pytest.approx((pose[r, 1] - before[r, 1]) / 0.1, abs=1e-12)
        assert gt.twist.angular.z == pytest.approx(bridge.normalize(pose[r, 2] - before[r, 2]) / 0.1, abs=1e-12)
        assert out[f"/robot_{r}/is_crashed"].data == int(backend.field("crashed")[r])     # Stalled(), :560-564
        g0 = first[f"/robot_{r}/base_pose_ground_truth"]               # "no previous readings": velocity 0 (:598-599)
        assert (g0.twist.linear.x, g0.twist.linear.y, g0.twist.angular.z) == (0.0, 0.0, 0.0)
    assert (out["/clock"].secs, out["/clock"].nsecs) == (0, 100000000)          # one tick of 100 ms (:778-783)
    # a single-robot world drops the robot_<i> prefix (mapName, :196-215)
    _b, br1, _ = make(R=1)
    assert set(br1.publish()) == {"/base_scan", "/odom", "/is_crashed", "/base_pose_ground_truth", "/clock"}


def test_commanded_velocity_persists_and_the_watchdog_is_global():
    bridge, br, backend = make(R=4, seed=9)
    br.cmd_pose(0, bridge.Pose(bridge.Vector3(0.0, 0.0, 0.0), bridge.quaternion_from_yaw(0.0)))
    br.cmd_pose(1, bridge.Pose(bridge.Vector3(0.0, 3.0, 0.0), bridge.quaternion_from_yaw(0.0)))
    br.cmd_pose(2, bridge.Pose(bridge.Vector3(0.0, -3.0, 0.0), bridge.quaternion_from_yaw(0.0)))
    br.cmd_pose(3, bridge.Pose(bridge.Vector3(4.0, 6.0, 0.0), bridge.quaternion_from_yaw(0.0)))
    br.cmd_vel(0, twist(bridge, 1.0, 0.0))                  # ONE command at t = 0
    xs = []
    for _ in range(4):
        br.update_world()
        xs.append(float(backend.field("pose")[0, 0]))
    # Stage keeps the velocity: ticks 1 and 2 move; at t = 0.2 s nobody has spoken for base_watchdog_timeout -> all stop
    assert xs[0] == pytest.approx(0.1, abs=1e-6) and xs[1] == pytest.approx(0.2, abs=1e-6)
    assert xs[2] == xs[1] and xs[3] == xs[1]
    # the watchdog is global (stageros.cpp:466-471): robot 1 talking keeps robot 0 moving
    br.cmd_vel(0, twist(bridge, 1.0, 0.0))
    for _ in range(5):
        br.cmd_vel(1, twist(bridge, 0.0, 0.1))
        br.update_world()
    assert float(backend.field("pose")[0, 0]) == pytest.approx(xs[1] + 0.5, abs=1e-5)


def test_cmd_pose_and_reset_positions():
    bridge, br, backend = make()
    q = bridge.quaternion_from_yaw(2.5)
    br.cmd_pose(3, bridge.Pose(bridge.Vector3(1.25, -4.5, 7.0), q))          # z is dropped (pose.z = 0, :292)
    assert np.allclose(backend.field("pose")[3], [1.25, -4.5, 2.5], atol=1e-6)
    for r in range(24):
        br.cmd_vel(r, twist(bridge, 1.0, 0.0))
    for _ in range(3):
        br.update_world()
    br.reset_positions()                                                      # cb_reset_srv, :260-269
    want = np.asarray(S.load_tables()["stage1"]["world_agents"], np.float64)
    got = backend.field("pose").astype(np.float64)
    assert np.abs(got[:, :2] - want[:, :2]).max() < 1e-6
    assert np.abs(np.arctan2(np.sin(got[:, 2] - want[:, 2]), np.cos(got[:, 2] - want[:, 2]))).max() < 1e-6
    assert int(backend.field("crashed").sum()) == 0                           # SetStall(false)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "stage_world1.py")), reason="reference checkout absent")
def test_the_references_own_callbacks_digest_the_bridge_messages():
    sys.path.insert(0, os.path.join(U.ROOT, "tools"))
    import make_golden_env as G
    G.install_stubs()
    sys.path.insert(0, REF)
    sys.modules.pop("stage_world1", None)     # (an earlier spmd run of this session may have left the drop-in of that name)
    import stage_world1
    assert os.path.dirname(os.path.abspath(stage_world1.__file__)) == os.path.abspath(REF)
    bridge, br, backend = make()
    rng = np.random.default_rng(0)
    worlds = [G.make(stage_world1.StageWorld, r) for r in range(24)]
    br.publish()
    for k in range(6):
        for r in range(24):
            br.cmd_vel(r, twist(bridge, float(rng.uniform(0, 1)), float(rng.uniform(-1, 1))))
        out = br.update_world()
        for r, w in enumerate(worlds):
            w.ground_truth_callback(_as_ros_odometry(out[f"/robot_{r}/base_pose_ground_truth"]))
            w.odometry_callback(_as_ros_odometry(out[f"/robot_{r}/odom"]))
            w.laser_scan_callback(out[f"/robot_{r}/base_scan"])
            w.crash_callback(out[f"/robot_{r}/is_crashed"])
        pose, sgt, sp = (backend.field(n).astype(np.float64) for n in ("pose", "speed_gt", "speed"))
        for r, w in enumerate(worlds):
            assert np.abs(np.asarray(w.get_self_stateGT()) - pose[r]).max() < 1e-6
            assert np.abs(np.asarray(w.get_self_state()) - pose[r]).max() < 1e-6
            # v = sqrt(vx^2 + vy^2) of the finite differences of fp32 positions over 0.1 s: 1e-5-ish of the env's (|v|, w)
            assert np.abs(np.asarray(w.get_self_speedGT()) - sgt[r]).max() < 5e-5, (k, r, w.get_self_speedGT(), sgt[r])
            assert np.abs(np.asarray(w.get_self_speed()) - sp[r]).max() < 1e-6
            assert w.get_crash_state() == int(backend.field("crashed")[r])
            assert np.array_equal(w.scan, backend.field("scan")[r].astype(np.float64))
            # ... and the reference's observation of that scan is the env's newest frame
            assert np.abs(w.get_laser_observation() - backend.field("obs")[r, -1]).max() < 1e-6


def _as_ros_odometry(m):
    """bridge.Odometry (pose / twist one level deep) -> the nesting of nav_msgs/Odometry the reference's callbacks
    read (msg.pose.pose.position.x, msg.twist.twist.linear.x, ...)."""
    class N:
        pass
    o = N()
    o.pose, o.twist = N(), N()
    o.pose.pose, o.twist.twist = m.pose, m.twist
    return o


# ------------------------------------------------------------------------------------------------ the product backend
_ON_ANY_BACKEND = [n for n, f in sorted(globals().items()) if n.startswith("test_") and callable(f)]


@pytest.mark.gpu
@pytest.mark.parametrize("name", _ON_ANY_BACKEND)
def test_on_the_hip_backend(name, monkeypatch):
    """SURVEY 8(f4) on the product: every test of this module once more with the bridge fed from HipBackend (the HIP env
    through the C ABI) instead of the oracle -- the published fields, the inbound topics and the service, Stage's persistent
    velocity with the global watchdog, and the reference's own subscriber callbacks digesting the messages."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.build()
    fn = globals()[name]
    for mark in getattr(fn, "pytestmark", []):
        if mark.name == "skipif" and mark.args and mark.args[0]:
            # on the GPU box "reference checkout absent" means the staged archive (tests/_reference.tgz) did not travel: the leg
            # that feeds the reference's own callbacks must not disappear silently
            pytest.fail(f"{name} on the HIP backend cannot run: {mark.kwargs.get('reason', 'skipif')} (and no "
                        "tests/_reference.tgz: tools/stage_reference.sh)")
    monkeypatch.setitem(globals(), "_BACKEND", "hip")
    fn()
