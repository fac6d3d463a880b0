"""The stageros topic contract, ROS-free: what ``stage_ros-add_pose_and_crash/src/stageros.cpp``
publishes per robot every tick and what it accepts, restated field for field as plain Python structures on top of the
batched device env -- so that an ``rclpy`` / ``rospy`` adapter is a loop of ``publisher.publish(convert(msg))`` for whoever
has ROS, and so that the reference's own subscriber callbacks (``stage_world1.py:88-114``) can be fed from the device
env (tests/test_bridge.py does exactly that, under ROS stand-ins).

Per robot i (``stageros.cpp:396-434``; names through ``mapName``, ``:196-215``: ``/robot_<i>/<name>`` when the world
holds more than one position model, ``/<name>`` otherwise):

  out  base_scan                sensor_msgs/LaserScan     stageros.cpp:479-516
  out  odom                     nav_msgs/Odometry         :543-558   (est_pose + the commanded velocity, GetVelocity())
  out  is_crashed               std_msgs/Int8             :560-564   (Stalled())
  out  base_pose_ground_truth   nav_msgs/Odometry         :575-611   (global pose + FINITE-DIFFERENCE velocity in the
                                                                      world frame, against the pose of the previous tick)
  out  /clock                   rosgraph_msgs/Clock       :778-783
  in   cmd_vel                  geometry_msgs/Twist       :272-280   SetSpeed(linear.x, linear.y, angular.z); stamps the watchdog
  in   cmd_pose                 geometry_msgs/Pose        :282-296   SetPose(x, y, 0, yaw of the quaternion)
  srv  reset_positions          std_srvs/Empty            :260-269   initial world-file poses, stall cleared

Semantics kept from stageros that the per-index facade (mrca/stage_world.py) simplifies: a commanded velocity PERSISTS
until the next cmd_vel (Stage keeps it), and the watchdog is GLOBAL: when no robot at all has sent a cmd_vel for
``base_watchdog_timeout`` = 0.2 s of simulated time every robot is stopped (``:318-320,466-471``).

``intensities`` are Stage's per-beam return value cast to uint8 (``:501-506``): 1 for the floorplan (``ranger_return``
default 1), 0 for another robot (``ranger_return 0.5``, worlds/stage1.world:95) and for a miss -- from the backend's
``hit_robot`` field (the device's MRCA_F_HIT_BITS plane).  Not restated: tf broadcasts, camera topics (no camera in any
world of the reference).
"""
import math
from dataclasses import dataclass, field

import numpy as np

FOV = math.pi            # worlds/stage1.world:12 fov 180
RANGE_MIN, RANGE_MAX = 0.0, 6.0      # worlds/stage1.world:13 range [0.0 6.0]
DT = 0.1                 # Stage's default interval_sim (the world files set none)
WATCHDOG = 0.2           # stageros.cpp:318-320


@dataclass
class Header:
    stamp: float = 0.0           # simulated time [s] (sim_time, stageros.cpp:456)
    frame_id: str = ""


@dataclass
class LaserScan:                 # stageros.cpp:493-513
    header: Header
    angle_min: float
    angle_max: float
    angle_increment: float
    range_min: float
    range_max: float
    ranges: np.ndarray           # float32[samples] (msg.ranges is float32[] on the wire, :505)
    intensities: np.ndarray      # float32[samples]
    time_increment: float = 0.0  # left at the message default by stageros
    scan_time: float = 0.0


@dataclass
class Quaternion:
    x: float = 0.0
    y: float = 0.0
    z: float = 0.0
    w: float = 1.0


@dataclass
class Vector3:
    x: float = 0.0
    y: float = 0.0
    z: float = 0.0


@dataclass
class Pose:                      # geometry_msgs/Pose
    position: Vector3 = field(default_factory=Vector3)
    orientation: Quaternion = field(default_factory=Quaternion)


@dataclass
class Twist:                     # geometry_msgs/Twist
    linear: Vector3 = field(default_factory=Vector3)
    angular: Vector3 = field(default_factory=Vector3)


@dataclass
class Odometry:                  # nav_msgs/Odometry (pose.pose / twist.twist flattened one level)
    header: Header
    pose: Pose
    twist: Twist


@dataclass
class Int8:
    data: int = 0


@dataclass
class Clock:
    secs: int = 0
    nsecs: int = 0


def quaternion_from_yaw(yaw):
    """tf::createQuaternionMsgFromYaw / setRPY(0, 0, yaw): a rotation about z."""
    return Quaternion(0.0, 0.0, math.sin(0.5 * yaw), math.cos(0.5 * yaw))


def yaw_from_quaternion(q):
    """tf::Matrix3x3(q).getRPY yaw (stageros.cpp:289-291)."""
    return math.atan2(2.0 * (q.w * q.z + q.x * q.y), 1.0 - 2.0 * (q.y * q.y + q.z * q.z))


def normalize(a):
    """Stg::normalize: an angle into (-pi, pi]."""
    return math.atan2(math.sin(a), math.cos(a))


class StageBridge:
    """``backend`` is anything with the facade's backend protocol -- ``reset(mask, poses, goals)``, ``step(actions)``,
    ``field(name)`` (mrca.stage_world.HipBackend on the GPU; tests inject the oracle) -- over ONE world of
    ``num_robots`` robots.  ``initial_poses`` [R,3] are the world file's agent poses (reset_positions)."""

    def __init__(self, backend, num_robots, initial_poses, samples=512):
        self.backend, self.R, self.samples = backend, int(num_robots), int(samples)
        self.initial_poses = np.asarray(initial_poses, np.float64).reshape(self.R, 3)
        self.cmd = np.zeros((self.R, 2), np.float32)       # the velocity Stage holds for each model
        self.ticks = 0
        self.base_last_cmd = 0.0                           # sim time of the last cmd_vel from ANY robot
        self.base_last_globalpos = None                    # [R,3] pose of the previous publish (stageros.cpp:584-600)
        self.multi = self.R > 1

    # ------------------------------------------------------------------ names
    def map_name(self, name, robot):
        return f"/robot_{robot}/{name}" if self.multi else f"/{name}"

    @property
    def sim_time(self):
        return self.ticks * DT

    # ------------------------------------------------------------------ inbound
    def cmd_vel(self, robot, twist):
        """StageNode::cmdvelReceived: SetSpeed(linear.x, linear.y, angular.z) -- a differential drive ignores y."""
        self.cmd[robot] = (twist.linear.x, twist.angular.z)
        self.base_last_cmd = self.sim_time

    def cmd_pose(self, robot, pose):
        """StageNode::poseReceived: teleport to (x, y, yaw of the quaternion); z, roll and pitch are dropped."""
        poses = np.asarray(self.backend.field("pose"), np.float32).copy()
        goals = np.asarray(self.backend.field("goal"), np.float32).copy()
        poses[robot] = (pose.position.x, pose.position.y, yaw_from_quaternion(pose.orientation))
        mask = np.zeros(self.R, np.uint8)
        mask[robot] = 1
        self.backend.reset(mask, poses, goals)

    def reset_positions(self):
        """StageNode::cb_reset_srv: every model back to its world-file pose, stall flags cleared."""
        goals = np.asarray(self.backend.field("goal"), np.float32).copy()
        poses = self.initial_poses.astype(np.float32).copy()
        poses[:, 2] = np.arctan2(np.sin(poses[:, 2]), np.cos(poses[:, 2]))
        self.backend.reset(np.ones(self.R, np.uint8), poses, goals)

    # ------------------------------------------------------------------ the tick
    def update_world(self):
        """One pass of the stageros main loop: UpdateWorld() then WorldCallback() (stageros.cpp:445-471,819-828)."""
        self.backend.step(self.cmd.copy())
        self.ticks += 1
        # the global watchdog runs at the top of WorldCallback, i.e. it acts on the NEXT tick's velocities
        if WATCHDOG > 0.0 and (self.sim_time - self.base_last_cmd) >= WATCHDOG - 1e-9:
            self.cmd[:] = 0.0
        return self.publish()

    def publish(self):
        """-> {topic: message} for every robot plus /clock (WorldCallback, stageros.cpp:473-783)."""
        b = self.backend
        pose = np.asarray(b.field("pose"), np.float64)
        speed = np.asarray(b.field("speed"), np.float64)
        scan = np.asarray(b.field("scan"), np.float32)
        crashed = np.asarray(b.field("crashed"))
        # intensities: Stage's return value of what the beam hit, cast to uint8 (stageros.cpp:501-506): the floorplan's 1
        # survives, a robot's 0.5 (ranger_return, stage1.world:95) and a miss become 0
        try:
            from_robot = np.asarray(b.field("hit_robot")).astype(bool)
        except AttributeError:
            from_robot = np.zeros(scan.shape, bool)
        intensity = ((scan < np.float32(RANGE_MAX)) & ~from_robot).astype(np.float32)
        now = self.sim_time
        out = {}
        prev = self.base_last_globalpos
        for r in range(self.R):
            x, y, a = (float(v) for v in pose[r])
            out[self.map_name("base_scan", r)] = LaserScan(
                Header(now, self.map_name("base_laser_link", r)), -FOV / 2.0, FOV / 2.0, FOV / (self.samples - 1),
                RANGE_MIN, RANGE_MAX, scan[r].astype(np.float32), intensity[r])
            out[self.map_name("odom", r)] = Odometry(
                Header(now, self.map_name("odom", r)), Pose(Vector3(x, y, 0.0), quaternion_from_yaw(a)),
                Twist(Vector3(float(speed[r, 0]), 0.0, 0.0), Vector3(0.0, 0.0, float(speed[r, 1]))))
            out[self.map_name("is_crashed", r)] = Int8(int(crashed[r]))
            # ground truth: velocity = finite difference against the previous publish (0 on the first one)
            gv = (0.0, 0.0, 0.0)
            if prev is not None:
                dT = DT
                gv = ((x - prev[r, 0]) / dT, (y - prev[r, 1]) / dT, normalize(a - prev[r, 2]) / dT)
            out[self.map_name("base_pose_ground_truth", r)] = Odometry(
                Header(now, self.map_name("odom", r)), Pose(Vector3(x, y, 0.0), quaternion_from_yaw(a)),
                Twist(Vector3(gv[0], gv[1], 0.0), Vector3(0.0, 0.0, gv[2])))
        self.base_last_globalpos = pose.copy()
        secs = int(math.floor(now + 1e-9))
        out["/clock"] = Clock(secs, int(round((now - secs) * 1e9)))
        return out
