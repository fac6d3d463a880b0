"""Per-index facade: the reference's ``StageWorld(beam_num, index, num_env)`` object
(stage_world1.py:16-274, stage_world2.py, circle_world.py) as a view on row ``index`` of ONE
process-global batched device world, so the reference's training scripts run unchanged
(``python -m mrca.spmd -np 24 ppo_stage1.py``).

Every rank (a thread of the SPMD runtime, mrca/spmd.py) owns one facade.  ``control_vel`` latches
the rank's action; the tick fires when every rank is blocked in ``rospy.sleep`` or in an MPI
collective (all robots have had their say), exactly once per loop iteration of the scripts.
Getters return host copies in the reference's Python types.

The batched world is created by a *backend factory*.  The product factory is the HIP library
(``HipBackend``); there is no CPU fallback -- tests may inject their own checker backend through
``set_backend_factory``.
"""
import os
import threading

import numpy as np

from . import scenario as S

_backend_factory = None
_shared = {}
_shared_lock = threading.Lock()
_base_seed = 0


def set_seed(seed):
    """Base seed of the per-facade pose / goal generators (see StageWorldBase.rng)."""
    global _base_seed
    _base_seed = int(seed)


def set_backend_factory(factory):
    """factory(scenario) -> backend.  None restores the product default (HIP)."""
    global _backend_factory
    _backend_factory = factory
    with _shared_lock:
        _shared.clear()


class HipBackend:
    """The product backend: VecStageWorld on the current GPU, host copies on demand."""

    def __init__(self, sc):
        import torch
        from .vec_env import VecStageWorld
        self.torch = torch
        self.env = VecStageWorld(sc)
        self.N = sc.num_robots

    def reset(self, mask, poses, goals):
        t = self.torch
        dev = self.env.device
        self.env.reset(t.from_numpy(np.ascontiguousarray(mask, np.uint8)).to(dev),
                       None if poses is None else t.from_numpy(np.ascontiguousarray(poses, np.float32)).to(dev),
                       None if goals is None else t.from_numpy(np.ascontiguousarray(goals, np.float32)).to(dev))

    def step(self, actions):
        self.env.step(self.torch.from_numpy(np.ascontiguousarray(actions, np.float32)).to(self.env.device))

    def field(self, name):
        return getattr(self.env, name).cpu().numpy()


class SharedWorld:
    """One batched world of ``num_env`` robots shared by all facades of a process."""

    def __init__(self, variant, num_env, sc):
        self.variant, self.num_env, self.sc = variant, num_env, sc
        factory = _backend_factory or HipBackend
        self.backend = factory(sc)
        self.lock = threading.RLock()
        self.latched = np.zeros((num_env, 2), np.float32)
        self.has_cmd = np.zeros(num_env, bool)
        self.ticks = 0
        self.cache = {}
        self.backend.reset(np.ones(num_env, np.uint8), None, None)

    def field(self, name):
        with self.lock:
            if name not in self.cache:
                self.cache[name] = self.backend.field(name)
            return self.cache[name]

    def latch(self, index, action):
        with self.lock:
            self.latched[index] = (float(action[0]), float(action[1]))
            self.has_cmd[index] = True

    def pending(self):
        return bool(self.has_cmd.any())

    def tick(self):
        """One Stage tick with the latched commands.  A robot that sent no cmd_vel this round
        keeps its previous command in Stage (stageros.cpp:272-280; the watchdog is global,
        :466-471); here it idles with (0,0) -- DESIGN.md "Oracle decisions" 7 -- unless the scenario
        asks for Stage's behaviour (``hold_velocity``, MRCA_HOLD_VELOCITY=1): then the last command stays."""
        with self.lock:
            if getattr(self.sc, "hold_velocity", False):
                act = self.latched.astype(np.float32).copy()
            else:
                act = np.where(self.has_cmd[:, None], self.latched, 0.0).astype(np.float32)
            self.backend.step(act)
            self.has_cmd[:] = False
            self.ticks += 1
            self.cache.clear()

    def reset_positions(self):
        """The ``reset_positions`` service (stageros.cpp:260-269): every model back at its world-file pose
        (worlds/*.world agent lines), stall flags cleared; goals stay."""
        with self.lock:
            table = S.load_tables()[self.variant]["world_agents"]
            poses = np.asarray(table, np.float32)[: self.num_env].copy()
            poses[:, 2] = np.arctan2(np.sin(poses[:, 2]), np.cos(poses[:, 2]))
            goals = self.field("goal").astype(np.float32).copy()
            self.backend.reset(np.ones(self.num_env, np.uint8), poses, goals)
            self.cache.clear()

    def teleport(self, index, pose=None, goal=None):
        with self.lock:
            mask = np.zeros(self.num_env, np.uint8)
            mask[index] = 1
            poses = self.field("pose").astype(np.float32).copy()
            goals = self.field("goal").astype(np.float32).copy()
            if pose is not None:
                poses[index] = pose
            if goal is not None:
                goals[index] = goal
            self.backend.reset(mask, poses, goals)
            self.cache.clear()


def shared_world(variant, num_env):
    with _shared_lock:
        key = (variant, num_env)
        if key not in _shared:
            if variant == "stage1":
                sc = S.stage1(num_worlds=1, robots_per_world=num_env)
            elif variant == "stage2":
                sc = S.stage2(num_worlds=1)
            else:
                sc = S.circle(num_worlds=1)
            if sc.robots_per_world != num_env:
                raise ValueError(f"{variant} world has {sc.robots_per_world} robots, NUM_ENV={num_env}")
            sc.auto_reset = S.AUTO_NONE  # the calling script owns the episode structure
            sc.hold_velocity = os.environ.get("MRCA_HOLD_VELOCITY", "0") not in ("", "0")
            _shared[key] = SharedWorld(variant, num_env, sc)
        return _shared[key]


_RESULT = {0: 0, 1: "Reach Goal", 2: "Crashed", 3: "Time out"}


class StageWorldBase:
    VARIANT = "stage1"

    def __init__(self, beam_num, index, num_env):
        self.index = index
        self.num_env = num_env
        self.beam_mum = beam_num          # sic: the reference's attribute name (stage_world1.py:23)
        self.laser_cb_num = 0
        self.self_speed = [0.0, 0.0]
        self.step_goal = [0.0, 0.0]
        self.step_r_cnt = 0.0
        self.map_size = np.array([8.0, 8.0], dtype=np.float32)
        self.goal_size = 0.5
        self.robot_value = 10.0
        self.goal_value = 0.0
        self.init_pose = None
        self.goal_point = [0.0, 0.0]
        self.pre_distance = 0.0
        self.distance = 0.0
        self.world = shared_world(self.VARIANT, num_env)
        # The reference draws poses / goals from the process-global ``np.random`` of its own MPI process
        # (stage_world1.py:251-274).  Ranks are threads here, so every facade owns its generator -- seeded from
        # (base seed, rank): a run is reproducible whatever order the rank threads are scheduled in.
        self.rng = np.random.RandomState((_base_seed * 1000003 + 7919 * index + 17) & 0x7FFFFFFF)

    # ---- raw getters (stage_world1.py:116-153)
    def get_self_stateGT(self):
        return [float(v) for v in self.world.field("pose")[self.index]]

    def get_self_speedGT(self):
        return [float(v) for v in self.world.field("speed_gt")[self.index]]

    def get_self_state(self):
        return self.get_self_stateGT()      # localization "gps": odom == ground truth (stage1.world:85)

    def get_self_speed(self):
        return [float(v) for v in self.world.field("speed")[self.index]]

    def get_crash_state(self):
        return int(self.world.field("crashed")[self.index])

    def get_sim_time(self):
        return 0.1 * self.world.ticks

    def get_laser_observation(self):
        """stage_world1.py:122-140: NaN/inf -> 6, left half ascending / right half descending
        sub-sampling to beam_num, scan/6 - 0.5, float64."""
        scan = np.array(self.world.field("scan")[self.index], dtype=np.float64)
        scan[~np.isfinite(scan)] = 6.0
        raw, sparse = len(scan), self.beam_mum
        step = float(raw) / sparse
        left = [scan[int(i * step)] for i in range(int(sparse / 2))]
        right = [scan[int(raw - 1.0 - i * step)] for i in range(int(sparse / 2))]
        return np.concatenate((left, right[::-1]), axis=0) / 6.0 - 0.5

    def get_local_goal(self):
        x, y, theta = self.get_self_stateGT()
        gx, gy = self.goal_point
        return [(gx - x) * np.cos(theta) + (gy - y) * np.sin(theta),
                -(gx - x) * np.sin(theta) + (gy - y) * np.cos(theta)]

    # ---- commands
    def control_vel(self, action):
        self.world.latch(self.index, action)

    def control_pose(self, pose):
        assert len(pose) == 3
        th = float(np.arctan2(np.sin(pose[2]), np.cos(pose[2])))   # quaternion round trip
        self.world.teleport(self.index, pose=[pose[0], pose[1], th])

    def reset_world(self):
        """stage_world1.py:162-168: the reset_positions service (stageros.cpp:260-269: every model back at its world-file
        pose, stall flags cleared), then the bookkeeping fields."""
        self.world.reset_positions()
        self.self_speed = [0.0, 0.0]
        self.step_goal = [0.0, 0.0]
        self.step_r_cnt = 0.0

    def reset_pose(self):
        self.control_pose(self._new_pose())

    def generate_goal_point(self):
        self.goal_point = list(self._new_goal())
        self.world.teleport(self.index, goal=self.goal_point)
        x, y = self.get_local_goal()
        self.pre_distance = float(self.world.field("prev_dist")[self.index])
        self.distance = self.pre_distance

    def get_reward_and_terminate(self, t):
        """stage_world1.py:180-211.  The reward is evaluated on the device during the tick (it does not depend
        on ``t``); the terminal flag and the result string are assembled here from the device's distance and
        stall flag and the CALLER's step counter ``t``, with the reference's precedence (Reach Goal < Crashed <
        Time out) -- so a script whose counter differs from the device's own gets the timeout it asked for."""
        w = self.world
        reward = float(w.field("reward")[self.index])
        self.pre_distance = self.distance
        self.distance = float(w.field("prev_dist")[self.index])
        terminate, result = False, 0
        if self.distance < self.goal_size:
            terminate, result = True, "Reach Goal"
        if int(w.field("crashed")[self.index]) == 1:
            terminate, result = True, "Crashed"
        if t > w.sc.timeout:
            terminate, result = True, "Time out"
        return reward, terminate, result

    # ---- variant hooks
    def _new_pose(self):
        return self.generate_random_pose()

    def _new_goal(self):
        return self.generate_random_goal()

    def generate_random_pose(self):
        """stage_world1.py:251-260 (host numpy RNG like the reference; one generator per facade)."""
        while True:
            x, y = self.rng.uniform(-9, 9), self.rng.uniform(-9, 9)
            if np.sqrt(x ** 2 + y ** 2) <= 9:
                return [x, y, self.rng.uniform(0, 2 * np.pi)]

    def generate_random_goal(self):
        """stage_world1.py:262-274."""
        self.init_pose = self.get_self_stateGT()
        while True:
            x, y = self.rng.uniform(-9, 9), self.rng.uniform(-9, 9)
            d_o = np.sqrt(x ** 2 + y ** 2)
            d_g = np.sqrt((x - self.init_pose[0]) ** 2 + (y - self.init_pose[1]) ** 2)
            if not (d_o > 9 or d_g > 10 or d_g < 8):
                return [x, y]


class Stage1World(StageWorldBase):
    VARIANT = "stage1"


class _RegionMixin:
    def _region_point(self):
        """stage_world2.py:250-287: x~U(9,19), y in two bands, >= 7 m from the robot."""
        x_r, y_r, _ = self.get_self_stateGT()
        while True:
            x = self.rng.uniform(9, 19)
            y = self.rng.uniform(0, 1)
            y = -(y * 10 + 1) if y <= 0.4 else -(y * 10 + 9)
            if not np.sqrt((x - x_r) ** 2 + (y - y_r) ** 2) < 7:
                return x, y

    def generate_random_pose(self):
        x, y = self._region_point()
        return [x, y, self.rng.uniform(0, 2 * np.pi)]

    def generate_random_goal(self):
        return list(self._region_point())


class Stage2World(_RegionMixin, StageWorldBase):
    VARIANT = "stage2"

    def _tables(self):
        return self.world.sc.init_table, self.world.sc.goal_table

    def _new_pose(self):                      # stage_world2.py:210-215
        if 33 < self.index < 44:
            return self.generate_random_pose()
        return [float(v) for v in self._tables()[0][self.index]]

    def _new_goal(self):                      # stage_world2.py:164-168
        if 33 < self.index < 44:
            return self.generate_random_goal()
        return [float(v) for v in self._tables()[1][self.index]]


class CircleWorld(_RegionMixin, StageWorldBase):
    VARIANT = "circle"

    def _new_pose(self):                      # circle_world.py:205-208
        return [float(v) for v in self.world.sc.init_table[self.index]]

    def _new_goal(self):                      # circle_world.py:164-167
        return [float(v) for v in self.world.sc.goal_table[self.index]]
