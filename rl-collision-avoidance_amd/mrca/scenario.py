"""Scenario descriptions: the constants the reference hard-codes in stage_world1.py,
stage_world2.py, circle_world.py and worlds/*.world, as one plain data object.

The occupancy grids and pose tables are shipped as data files (``mrca/data``), produced from the
reference's world files by ``tools/make_maps.py``; nothing here reads /root/reference.
"""
import json
import os
from dataclasses import dataclass, field

import numpy as np

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

AUTO_NONE, AUTO_ROBOT, AUTO_GROUP = 0, 1, 2
RESET_TABLE, RESET_DISC, RESET_REGION = 0, 1, 2


@dataclass
class GridData:
    """Bit-packed occupancy grid: row 0 = lowest y, bit b of word w on a row = column 32*w+b."""
    bits: np.ndarray  # uint32 [height, words_per_row]
    width: int
    height: int
    cell: float
    x0: float
    y0: float

    @property
    def words_per_row(self):
        return int(self.bits.shape[1])

    def dense(self):
        b = ((self.bits[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)
        return b.reshape(self.height, -1)[:, : self.width]

    @classmethod
    def from_dense(cls, occ, cell, x0, y0):
        occ = np.asarray(occ, dtype=bool)
        h, w = occ.shape
        wpr = (w + 31) // 32
        pad = np.zeros((h, wpr * 32), dtype=bool)
        pad[:, :w] = occ
        words = (pad.reshape(h, wpr, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(-1)
        return cls(words.astype(np.uint32), w, h, float(cell), float(x0), float(y0))


def load_map(name):
    d = np.load(os.path.join(DATA_DIR, name + ".npz"))
    m = json.loads(str(d["meta"]))
    return GridData(np.ascontiguousarray(d["bits"], dtype=np.uint32), m["width"], m["height"], m["cell"], m["x0"],
                    m["y0"])


def load_tables():
    with open(os.path.join(DATA_DIR, "scenarios.json")) as f:
        return json.load(f)


@dataclass
class Scenario:
    name: str
    num_worlds: int
    robots_per_world: int
    grid: GridData
    timeout: int = 150            # stage_world1.py:206
    w_thresh: float = 1.05        # stage_world1.py:203
    pre_dist_zero: bool = False   # stage_world2.py:170-171 / circle_world.py:166-167
    auto_reset: int = AUTO_ROBOT
    seed: int = 0
    hold_velocity: bool = False    # fidelity: Stage's SetSpeed persistence (dead robots keep driving, speed survives a reset)
    collision_raster: float = 0.0  # fidelity mode: > 0 = robots collide when their outlines share a raster cell of this size
    beams: int = 512              # stage1.world:14
    frames: int = 3               # LASER_HIST ppo_stage1.py:24
    reset_mode: np.ndarray = field(default=None)
    goal_mode: np.ndarray = field(default=None)
    init_table: np.ndarray = field(default=None)  # [R,3]
    goal_table: np.ndarray = field(default=None)  # [R,2]
    group_id: np.ndarray = field(default=None)

    def __post_init__(self):
        R = self.robots_per_world
        if self.reset_mode is None:
            self.reset_mode = np.full(R, RESET_DISC, np.int32)
        if self.goal_mode is None:
            self.goal_mode = np.asarray(self.reset_mode, np.int32).copy()
        if self.init_table is None:
            self.init_table = np.zeros((R, 3), np.float64)
        if self.goal_table is None:
            self.goal_table = np.zeros((R, 2), np.float64)
        if self.group_id is None:
            self.group_id = np.zeros(R, np.int32)
        self.reset_mode = np.ascontiguousarray(self.reset_mode, np.int32)
        self.goal_mode = np.ascontiguousarray(self.goal_mode, np.int32)
        self.group_id = np.ascontiguousarray(self.group_id, np.int32)
        self.init_table = np.ascontiguousarray(self.init_table, np.float64).reshape(R, 3)
        self.goal_table = np.ascontiguousarray(self.goal_table, np.float64).reshape(R, 2)

    @property
    def num_robots(self):
        return self.num_worlds * self.robots_per_world


# The shipped maps exist at two cell sizes: the default (0.05 m for Stage-1/2, 0.1 m for the circle world: fine enough
# that walls keep their bitmap shape, small enough to stay L2-resident) and the reference's OWN Stage ``resolution``
# (worlds/stage1.world:3 and stage2.world:3: 0.2 m; circle.world:3: 0.01 m).  ``stage_resolution=True`` selects the
# latter: lidar ranges against walls and robot-vs-wall clearances are then quantised exactly as coarsely (or, for the
# circle world, as finely) as in Stage's raster.  Together with ``collision_raster`` (robot-robot collision by shared
# raster cells instead of exact rectangles) this is the fidelity mode for evaluating a policy under
# Stage-like clearances.
_STAGE_RES_MAPS = {"stage1_rink": "stage1_rink_r0200", "stage2_testenv": "stage2_testenv_r0200",
                   "circle_rink": "circle_rink_r0010"}


def _map(name, stage_resolution):
    return load_map(_STAGE_RES_MAPS[name] if stage_resolution else name)


def stage1(num_worlds=1, robots_per_world=24, seed=0, grid=None, stage_resolution=False, hold_velocity=False):
    """Stage-1 rink: random poses/goals in the 9 m disc (stage_world1.py), NUM_ENV=24
    (ppo_stage1.py:32), every robot restarts on its own (ppo_stage1.py:51-58)."""
    return Scenario("stage1", num_worlds, robots_per_world, grid or _map("stage1_rink", stage_resolution), timeout=150,
                    w_thresh=1.05, pre_dist_zero=False, auto_reset=AUTO_ROBOT, seed=seed,
                    collision_raster=0.2 if stage_resolution else 0.0, hold_velocity=hold_velocity)


def stage2(num_worlds=1, seed=0, grid=None, stage_resolution=False, hold_velocity=False):
    """Stage-2 map: 44 robots, tables for 0..33, random region for 34..43 (stage_world2.py:164-171,
    210-221), group-synchronous episodes (ppo_stage2.py:72-107, model/utils.py:83)."""
    tb = load_tables()["stage2"]
    R = tb["num_agents"]
    lo, hi = tb["random_index_range"]
    mode = np.full(R, RESET_TABLE, np.int32)
    mode[lo:hi] = RESET_REGION
    init = np.asarray(tb["init_pose"], np.float64)
    goal = np.zeros((R, 2))
    goal[: len(tb["goal_point"])] = np.asarray(tb["goal_point"], np.float64)
    bounds = tb["groups"]
    gid = np.zeros(R, np.int32)
    for g in range(len(bounds) - 1):
        gid[bounds[g]: bounds[g + 1]] = g
    return Scenario("stage2", num_worlds, R, grid or _map("stage2_testenv", stage_resolution), timeout=200, w_thresh=1.05,
                    pre_dist_zero=True, auto_reset=AUTO_GROUP, seed=seed, reset_mode=mode, goal_mode=mode.copy(),
                    init_table=init, goal_table=goal, group_id=gid, collision_raster=0.2 if stage_resolution else 0.0,
                    hold_velocity=hold_velocity)


def circle(num_worlds=1, seed=0, grid=None, stage_resolution=False):
    """Circle test: 50 robots on r = 25 m, antipodal goals (circle_world.py:164-167,205-208),
    omega-penalty threshold 0.7 (:195), timeout 10000 (:198), nothing resets (circle_test.py:36-83)."""
    tb = load_tables()["circle"]
    R = tb["num_agents"]
    mode = np.full(R, RESET_TABLE, np.int32)
    return Scenario("circle", num_worlds, R, grid or _map("circle_rink", stage_resolution), timeout=10000, w_thresh=0.7,
                    pre_dist_zero=True, auto_reset=AUTO_NONE, seed=seed, reset_mode=mode, goal_mode=mode.copy(),
                    init_table=np.asarray(tb["init_pose"], np.float64),
                    goal_table=np.asarray(tb["goal_point"], np.float64))


def empty_grid(cells=8, cell=1.0):
    """An open world: a tiny all-free grid at the origin (cells outside a grid are free)."""
    return GridData.from_dense(np.zeros((cells, cells), bool), cell, -0.5 * cells * cell, -0.5 * cells * cell)


def circle_big(num_robots, num_worlds=1, seed=0, spacing=None, grid=None):
    """The circle test as ONE circle of ``num_robots`` robots with the radius scaled so that the neighbour spacing
    of the reference's 50-robot / 25 m circle (2*pi*25/50 = 3.14 m) is kept: radius = 25 * num_robots / 50
    (SURVEY 8d C5 "single circles with radius proportional to R").  Poses as in model/utils.py:6-38: robot i at
    angle 2*pi*i/R facing the centre, goal = the antipodal point.  The reference's rink (60 m for r = 25 m) would
    be a 60 * R/50 m map -- 6000^2 cells at 500 robots -- whose wall sits 5 m BEHIND every robot; the scaled
    scenario runs in an open world instead (``grid`` overrides)."""
    R = int(num_robots)
    sp = 2.0 * np.pi * 25.0 / 50.0 if spacing is None else float(spacing)
    radius = sp * R / (2.0 * np.pi)
    ang = 2.0 * np.pi * np.arange(R) / R
    init = np.stack([radius * np.cos(ang), radius * np.sin(ang), ang + np.pi], 1)
    goal = -init[:, :2]
    mode = np.full(R, RESET_TABLE, np.int32)
    return Scenario("circle_big", num_worlds, R, grid or empty_grid(), timeout=1000000, w_thresh=0.7,
                    pre_dist_zero=True, auto_reset=AUTO_NONE, seed=seed, reset_mode=mode, goal_mode=mode.copy(),
                    init_table=init, goal_table=goal)


def circle_train(num_worlds=1, seed=0, timeout=900, grid=None):
    """The circle scenario as a TRAINING world (not in the reference, whose circle_world.py is evaluation only): same
    map, poses, goals and reward constants as ``circle`` (circle_world.py:164-208), but episodes end -- the 50 robots
    form one group that restarts together once every robot is done (Stage-2's group-synchronous rule,
    ppo_stage2.py:72-107) or after ``timeout`` ticks (50 m at 1 m/s are 500)."""
    sc = circle(num_worlds, seed, grid)
    sc.name, sc.timeout, sc.auto_reset = "circle_train", int(timeout), AUTO_GROUP
    return sc


def circle_n(robots, radius, num_worlds=1, seed=0, train=False, timeout=None, grid=None):
    """Smaller relatives of the circle test inside the same 60 m rink: ``robots`` (<= 64) robots evenly spaced on a
    circle of ``radius`` metres (<= 27), facing the centre, antipodal goals; reward constants of circle_world.py.
    ``train=True``: one group that restarts together (like ``circle_train``).  The paper's evaluation uses such circles
    of 4 ... 20 robots (Long et al. 2018, Sec. V); the reference checkout only ships the 50-robot table."""
    R = int(robots)
    ang = 2.0 * np.pi * np.arange(R) / R
    init = np.stack([radius * np.cos(ang), radius * np.sin(ang), ang + np.pi], 1)
    mode = np.full(R, RESET_TABLE, np.int32)
    t = int(timeout if timeout is not None else (40 * radius + 300 if train else 10000))
    return Scenario("circle_train" if train else "circle", num_worlds, R, grid or load_map("circle_rink"), timeout=t,
                    w_thresh=0.7, pre_dist_zero=True, auto_reset=AUTO_GROUP if train else AUTO_NONE, seed=seed,
                    reset_mode=mode, goal_mode=mode.copy(), init_table=init, goal_table=-init[:, :2])
