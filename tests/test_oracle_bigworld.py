"""Worlds with more than 64 robots (single circles with radius proportional to R; 8e row 3): the
oracle's distance-culled passes must equal its all-pairs passes, the C restatement must equal the NumPy oracle, and
the one-world-on-several-ranks tick (mrca.sharded.ShardedWorld: one all-gather of the commands, replicated move,
sharded lidar) must reproduce the single-process world -- here with two gloo ranks on the CPU, the C oracle standing
in for the device env behind the same ``step(actions, ray_slice)`` surface."""
import os
import socket
import tempfile

import numpy as np
import torch
import torch.multiprocessing as mp

import util as U
from util import S, O


def _run(sc, steps, seed, big_threshold):
    old = O.BIG_WORLD
    O.BIG_WORLD = big_threshold
    try:
        env = U.oracle_env(sc, np.float32)
        env.reset()
        rng = np.random.default_rng(seed)
        for _ in range(steps):
            env.step(U.random_actions(rng, sc.num_robots))
    finally:
        O.BIG_WORLD = old
    return env


def test_culled_passes_equal_all_pairs_passes():
    sc = S.stage1(num_worlds=2, robots_per_world=70, seed=5)      # 70 robots in the 9 m disc: crowded
    brute = _run(sc, 10, 0, big_threshold=10 ** 9)
    culled = _run(sc, 10, 0, big_threshold=64)
    U.assert_state_equal(culled, brute, what="culled vs all-pairs")
    assert brute.episode.max() >= 2          # robots did crash and restart on the way


def test_c_oracle_equals_numpy_oracle_on_big_worlds():
    for sc in (S.stage1(num_worlds=2, robots_per_world=90, seed=7), S.circle_big(120, spacing=0.9)):
        a, b = U.oracle_env(sc, np.float32), U.COracleEnv(sc)
        a.reset()
        b.reset()
        U.assert_state_equal(b, a, what=f"{sc.name} reset")
        rng = np.random.default_rng(3)
        for k in range(8):
            act = U.random_actions(rng, sc.num_robots)
            a.step(act)
            b.step(act)
            U.assert_state_equal(b, a, what=f"{sc.name} step {k}")


def test_circle_big_keeps_the_reference_spacing_and_poses():
    tb = S.load_tables()["circle"]
    sc = S.circle_big(50)
    assert np.abs(sc.init_table - np.asarray(tb["init_pose"])).max() < 0.006    # the reference rounds to 2 decimals
    assert np.abs(sc.goal_table - np.asarray(tb["goal_point"])).max() < 0.006
    big = S.circle_big(500)
    d = np.hypot(*(big.init_table[1, :2] - big.init_table[0, :2]))
    assert abs(d - 3.1415) < 1e-3 and abs(np.hypot(*big.init_table[0, :2]) - 250.0) < 1e-9


# ------------------------------------------------------------------------------------------------
class _CpuWorld:
    """The VecStageWorld surface ShardedWorld needs, on the C oracle (tests only)."""

    def __init__(self, sc):
        self.env = U.COracleEnv(sc)
        self.N = sc.num_robots
        self.device = torch.device("cpu")
        for k in U.STATE_FIELDS:
            setattr(self, k, torch.from_numpy(getattr(self.env, k)))

    def reset(self):
        self.env.reset()

    def policy_obs(self):
        """The oracle keeps the stacks in deque order: that IS a ring whose newest frame sits in the last slot."""
        return self.obs, torch.full((self.N,), self.obs.shape[1] - 1, dtype=torch.uint8)

    def step(self, actions, ray_slice=None):
        lo, cnt = ray_slice if ray_slice is not None else (0, self.N)
        keep = {k: getattr(self.env, k).copy() for k in ("scan", "obs", "local_goal")}
        self.env.step(actions.numpy())
        for k, v in keep.items():             # mrca_step_slice leaves the lidar outputs of the other robots untouched
            arr = getattr(self.env, k)
            arr[:lo] = v[:lo]
            arr[lo + cnt:] = v[lo + cnt:]


def _actions(k, n):
    g = torch.Generator().manual_seed(100 + k)
    return torch.stack([torch.rand(n, generator=g), torch.rand(n, generator=g) * 2 - 1], 1)


def _worker(rank, world, port, out, n_robots):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mrca.sharded import ShardedWorld
    sc = S.circle_big(n_robots, spacing=0.9)
    sw = ShardedWorld(_CpuWorld(sc), dist)
    sw.reset()
    for k in range(6):
        sw.step(_actions(k, n_robots)[sw.lo:sw.hi])          # every rank only knows its own robots' commands
    assert torch.equal(sw.obs(), sw.local("obs"))            # the slice's stacks gathered from its ring rows only
    torch.save({"lo": sw.lo, "hi": sw.hi, "obs": sw.obs().clone(), "pose": sw.env.pose.clone(),
                "reward": sw.env.reward.clone(), "scan": sw.local("scan").clone()}, f"{out}.{rank}")
    dist.destroy_process_group()


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_one_world_on_two_ranks_equals_the_single_process_world():
    n = 101                                     # odd: the second rank's slice is shorter
    out = os.path.join(tempfile.mkdtemp(), "shard")
    mp.spawn(_worker, args=(2, _port(), out, n), nprocs=2, join=True)
    ref = _CpuWorld(S.circle_big(n, spacing=0.9))
    ref.reset()
    for k in range(6):
        ref.step(_actions(k, n))
    covered = 0
    for r in range(2):
        d = torch.load(f"{out}.{r}")
        lo, hi = d["lo"], d["hi"]
        assert torch.equal(d["pose"], ref.pose) and torch.equal(d["reward"], ref.reward)   # replicated state
        assert torch.equal(d["obs"], ref.obs[lo:hi]) and torch.equal(d["scan"], ref.scan[lo:hi])   # sharded lidar
        covered += hi - lo
    assert covered == n


def test_slice_bounds_cover_every_robot_once():
    from mrca.sharded import slice_bounds
    for n, size in ((101, 2), (5, 4), (50000, 8), (64, 8), (3, 8)):
        seen = []
        for r in range(size):
            lo, hi, per = slice_bounds(n, size, r)
            seen += list(range(lo, hi))
            assert hi - lo <= per
        assert seen == list(range(n))


# ------------------------------------------------------------------------------------------------
# fidelity mode: Stage's raster rule for robot-robot collisions + maps at the reference's own Stage resolutions
def test_raster_collision_c_oracle_equals_numpy_oracle_and_is_stricter():
    sc = S.stage1(num_worlds=2, robots_per_world=24, seed=5, stage_resolution=True)
    assert sc.grid.cell == 0.2 and sc.collision_raster == 0.2          # worlds/stage1.world:3
    exact = S.stage1(num_worlds=2, robots_per_world=24, seed=5, stage_resolution=True)
    exact.collision_raster = 0.0
    a, b, c = U.oracle_env(sc, np.float32), U.COracleEnv(sc), U.oracle_env(exact, np.float32)
    for e in (a, b, c):
        e.reset()
    rng = np.random.default_rng(1)
    crashes_raster = crashes_exact = 0
    for k in range(40):
        act = U.random_actions(rng, sc.num_robots)
        for e in (a, b, c):
            e.step(act)
        U.assert_state_equal(b, a, what=f"raster step {k}")
        crashes_raster += int(((a.result == 2) & (a.done == 1)).sum())
        crashes_exact += int(((c.result == 2) & (c.done == 1)).sum())
    assert crashes_raster > crashes_exact > 0        # outlines that share a 0.2 m cell collide up to a cell apart


def test_outline_cells_contain_the_corner_cells_and_stay_near_the_footprint():
    f = np.float32
    rng = np.random.default_rng(0)
    for _ in range(200):
        x, y, th = f(rng.uniform(-5, 5)), f(rng.uniform(-5, 5)), f(rng.uniform(-np.pi, np.pi))
        s, c = O.sincos(np.array([th], f), f)
        cells = O.outline_cells(0.2, x, y, s[0], c[0], f)
        cx, cy = O.footprint_corners(np.array([x]), np.array([y]), s, c, f)
        for k in range(4):
            assert (int(np.floor(cx[0, k] * f(5.0))), int(np.floor(cy[0, k] * f(5.0)))) in cells
        for (ix, iy) in cells:      # every cell lies within circumradius + one cell diagonal of the centre
            assert np.hypot((ix + 0.5) * 0.2 - x, (iy + 0.5) * 0.2 - y) < 0.2907 + 0.2 * 1.4143
        assert 6 <= len(cells) <= 24
