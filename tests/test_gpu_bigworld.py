"""GPU: worlds with more than 64 robots -- per-robot threads, per-tick spatial hashes, the
ordered collision pass as dependency rounds, chunked lidar neighbour lists -- bit-exact against the C oracle, and
the sharded tick (mrca_step_slice)."""
import numpy as np
import pytest
import torch

import util as U
from util import S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.build()
    from mrca import vec_env
    return vec_env


def _exact(hip, sc, steps, seed, poses=None, check_every=4, actions=None):
    env = hip.VecStageWorld(sc)
    ora = U.COracleEnv(sc)
    if poses is None:
        env.reset()
        ora.reset()
    else:
        mask = np.ones(sc.num_robots, np.uint8)
        env.reset(torch.from_numpy(mask).cuda(), torch.from_numpy(poses).cuda(), None)
        ora.reset(mask, poses, None)
    torch.cuda.synchronize()
    U.assert_state_equal(U.HostView(env), ora, what=f"{sc.name} reset")
    rng = np.random.default_rng(seed)
    for k in range(steps):
        a = U.random_actions(rng, sc.num_robots) if actions is None else actions(k, ora)
        env.step(torch.from_numpy(a).cuda())
        ora.step(a)
        if k % check_every == check_every - 1 or k == steps - 1:
            torch.cuda.synchronize()
            U.assert_state_equal(U.HostView(env), ora, what=f"{sc.name} step {k}")
            U.assert_hits_equal(env, ora, what=f"{sc.name} step {k}")
    env.check()          # mrca_check: the ordered collision pass never gave up on a robot
    env.close()
    return ora


def test_crowded_rink_200_robots_bit_exact(hip):
    """200 robots in the 9 m Stage-1 disc (per-robot restarts): constant collisions, long dependency chains in the
    ordered pass, > 64 lidar neighbours per robot (chunked lists)."""
    o = _exact(hip, S.stage1(num_worlds=2, robots_per_world=200, seed=3), 24, 1)
    assert o.episode.max() >= 3


@pytest.mark.parametrize("radius", [9.98, 13.0])
def test_robots_outside_a_walled_map_bit_exact(hip, radius):
    """A big world whose robots stand OUTSIDE the map: a walled 8 m box (0.5 m cells) at the origin, 66 robots on a circle
    round it driving inwards.  At 9.98 m the box's corners and faces come into the 6 m lidar reach within the first ticks
    (beams hit its walls from outside), at 13 m it stays out of reach for a while and then comes in: both sides of the
    ray cast's "the map is out of this robot's reach: nothing to march through" test (raycast_kernel, big worlds), every
    field and every hit flag bit-exact against the C oracle's plain cell walk."""
    from mrca.scenario import GridData
    occ = np.zeros((16, 16), bool)
    occ[0, :] = occ[-1, :] = occ[:, 0] = occ[:, -1] = True
    occ[6:9, 7] = True                               # something inside the box too (seen through nothing: walls are closed)
    grid = GridData.from_dense(occ, 0.5, -4.0, -4.0)
    n = 66
    sc = S.circle_big(n, spacing=2.0 * np.pi * radius / n, grid=grid)
    o = _exact(hip, sc, 48, 0, check_every=2, actions=_go_to_goal(sc))
    scan = np.asarray(o.scan)
    assert (scan < 6.0).any(), "no beam ever saw the box: the test does not test what it says"
    assert (scan[:, :] >= 6.0).any()


def test_single_circle_500_robots_bit_exact(hip):
    """One circle of 500 robots, radius proportional to R (250 m, spacing 3.14 m), open world, the commands of a
    go-to-goal controller: the scenario of SURVEY 8d C5."""
    sc = S.circle_big(500)

    def act(k, ora):
        lg = ora.local_goal
        bearing = np.arctan2(lg[:, 1], lg[:, 0])
        return np.stack([np.full(sc.num_robots, 1.0), np.clip(2.0 * bearing, -1, 1)], 1).astype(np.float32)

    _exact(hip, sc, 40, 0, actions=act, check_every=8)


def _go_to_goal(sc):
    def act(k, ora):
        lg = ora.local_goal
        bearing = np.arctan2(lg[:, 1], lg[:, 0])
        return np.stack([np.full(sc.num_robots, 1.0), np.clip(2.0 * bearing, -1, 1)], 1).astype(np.float32)
    return act


def _lattice_jam(n, seed):
    """n robots packed on a jittered 0.8 m lattice with random headings: the centre of the big circle when everybody
    arrives."""
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(n)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2)[:n]
    xy = (ij - side / 2) * 0.8 + rng.uniform(-0.12, 0.12, (n, 2))
    return np.concatenate([xy, rng.uniform(-np.pi, np.pi, (n, 1))], 1).astype(np.float32)


@pytest.mark.parametrize("n", [5000, 50000])
def test_single_circle_at_the_quoted_sizes_bit_exact(hip, n):
    """The sizes the throughput figures are quoted on (one circle of 5 000 and of 50 000 robots, radius
    proportional to R): the 131 072-bucket hashes, the multi-block scan of the lidar hash, neighbour chunks crossing many
    buckets.  12 ticks of the go-to-goal controller, EVERY field of EVERY robot bit-exact against the C oracle (its
    culled passes finish a 50 000-robot tick in well under a second per host thread team), status word clear."""
    sc = S.circle_big(n)
    _exact(hip, sc, 12, 0, actions=_go_to_goal(sc), check_every=4)


def test_jam_of_50000_robots_bit_exact(hip):
    """50 000 robots on the 0.8 m lattice (179 m across), random commands: every robot in a dependency chain of the
    ordered pass, > 150 lidar neighbours each, the collision hash at its working size.  8 ticks, bit-exact."""
    n = 50000
    sc = S.circle_big(n)
    o = _exact(hip, sc, 8, 5, poses=_lattice_jam(n, 13), check_every=4)
    assert o.crashed.sum() > 2000


def test_jam_of_500_robots_bit_exact(hip):
    """The end game of the big circle: 500 robots packed on a jittered 0.8 m lattice (the centre of the circle when
    everybody arrives), random commands: > 150 robots within lidar reach of each other, collisions everywhere."""
    sc = S.circle_big(500)
    rng = np.random.default_rng(11)
    side = 23
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2)[:500]
    xy = (ij - side / 2) * 0.8 + rng.uniform(-0.12, 0.12, (500, 2))
    poses = np.concatenate([xy, rng.uniform(-np.pi, np.pi, (500, 1))], 1).astype(np.float32)
    o = _exact(hip, sc, 20, 2, poses=poses)
    assert o.crashed.sum() > 20


def test_the_jam_leaves_the_status_word_clear_and_a_raised_word_is_reported(hip):
    """The ordered collision pass waits (bounded) for lower-indexed robots; had it ever given up, the env's sticky status
    word would say so and mrca_check would fail loudly.  Through the jam -- where almost every robot is in a dependency
    chain -- the word must stay clear; a word raised by hand must come back as an error, once."""
    sc = S.circle_big(500)
    rng = np.random.default_rng(11)
    ij = np.stack(np.meshgrid(np.arange(23), np.arange(23)), -1).reshape(-1, 2)[:500]
    xy = (ij - 23 / 2) * 0.8 + rng.uniform(-0.12, 0.12, (500, 2))
    poses = np.concatenate([xy, rng.uniform(-np.pi, np.pi, (500, 1))], 1).astype(np.float32)
    env = hip.VecStageWorld(sc)
    env.reset(torch.ones(500, dtype=torch.uint8, device="cuda"), torch.from_numpy(poses).cuda(), None)
    g = torch.Generator(device="cpu").manual_seed(3)
    for _ in range(30):
        env.step(torch.stack([torch.rand(500, generator=g), torch.rand(500, generator=g) * 2 - 1], 1).float().cuda())
    env.check()
    assert int(env.crashed.sum()) > 20
    # raise the word by hand (the last 4 bytes the layout hands out: behind every field)
    status = env.arena[-256:].view(torch.int32)
    assert int(status.abs().sum()) == 0
    status[0] = 1
    with pytest.raises(RuntimeError, match="collision pass"):
        env.check()
    env.check()                                     # reported once, then clear again
    env.close()


def test_two_big_worlds_are_independent(hip):
    sc2 = S.circle_big(100, num_worlds=2, spacing=0.9)
    sc1 = S.circle_big(100, num_worlds=1, spacing=0.9)
    a, b = hip.VecStageWorld(sc2), hip.VecStageWorld(sc1)
    a.reset()
    b.reset()
    g = torch.Generator(device="cpu").manual_seed(4)
    for _ in range(10):
        act = torch.stack([torch.rand(200, generator=g), torch.rand(200, generator=g) * 2 - 1], 1).float().cuda()
        a.step(act)
        b.step(act[:100].contiguous())
    for f in ("pose", "scan", "reward", "crashed"):
        assert torch.equal(getattr(a, f)[:100], getattr(b, f)), f
    a.close()
    b.close()


def test_step_slice_casts_only_the_slice(hip):
    """mrca_step_slice: every robot advances (state identical to a full step), the lidar outputs change only for the
    robots of the slice -- and there they equal the full step's."""
    sc = S.circle_big(150, spacing=0.9)
    full, part = hip.VecStageWorld(sc), hip.VecStageWorld(sc)
    full.reset()
    part.reset()
    g = torch.Generator(device="cpu").manual_seed(9)
    lo, cnt = 40, 70
    for _ in range(6):
        act = torch.stack([torch.rand(150, generator=g), torch.rand(150, generator=g) * 2 - 1], 1).float().cuda()
        before = part.obs.clone()
        full.step(act)
        part.step(act, ray_slice=(lo, cnt))
        for f in ("pose", "speed", "speed_gt", "reward", "done", "crashed", "t"):
            assert torch.equal(getattr(full, f), getattr(part, f)), f
        assert torch.equal(part.scan[lo:lo + cnt], full.scan[lo:lo + cnt])
        assert torch.equal(part.obs[lo:lo + cnt, -1], full.obs[lo:lo + cnt, -1])
        assert torch.equal(part.obs[:lo], before[:lo]) and torch.equal(part.obs[lo + cnt:], before[lo + cnt:])
    with pytest.raises(RuntimeError, match="slice"):
        part.step(act, ray_slice=(100, 100))
    full.close()
    part.close()


def test_group_synchronous_big_worlds_are_refused(hip):
    sc = S.stage1(num_worlds=1, robots_per_world=65)
    sc.auto_reset = S.AUTO_GROUP
    with pytest.raises(RuntimeError, match="robots_per_world"):
        hip.VecStageWorld(sc)
