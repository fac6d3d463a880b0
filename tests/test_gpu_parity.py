"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(libmrca_env.so via mrca.vec_env), against the oracle.

Bars
  * fp32 mode of the oracle (same operation order): BIT-EXACT on every field, flags included.
  * fp64 mode of the oracle (clean maths), one tick from identical state: pose / goal / reward /
    local goal within 1e-5 (north-star tolerance); scan within 1e-5 on >= 99.9 % of beams.  Ranges are
    QUANTISED to the cell grid (entry distance of the first occupied cell, like Stage's raster): a beam that
    grazes a cell corner enters a different first cell in fp32 than in fp64, so the remaining beams (measured
    2.7e-4 of them) differ by up to the extent of one cell along the ray -- bounded below by 1.5 cells
    (0.075 m); a robot-robot slab hit is continuous and stays within 1e-5.  Flags equal on >= 99.5 % of robots.
  * full BASELINE sizes: the oracle checks a slice of worlds bit-exactly (worlds are independent)
    and the whole batch is checked through size-independent properties.
"""
import numpy as np
import pytest
import torch

import util as U
from util import S, O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.build()
    from mrca import vec_env
    return vec_env


def _run_exact(hip, sc, steps, seed, check_every=1):
    env = hip.VecStageWorld(sc)
    ora = U.oracle_env(sc, np.float32)
    env.reset()
    ora.reset()
    torch.cuda.synchronize()
    U.assert_state_equal(U.HostView(env), ora, what=f"{sc.name} reset")
    rng = np.random.default_rng(seed)
    for k in range(steps):
        a = U.random_actions(rng, sc.num_robots)
        env.step(torch.from_numpy(a).cuda())
        ora.step(a)
        if k % check_every == 0 or k == steps - 1:
            torch.cuda.synchronize()
            U.assert_state_equal(U.HostView(env), ora, what=f"{sc.name} step {k}")
            U.assert_hits_equal(env, ora, what=f"{sc.name} step {k}")
    env.close()
    return ora


def test_stage1_bit_exact(hip):
    o = _run_exact(hip, S.stage1(num_worlds=3, robots_per_world=8, seed=11), 160, 3)
    assert o.episode.max() >= 2


def test_stage1_reference_size_24_robots(hip):
    """configs[0]: the reference's own case, 24 robots in the Stage-1 rink (ppo_stage1.py:32)."""
    _run_exact(hip, S.stage1(num_worlds=1, robots_per_world=24, seed=1), 60, 7, check_every=4)


def test_stage2_bit_exact_group_episodes(hip):
    o = _run_exact(hip, S.stage2(num_worlds=1, seed=5), 215, 3, check_every=5)
    assert o.episode.max() >= 2


def test_stage2_hold_velocity_bit_exact(hip):
    """mrca_config.hold_velocity: dead robots keep driving at their last command until the group restarts (stageros'
    SetSpeed persistence under ppo_stage2.py:72-74), and the speed input survives the restart."""
    o = _run_exact(hip, S.stage2(num_worlds=2, seed=5, hold_velocity=True), 215, 3, check_every=5)
    assert o.episode.max() >= 2
    _run_exact(hip, S.stage1(num_worlds=3, robots_per_world=8, seed=11, hold_velocity=True), 100, 3, check_every=4)


def test_stage2_two_worlds(hip):
    _run_exact(hip, S.stage2(num_worlds=2, seed=8), 30, 5, check_every=3)


def test_circle_bit_exact(hip):
    _run_exact(hip, S.circle(num_worlds=1, seed=2), 30, 4)


def test_odd_sizes_and_synthetic_map(hip):
    """ragged sizes: 1 robot, 64 robots per world (full wavefront), 5 frames, 256 beams."""
    g = U.small_grid(cell=0.1, size=16.0, ring_radius=7.0, blocks=[(-1.0, -1.0, 1.0, 0.5)])
    for R, W, frames, beams in ((1, 5, 3, 512), (64, 1, 3, 512), (7, 3, 5, 256)):
        sc = S.stage1(num_worlds=W, robots_per_world=R, seed=21 + R, grid=g)
        sc.frames, sc.beams = frames, beams
        _run_exact(hip, sc, 25, R, check_every=2)
    # a fine grid (2.5 cm cells), the widest scan (1024 beams) and the deepest frame stack (8)
    gf = U.small_grid(cell=0.025, size=14.0, ring_radius=6.2, blocks=[(0.5, 0.5, 1.5, 1.0)])
    sc = S.stage1(num_worlds=2, robots_per_world=5, seed=77, grid=gf)
    sc.frames, sc.beams = 8, 1024
    env = hip.VecStageWorld(sc)
    ora = U.COracleEnv(sc)
    rng = np.random.default_rng(3)
    poses = np.stack([rng.uniform(-4, 4, 10), rng.uniform(-4, 4, 10), rng.uniform(-3, 3, 10)], 1).astype(np.float32)
    goals = rng.uniform(-4, 4, (10, 2)).astype(np.float32)
    env.reset(None, torch.from_numpy(poses).cuda(), torch.from_numpy(goals).cuda())
    ora.reset(None, poses, goals)
    for k in range(20):
        a = U.random_actions(rng, 10)
        env.step(torch.from_numpy(a).cuda())
        ora.step(a)
    torch.cuda.synchronize()
    U.assert_state_equal(U.HostView(env), ora, what="fine grid / 1024 beams / 8 frames")
    env.close()


def test_masked_reset_and_overrides(hip):
    sc = S.stage1(num_worlds=1, robots_per_world=6, seed=9)
    env = hip.VecStageWorld(sc)
    ora = U.oracle_env(sc)
    env.reset()
    ora.reset()
    mask = np.array([1, 0, 1, 0, 0, 1], np.uint8)
    poses = np.zeros((6, 3), np.float32)
    poses[:, 0] = np.arange(6) - 2.5
    poses[:, 2] = 0.3
    goals = np.tile(np.array([[4.0, 4.0]], np.float32), (6, 1))
    env.reset(torch.from_numpy(mask).cuda(), torch.from_numpy(poses).cuda(), torch.from_numpy(goals).cuda())
    ora.reset(mask, poses, goals)
    torch.cuda.synchronize()
    U.assert_state_equal(U.HostView(env), ora, what="masked reset")
    env.close()


def _fp64_scenarios():
    return {
        "stage1": (lambda: S.stage1(num_worlds=8, robots_per_world=24, seed=4), 40),
        "stage2": (lambda: S.stage2(num_worlds=1, seed=4), 40),
        "circle": (lambda: S.circle(num_worlds=1, seed=4), 30),
        "big_world_500": (lambda: S.stage1(num_worlds=1, robots_per_world=500, seed=4), 12),
        "stage1_fidelity": (lambda: S.stage1(num_worlds=4, robots_per_world=24, seed=4, stage_resolution=True), 40),
    }


@pytest.mark.parametrize("name", ["stage1", "stage2", "circle", "big_world_500", "stage1_fidelity"])
def test_fp64_oracle_one_tick_from_identical_state(hip, name):
    """North-star tolerance: per-step pose/scan/reward agreement with the float64 NumPy
    re-implementation to 1e-5 (tolerances written out below) -- on the Stage-1 rink, the Stage-2 map (group episodes), the
    circle world (go-to-goal actions would be the same ticks: random ones exercise more), one 500-robot world (the
    per-robot-thread path) and the Stage-1 rink in fidelity mode (0.2 m raster: ranges there are quantised to 0.2 m cells,
    so the grazing-beam bound is 1.5 x 0.2 m)."""
    make, ticks = _fp64_scenarios()[name]
    sc = make()
    env = hip.VecStageWorld(sc)
    env.reset()
    rng = np.random.default_rng(5)
    bad_beams, n_beams, flag_mismatch, n_rob = 0, 0, 0, 0
    worst_scan = 0.0
    for k in range(ticks):
        torch.cuda.synchronize()
        o64 = U.oracle_env(sc, np.float64)
        for f in ("pose", "speed", "speed_gt", "goal", "init_pose", "prev_dist", "reward", "scan", "obs"):
            setattr(o64, f, getattr(env, f).cpu().numpy().astype(np.float64))
        for f in ("t", "episode", "crashed", "live", "done", "result", "first_result"):
            setattr(o64, f, getattr(env, f).cpu().numpy().astype(getattr(o64, f).dtype))
        a = U.random_actions(rng, sc.num_robots)
        env.step(torch.from_numpy(a).cuda())
        o64.step(a.astype(np.float64))
        torch.cuda.synchronize()
        h = U.HostView(env)
        same = (h.done == o64.done) & (h.crashed == o64.crashed) & (h.result == o64.result)
        flag_mismatch += int((~same).sum())
        n_rob += sc.num_robots
        ok = same & (h.done == 0)  # robots that were reset drew new poses; compare the others
        assert np.abs(h.pose[ok] - o64.pose[ok]).max() <= 1e-5
        assert np.abs(h.reward[ok] - o64.reward[ok]).max() <= 1e-5 * 16  # |reward| up to 15: 1e-5 relative
        assert np.abs(h.local_goal[ok] - o64.local_goal[ok]).max() <= 2e-5
        ds = np.abs(h.scan[ok] - o64.scan[ok])
        bad_beams += int((ds > 1e-5).sum())
        n_beams += ds.size
        worst_scan = max(worst_scan, float(ds.max()))
    print(f"fp64 check: beams off by >1e-5: {bad_beams}/{n_beams} = {bad_beams / n_beams:.2e}; worst {worst_scan:.2e}; "
          f"flag mismatches {flag_mismatch}/{n_rob}")
    cell = max(sc.grid.cell, getattr(sc, "collision_raster", 0.0))
    # what is guaranteed: >= 99.9 % of beams within 1e-5 (fidelity mode: 99.5 % -- robots are seen through 0.2 m raster cells
    # too, so more beams sit near a cell corner) ...
    # (the Stage-2 map: 99.8 % -- obstacles everywhere, measured 1.4e-3 of the beams graze a cell corner)
    assert bad_beams / n_beams <= {"stage1_fidelity": 5e-3, "stage2": 2e-3}.get(name, 1e-3)
    assert worst_scan <= 1.5 * cell               # ... and the grazing ones within one cell's extent along the ray
    assert flag_mismatch / n_rob <= 0.005
    env.close()


@pytest.mark.parametrize("chains,graph", [(1, True), (2, True), (4, True), (2, False), (3, False), (1, "native"), (2, "native"),
                                          (3, "native"), (5, "native"), (2, "chained"), (3, "chained")])
@pytest.mark.parametrize("name", ["stage1", "stage2", "stage1_fidelity"])
def test_bench_schedule_leaves_the_world_where_the_c_oracle_leaves_it(hip, name, chains, graph):
    """The execution modes bench.py TIMES -- one mrca_step_many call ("native": the library enqueues every launch, world ranges
    on streams of their own), ticks replayed as hipGraphs off the 16-deep action pool, or launched tick by tick; optionally as
    world ranges half a tick apart (bench.TickSchedule: mrca_move_worlds / mrca_observe_worlds) -- against
    the C oracle stepping the same action sequence tick by tick: 37 ticks (two full graphs and a remainder), every field of
    every robot."""
    import bench
    sc = {"stage1": lambda: S.stage1(num_worlds=6, robots_per_world=16, seed=31),
          "stage2": lambda: S.stage2(num_worlds=3, seed=31),
          "stage1_fidelity": lambda: S.stage1(num_worlds=5, robots_per_world=16, seed=31, stage_resolution=True)}[name]()
    env = hip.VecStageWorld(sc)
    ora = U.COracleEnv(sc)
    pool = bench.action_pool(sc.num_robots, env.device, 5)
    host_pool = [a.cpu().numpy() for a in pool]
    sched = bench.TickSchedule(env, pool, chains=chains, graph=graph is True, native=graph in ("native", "chained"),
                               chained=graph == "chained")
    sched.ticks_per_graph = 16
    assert sched.chains == min(chains, sc.num_worlds) and sum(c for _f, c in sched.ranges) == sc.num_worlds
    env.reset()
    ora.reset()
    for k in range(3):
        env.step(pool[k])
        ora.step(host_pool[k])
    torch.cuda.synchronize()
    sched.capture(0, 37)
    torch.cuda.synchronize()
    U.assert_state_equal(U.HostView(env), ora, what=f"{name}: a capture must not move the world")
    sched.run(0, 37)
    for k in range(37):
        ora.step(host_pool[k % len(host_pool)])
    torch.cuda.synchronize()
    U.assert_state_equal(U.HostView(env), ora, what=f"{name} after 37 scheduled ticks, chains={chains} graph={graph}")
    U.assert_hits_equal(env, ora, what=f"{name} after 37 scheduled ticks")
    env.check()
    env.close()


@pytest.mark.parametrize("chains", [1, 2, 3])
def test_step_many_of_every_short_length(hip, chains):
    """The run-ahead pass enqueues its ticks in blocks -- [0], [1], [2], [3], then fours -- with the move launches three blocks
    ahead of the ray casts (csrc/mrca_abi.hip run_ahead_pass): calls of 1, 2, ... 11 ticks one after the other (every shape of a
    pass's head and tail, blocks that do not exist included) leave the world where the C oracle's tick-by-tick run leaves it."""
    import bench
    sc = S.stage1(num_worlds=6, robots_per_world=16, seed=77)
    env = hip.VecStageWorld(sc)
    ora = U.COracleEnv(sc)
    pool = bench.action_pool(sc.num_robots, env.device, 9, depth=80)
    host_pool = [a.cpu().numpy() for a in pool]
    env.reset()
    ora.reset()
    k = 0
    for K in range(1, 12):
        env.step_many(pool, k, K, chains)
        for j in range(K):
            ora.step(host_pool[(k + j) % len(host_pool)])
        k += K
        torch.cuda.synchronize()
        U.assert_state_equal(U.HostView(env), ora, what=f"after a call of {K} ticks ({k} in all), chains={chains}")
    U.assert_hits_equal(env, ora, what="after calls of 1 .. 11 ticks")
    env.check()
    env.close()


@pytest.mark.parametrize("chains", [1, 2])
def test_step_many_inside_a_graph_capture(hip, chains):
    """mrca_step_many is "capturable like any other call" (include/mrca_env.h): eight ticks of two world ranges captured into
    one hipGraph from the caller's stream (the env's own streams join the capture through the stagger event and leave it through
    the join), replayed three times -- the world must stand where the C oracle stands after the same 24 ticks."""
    import bench
    sc = S.stage1(num_worlds=6, robots_per_world=12, seed=17)
    env = hip.VecStageWorld(sc)
    ora = U.COracleEnv(sc)
    pool = bench.action_pool(sc.num_robots, env.device, 9, depth=8)
    host_pool = [a.cpu().numpy() for a in pool]
    env.reset()
    ora.reset()
    env.step_many(pool, 0, 8, chains)          # (creates the env's streams and events outside the capture)
    for k in range(8):
        ora.step(host_pool[k])
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        env.step_many(pool, 0, 8, chains)
    torch.cuda.synchronize()
    U.assert_state_equal(U.HostView(env), ora, what="a capture must not move the world")
    for rep in range(3):
        g.replay()
        for k in range(8):
            ora.step(host_pool[k])
    torch.cuda.synchronize()
    env.invalidate_views()
    U.assert_state_equal(U.HostView(env), ora, what=f"three replays of a captured mrca_step_many, chains={chains}")
    env.check()
    env.close()


@pytest.mark.parametrize("chains", [2, -2])
def test_step_many_first_call_inside_a_capture_and_passes_longer_than_the_ring(hip, chains):
    """Two corners of mrca_step_many: (a) the env's streams and events exist since mrca_create, so the FIRST call may sit inside
    a (global-mode) graph capture; (b) a call of more ticks than the run-ahead ring has slots (300 > 256) is cut into passes by
    the library itself.  Run-ahead schedule (chains > 0) and round 5's chained one (chains < 0), against the C oracle."""
    import bench
    sc = S.stage2(num_worlds=3, seed=23)
    env = hip.VecStageWorld(sc)
    ora = U.COracleEnv(sc)
    pool = bench.action_pool(sc.num_robots, env.device, 11, depth=300)
    host_pool = [a.cpu().numpy() for a in pool]
    env.reset()
    ora.reset()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        env.step_many(pool, 0, 5, chains)
    g.replay()
    for k in range(5):
        ora.step(host_pool[k])
    torch.cuda.synchronize()
    env.invalidate_views()
    U.assert_state_equal(U.HostView(env), ora, what=f"first call captured, chains={chains}")
    env.step_many(pool, 0, 300, chains)
    for k in range(300):
        ora.step(host_pool[k])
    torch.cuda.synchronize()
    env.invalidate_views()
    U.assert_state_equal(U.HostView(env), ora, what=f"300 ticks from one call, chains={chains}")
    U.assert_hits_equal(env, ora, what="300 ticks from one call")
    env.check()
    env.close()


def test_step_many_from_alternating_caller_streams(hip):
    """mrca_step_many checks its streams against the caller's stream at the first call on it (hardware-queue sharing,
    DESIGN.md 5.10); a caller that alternates between the NULL stream and two streams of its own must get the same world
    whichever stream a call came in on -- each call ordered behind the one before by the caller's own events."""
    import bench
    sc = S.stage1(num_worlds=6, robots_per_world=16, seed=41)
    env = hip.VecStageWorld(sc)
    ora = U.COracleEnv(sc)
    pool = bench.action_pool(sc.num_robots, env.device, 13, depth=48)
    host_pool = [a.cpu().numpy() for a in pool]
    env.reset()
    ora.reset()
    torch.cuda.synchronize()
    streams = [torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()]
    k = 0
    for rep in range(6):
        s = streams[rep % 3]
        s.wait_stream(streams[(rep - 1) % 3])
        with torch.cuda.stream(s):
            env.step_many(pool, k, 8, 2)
        for j in range(8):
            ora.step(host_pool[k + j])
        k += 8
    torch.cuda.synchronize()
    env.invalidate_views()
    U.assert_state_equal(U.HostView(env), ora, what="48 ticks from three alternating caller streams")
    env.check()
    env.close()


def test_world_range_calls_leave_the_other_worlds_alone(hip):
    """mrca_step_worlds / mrca_move_worlds / mrca_observe_worlds: only the worlds of the range tick; a world stepped alone
    ends where the same world of a fully stepped env ends."""
    sc = S.stage1(num_worlds=5, robots_per_world=12, seed=8)
    a_env, b_env = hip.VecStageWorld(sc), hip.VecStageWorld(sc)
    a_env.reset()
    b_env.reset()
    R = sc.robots_per_world
    g = torch.Generator(device="cpu").manual_seed(3)
    for k in range(20):
        a = torch.stack([torch.rand(sc.num_robots, generator=g), torch.rand(sc.num_robots, generator=g) * 2 - 1], 1).float().cuda()
        a_env.step(a)
        before = U.HostView(b_env)
        if k % 2:
            b_env.step(a, worlds=(1, 3))
        else:
            b_env.move(a, (1, 3))
            b_env.observe((1, 3))
        torch.cuda.synchronize()
        after = U.HostView(b_env)
        for f in U.STATE_FIELDS:        # worlds 0 and 4 untouched
            for lo, hi in ((0, R), (4 * R, 5 * R)):
                assert np.array_equal(getattr(before, f)[lo:hi], getattr(after, f)[lo:hi]), f
        U.assert_state_equal(U.HostView(b_env, R, 4 * R), U.HostView(a_env, R, 4 * R), what=f"worlds 1..3 step {k}")
    with pytest.raises(RuntimeError):
        b_env.step(a, worlds=(3, 3))
    a_env.close()
    b_env.close()


def test_eager_views_env_matches_the_lazy_one(hip):
    """lazy_obs = 0 (what a plain C caller of the reference-shaped getters uses): MRCA_F_SCAN / MRCA_F_OBS are formed inside
    every step and reset -- equal to the views the default env forms on demand."""
    sc = S.stage1(num_worlds=3, robots_per_world=8, seed=2)
    lazy, eager = hip.VecStageWorld(sc), hip.VecStageWorld(sc, lazy_obs=False)
    lazy.reset()
    eager.reset()
    rng = np.random.default_rng(1)
    for k in range(12):
        a = torch.from_numpy(U.random_actions(rng, sc.num_robots)).cuda()
        lazy.step(a)
        eager.step(a)
        torch.cuda.synchronize()
        assert torch.equal(eager._obs, lazy.obs) and torch.equal(eager._scan, lazy.scan)   # _obs: the raw field, no materialise call
    # ... and through mrca_step_many (two world ranges, the run-ahead pass): the ray casts form the views of their own robots
    import bench
    pool = bench.action_pool(sc.num_robots, lazy.device, 3, depth=40)
    for first, count in ((0, 1), (1, 7), (8, 30)):
        lazy.step_many(pool, first, count, 2)
        eager.step_many(pool, first, count, 2)
        torch.cuda.synchronize()
        assert torch.equal(eager._obs, lazy.obs) and torch.equal(eager._scan, lazy.scan), (first, count)
    lazy.close()
    eager.close()


def _properties(env, prev_obs, prev_fresh_next=None):
    sc = env.scenario
    scan, obs = env.scan, env.obs
    assert float(scan.min()) >= 0.0 and float(scan.max()) <= 6.0
    assert torch.equal(obs[:, -1].cpu(), scan.cpu() / 6.0 - 0.5)   # IEEE division on the host
    fresh = env.fresh.bool()
    if prev_obs is not None:
        keep = ~fresh
        assert torch.equal(obs[keep][:, :-1], prev_obs[keep][:, 1:])       # frame stack shifted by one
    fr = obs[fresh]
    if fr.numel():
        assert torch.equal(fr[:, 0], fr[:, -1]) and torch.equal(fr[:, 1], fr[:, -1])  # deque([obs]*3)
    th = env.pose[:, 2]
    assert float(th.max()) <= np.float32(np.pi) and float(th.min()) > -np.float32(np.pi)
    assert bool(((env.result == 0) == (env.done == 0))[env.live.bool() & ~fresh].all())


@pytest.mark.parametrize("name,worlds,R", [("stage1", 128, 32), ("stage2", 187, 44)])
def test_full_size_slice_exact_and_properties(hip, name, worlds, R):
    """BASELINE configs[1] (4096 robots, Stage-1 rink) and configs[2] (8192+ robots, Stage-2 map):
    first and last world checked bit-exactly against the oracle, whole batch through properties,
    and a second env with the same seed must be bit-identical (determinism)."""
    sc = S.stage1(num_worlds=worlds, robots_per_world=R, seed=77) if name == "stage1" else S.stage2(num_worlds=worlds, seed=77)
    env = hip.VecStageWorld(sc)
    env2 = hip.VecStageWorld(sc)
    first = U.oracle_env(sc, np.float32, first_world=0, num_worlds=1)
    last = U.oracle_env(sc, np.float32, first_world=worlds - 1, num_worlds=1)
    env.reset()
    env2.reset()
    first.reset()
    last.reset()
    N = sc.num_robots
    g = torch.Generator(device="cpu").manual_seed(1)
    prev = None
    for k in range(12):
        a = torch.stack([torch.rand(N, generator=g), torch.rand(N, generator=g) * 2 - 1], 1).float()
        ad = a.cuda()
        env.step(ad)
        env2.step(ad)
        first.step(a[:R].numpy())
        last.step(a[N - R:].numpy())
        torch.cuda.synchronize()
        U.assert_state_equal(U.HostView(env, 0, R), first, what=f"{name} first world step {k}")
        U.assert_state_equal(U.HostView(env, N - R, N), last, what=f"{name} last world step {k}")
        _properties(env, prev)
        prev = env.obs.clone()
    for f in U.STATE_FIELDS:
        assert torch.equal(getattr(env, f), getattr(env2, f)), f
    env.close()
    env2.close()


@pytest.mark.parametrize("name", ["stage1", "stage2"])
def test_fidelity_mode_bit_exact(hip, name):
    """Fidelity mode: the maps at the reference's OWN Stage resolution (0.2 m, worlds/stage1.world:3) and robots
    colliding when their outlines share a 0.2 m raster cell -- bit-exact against the C oracle, which must also see
    more crashes than with exact rectangles."""
    sc = S.stage1(num_worlds=6, robots_per_world=24, seed=21, stage_resolution=True) if name == "stage1" else \
        S.stage2(num_worlds=2, seed=21, stage_resolution=True)
    assert sc.grid.cell == 0.2 and sc.collision_raster == 0.2
    env = hip.VecStageWorld(sc)
    ora = U.COracleEnv(sc)
    env.reset()
    ora.reset()
    rng = np.random.default_rng(4)
    crashes = 0
    for k in range(80):
        a = U.random_actions(rng, sc.num_robots)
        env.step(torch.from_numpy(a).cuda())
        ora.step(a)
        crashes += int(((ora.result == 2) & (ora.done == 1)).sum())
        if k % 4 == 3:
            torch.cuda.synchronize()
            U.assert_state_equal(U.HostView(env), ora, what=f"fidelity {name} step {k}")
    assert crashes > 5
    env.close()


def test_circle_world_at_stage_resolution_bit_exact(hip):
    """circle.world:3 resolution 0.01 m: a 6000 x 6000 cell map (144 MB free-rectangle field: far beyond L2).  Twenty
    ticks of the go-to-goal controller, bit-exact against the C oracle's plain cell walk; the launch times go to the
    log (what the fine raster costs)."""
    sc = S.circle(num_worlds=2, stage_resolution=True)
    assert sc.grid.cell == 0.01 and sc.grid.width == 6000
    env = hip.VecStageWorld(sc)
    ora = U.COracleEnv(sc)
    env.reset()
    ora.reset()
    torch.cuda.synchronize()
    U.assert_state_equal(U.HostView(env), ora, what="circle r0.01 reset")
    env.enable_timing(1)
    for k in range(20):
        lg = ora.local_goal
        bearing = np.arctan2(lg[:, 1], lg[:, 0])
        a = np.stack([np.full(sc.num_robots, 1.0), np.clip(2.0 * bearing, -1, 1)], 1).astype(np.float32)
        env.step(torch.from_numpy(a).cuda())
        ora.step(a)
        if k % 5 == 4:
            torch.cuda.synchronize()
            U.assert_state_equal(U.HostView(env), ora, what=f"circle r0.01 step {k}")
    mv, ry, n = env.read_timing()
    print(f"circle world at 0.01 m cells, 100 robots: move {mv / n * 1e3:.1f} us, ray cast {ry / n * 1e3:.1f} us per tick")
    env.close()


def test_gae_kernel(hip):
    """mrca_gae vs generate_train_data (model/ppo.py:122-139) restated in the oracle."""
    rng = np.random.default_rng(0)
    for T, N in ((128, 24), (128, 4096), (5, 1), (33, 1000)):
        r = rng.normal(size=(T, N)).astype(np.float32)
        v = rng.normal(size=(T, N)).astype(np.float32)
        lv = rng.normal(size=N).astype(np.float32)
        d = (rng.uniform(size=(T, N)) < 0.05).astype(np.uint8)
        tg, adv = hip.gae(torch.from_numpy(r).cuda(), torch.from_numpy(v).cuda(), torch.from_numpy(lv).cuda(),
                          torch.from_numpy(d).cuda(), 0.99, 0.95)
        t32, a32 = O.gae(r, v, lv, d, 0.99, 0.95, np.float32)
        assert (tg.cpu().numpy().view(np.uint32) == t32.view(np.uint32)).all()
        assert (adv.cpu().numpy().view(np.uint32) == a32.view(np.uint32)).all()
        t64, a64 = O.gae(r, v, lv, d, 0.99, 0.95, np.float64)
        assert np.abs(tg.cpu().numpy() - t64).max() <= 1e-4 and np.abs(adv.cpu().numpy() - a64).max() <= 1e-4


def test_error_behaviour(hip):
    env = hip.VecStageWorld(S.stage1(num_worlds=1, robots_per_world=4))
    with pytest.raises(ValueError):
        env.step(torch.zeros(3, 2, device="cuda"))
    env.close()


@pytest.mark.parametrize("name,worlds,R,steps", [("stage1", 128, 32, 24), ("stage2", 187, 44, 12), ("stage1_fidelity", 128, 32, 16),
                                                 ("stage2_fidelity", 187, 44, 8)])
def test_full_batch_bit_exact_vs_c_oracle(hip, name, worlds, R, steps):
    """BASELINE configs[1]/[2] at FULL size, every robot, every field, bit-for-bit against the plain-C
    restatement of the oracle (itself bit-identical to the NumPy oracle: tests/test_oracle_c.py) -- in the default mode and
    in fidelity mode (Stage's 0.2 m raster: outline bitmaps, the closed-form raster lidar)."""
    fid = name.endswith("_fidelity")
    sc = S.stage1(num_worlds=worlds, robots_per_world=R, seed=123, stage_resolution=fid) if name.startswith("stage1") else \
        S.stage2(num_worlds=worlds, seed=123, stage_resolution=fid)
    env = hip.VecStageWorld(sc)
    ora = U.COracleEnv(sc)
    env.reset()
    ora.reset()
    torch.cuda.synchronize()
    U.assert_state_equal(U.HostView(env), ora, what=f"{name} full reset")
    g = torch.Generator(device="cpu").manual_seed(2)
    N = sc.num_robots
    for k in range(steps):
        a = torch.stack([torch.rand(N, generator=g), torch.rand(N, generator=g) * 2 - 1], 1).float()
        env.step(a.cuda())
        ora.step(a.numpy())
        if k % 4 == 3 or k == steps - 1:
            torch.cuda.synchronize()
            U.assert_state_equal(U.HostView(env), ora, what=f"{name} full step {k}")
    env.close()


@pytest.mark.parametrize("name,worlds,R,ticks", [("stage1", 128, 32, 400), ("stage2", 187, 44, 400),
                                                 ("stage1_fidelity", 128, 32, 200)])
def test_soak_at_the_benchmarks_own_shape(hip, name, worlds, R, ticks):
    """The benchmark's OWN shape, long: BASELINE configs[1] / configs[2] at full size through the schedule bench.py times
    (one mrca_step_many call per 50 ticks, two world ranges half a tick apart on two streams, actions drawn i.i.d. per tick)
    for hundreds of ticks against the C oracle stepping the same actions tick by tick -- every field of every robot compared
    every 50 ticks.  Long enough that every Stage-1 robot goes through at least two whole episodes (crash or the t > 150
    time-out, then the in-kernel restart with fresh Philox draws) and every Stage-2 group restarts several times."""
    import bench
    fid = name.endswith("_fidelity")
    sc = S.stage1(num_worlds=worlds, robots_per_world=R, seed=77, stage_resolution=fid) if name.startswith("stage1") else \
        S.stage2(num_worlds=worlds, seed=77, stage_resolution=fid)
    env = hip.VecStageWorld(sc)
    ora = U.COracleEnv(sc)
    pool = bench.action_pool(sc.num_robots, env.device, 7, depth=ticks)
    host_pool = [a.cpu().numpy() for a in pool]
    sched = bench.TickSchedule(env, pool, chains=2, native=True)
    env.reset()
    ora.reset()
    torch.cuda.synchronize()
    U.assert_state_equal(U.HostView(env), ora, what=f"{name} soak reset")
    timeouts = 0
    for first in range(0, ticks, 50):
        sched.run(first, 50)
        for k in range(first, first + 50):
            ora.step(host_pool[k])
            timeouts += int((np.asarray(ora.result) == 3).sum())
        torch.cuda.synchronize()
        U.assert_state_equal(U.HostView(env), ora, what=f"{name} soak after {first + 50} scheduled ticks")
    U.assert_hits_equal(env, ora, what=f"{name} soak, final tick")
    env.check()
    ep = np.asarray(ora.episode)
    if name.startswith("stage1"):
        assert ep.min() >= (2 if ticks >= 400 else 1), ep.min()
    else:
        assert ep.min() >= 2, ep.min()               # every group restarted at least twice
    assert timeouts > 0                               # and the time-out path ran
    env.close()


@pytest.mark.parametrize("knob,label", [(256, "1 beam per thread"), (512, "2 beams per thread, one after the other"),
                                        (512 + 4096, "2 beams per thread in lock step"),
                                        (768, "4 beams per thread, one after the other (2 waves per workgroup)"),
                                        (768 + 4096, "4 beams per thread in lock step"),
                                        (256 + 2048, "1 beam per thread, dedicated preparation wave"),
                                        (512 + 2048, "2 beams sequential, dedicated preparation wave"),
                                        (512 + 4096 + 2048, "2 beams lock step, dedicated preparation wave"),
                                        (768 + 4096 + 2048, "4 beams lock step, dedicated preparation wave"),
                                        (64, "phase stamps drain the memory queues")])
def test_raycast_launch_shapes_bit_exact(hip, knob, label):
    """The launch-shape knobs of the PROFILING build (beams per marching thread, with / without the dedicated
    preparation wave) only change how the work is dealt to threads: every shape must match the oracle bit-for-bit."""
    from mrca import _lib
    for sc in (S.stage1(num_worlds=4, robots_per_world=16, seed=5), S.stage2(num_worlds=1, seed=5)):
        env = hip.VecStageWorld(sc, lib_path=_lib.PROFILING_LIB_PATH)
        env.set_debug_flags(knob)
        ora = U.COracleEnv(sc)
        env.reset()
        ora.reset()
        rng = np.random.default_rng(9)
        for k in range(40):
            a = U.random_actions(rng, sc.num_robots)
            env.step(torch.from_numpy(a).cuda())
            ora.step(a)
            if k % 8 == 7:
                torch.cuda.synchronize()
                U.assert_state_equal(U.HostView(env), ora, what=f"{label} {sc.name} step {k}")
        env.close()


def test_product_library_has_no_ablation_switches(hip):
    env = hip.VecStageWorld(S.stage1(num_worlds=1, robots_per_world=4))
    with pytest.raises(RuntimeError, match="profiling build"):
        env.set_debug_flags(1)
    env.close()
