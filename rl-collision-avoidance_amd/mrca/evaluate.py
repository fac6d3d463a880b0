"""Circle-test evaluator: the reference's ``circle_test.py`` demo loop (``enjoy``, :36-83) for any
number of independent 50-robot circles, with the success metric the reference never computes.

Success rate (decision 12): fraction of robots whose FIRST terminal event since reset is
"Reach Goal" -- latched on the device in ``first_result`` (the reference does not latch,
circle_test.py:64-70).  Also reported: crash rate, unfinished rate, mean ticks-to-goal of the
successful robots and their mean path length ratio vs the 50 m straight line.
"""
import argparse
import json
import math

import torch

from . import ppo, scenario
from .net import CNNPolicy

ACTION_BOUND = ((0.0, -1.0), (1.0, 1.0))     # circle_test.py:92


def go_to_goal_policy(obs, local_goal, speed):
    """Stand-in controller (no learning): drive at the goal, turn towards it, slow down when the
    lidar sees something close ahead.  Used where ``policy/stage2.pth`` is unavailable."""
    lx, ly = local_goal[:, 0], local_goal[:, 1]
    bearing = torch.atan2(ly, lx)
    ahead = (obs[:, -1, 192:320].min(dim=1).values + 0.5) * 6.0      # nearest return in the front 45 degrees
    v = torch.clamp(0.4 * ahead, 0.0, 1.0) * (bearing.abs() < 1.0).float()
    w = torch.clamp(2.0 * bearing, -1.0, 1.0)
    return torch.stack([v, w], dim=1)


def staggered_roundabout_policy(num_robots, bias=1.4, near=3.5, vgain=0.8, stop=0.5, slowest=0.5):
    """A stand-in with NON-TRIVIAL outcomes on the circle test (78 % success / 22 % crashes for one circle):
    head for the goal, veer right in proportion to how close the nearest return in the front sector is (a
    roundabout), slow down before obstacles, and -- to break the perfect symmetry of the scenario, which no
    memoryless symmetric rule survives -- give every robot its own speed limit in [slowest, 1] m/s."""
    idx = torch.arange(num_robots) % 50
    scale = slowest + (1.0 - slowest) * ((idx * 7) % 50).float() / 49.0

    def fn(obs, local_goal, speed):
        lx, ly = local_goal[:, 0], local_goal[:, 1]
        bearing = torch.atan2(ly, lx)
        dist = torch.sqrt(lx * lx + ly * ly)
        front = ((obs[:, -1, 176:336] + 0.5) * 6.0).min(dim=1).values
        prox = ((near - front) / near).clamp(0.0, 1.0)
        target = bearing - bias * prox * (dist > 1.0).float()
        w = torch.clamp(2.5 * target, -1.0, 1.0)
        v = torch.clamp(vgain * (front - stop), 0.0, 1.0) * (target.abs() < 1.0).float() * scale.to(obs.device)
        return torch.stack([v, w], dim=1)
    return fn


def cnn_policy_fn(policy, fused=False, env=None):
    """``fused`` with ``env``: the policy's HIP front end reads the env's frame ring in place (no materialised stacks)."""
    def fn(obs, local_goal, speed):
        head = None
        if fused and env is not None:
            obs, head = ppo.policy_input(env, True)
        _mean, scaled = ppo.generate_action_no_sampling(policy, obs, local_goal, speed, ACTION_BOUND, fused=fused,
                                                        obs_head=head)
        return scaled
    fn.wants_obs = not (fused and env is not None)
    return fn


def perturbed_start(env, jitter_xy, jitter_th, seed):
    """Start poses of a circle world with every robot moved off its table pose by U(-jitter_xy, jitter_xy) metres in x
    and y and U(-jitter_th, jitter_th) radians of heading -- drawn from a Philox generator seeded with ``seed``, so that
    W circles are W DIFFERENT scenarios (the reference's table, model/utils.py:6-38, is one perfectly symmetric
    scenario: a deterministic policy either solves all of its copies or none).  Goals stay the table's.
    -> (poses f32[N,3], goals f32[N,2]) on the env's device."""
    sc = env.scenario
    dev = env.local_goal.device
    W, R = sc.num_worlds, sc.robots_per_world
    base = torch.as_tensor(sc.init_table, dtype=torch.float32, device=dev).unsqueeze(0).expand(W, R, 3)
    goal = torch.as_tensor(sc.goal_table, dtype=torch.float32, device=dev).unsqueeze(0).expand(W, R, 2)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    u = torch.rand(W, R, 3, generator=g, device=dev) * 2.0 - 1.0
    scale = torch.tensor([jitter_xy, jitter_xy, jitter_th], dtype=torch.float32, device=dev)
    poses = base + u * scale
    poses[..., 2] = torch.atan2(torch.sin(poses[..., 2]), torch.cos(poses[..., 2]))
    return poses.reshape(W * R, 3).contiguous(), goal.reshape(W * R, 2).contiguous()


def circle_test(env, policy_fn, max_ticks=1200, perturb=None, seed=0):
    """Runs the circle scenario on ``env`` (any object with the VecStageWorld surface) and returns
    the metrics dict.  Mirrors circle_test.py:52-80: deterministic action, and a robot whose last
    ``get_reward_and_terminate`` said terminal gets v = 0 (``real_action[0] = 0``, :64-65).
    ``perturb = (jitter_xy, jitter_th)``: every circle starts from its own jittered poses (``perturbed_start``); the
    metrics then carry the mean success rate over circles with a 95 % interval."""
    if perturb is None:
        env.reset()
    else:
        poses, goals = perturbed_start(env, perturb[0], perturb[1], seed)
        env.reset(None, poses, goals)
    N = env.N
    dev = env.local_goal.device
    ticks_to_goal = torch.zeros(N, device=dev)
    path = torch.zeros(N, device=dev)
    last_terminal = torch.zeros(N, dtype=torch.bool, device=dev)
    for k in range(max_ticks):
        a = policy_fn(env.obs if getattr(policy_fn, "wants_obs", True) else None, env.local_goal, env.speed).float().clone()
        a[:, 0] = torch.where(last_terminal, torch.zeros_like(a[:, 0]), a[:, 0])
        pending = env.first_result == 0
        env.step(a.contiguous())
        last_terminal = env.done.bool().clone()
        path += torch.where(pending, env.speed_gt[:, 0] * 0.1, torch.zeros_like(path))
        ticks_to_goal += pending.float()
        if k % 25 == 24 and bool((env.first_result != 0).all()):    # one host round trip every 25 ticks
            break
    if hasattr(env, "check"):
        env.check()          # mrca_check: raises if a device-side pass flagged an untrusted state (worlds with > 64 robots)
    fr = env.first_result
    reach = fr == 1
    n_reach = int(reach.sum())
    # the paper's metrics (Long et al. 2018, Sec. V-B) over the robots that reached their goal; the straight
    # line is |goal - init| - goal radius (0.5 m) at the speed limit of 1 m/s, one tick = 0.1 s
    straight = ((env.goal - env.init_pose[:, :2]).norm(dim=1) - 0.5).clamp(min=0.0)
    time_s = ticks_to_goal * 0.1
    extra_time = (time_s - straight / 1.0)[reach]
    extra_dist = (path - straight)[reach]
    avg_speed = (path / time_s.clamp(min=0.1))[reach]
    R = getattr(getattr(env, "scenario", None), "robots_per_world", None)
    per_circle = {}
    if R and N % R == 0 and N // R > 1:
        # the circle is the unit of randomisation: mean of the per-circle success fractions, normal 95 % interval of
        # that mean (robots of one circle succeed or fail together far too often to count as independent samples)
        sr = reach.view(N // R, R).float().mean(dim=1)
        sd = float(sr.std(unbiased=True))
        half = 1.96 * sd / math.sqrt(N // R)
        per_circle = {"circles": N // R, "success_rate_ci95": [max(0.0, float(sr.mean()) - half), min(1.0, float(sr.mean()) + half)],
                      "circles_fully_solved": float((sr == 1.0).float().mean()), "worst_circle": float(sr.min())}
    return {
        **per_circle,
        "robots": N, "ticks_run": k + 1,
        "success_rate": n_reach / N,
        "crash_rate": float((fr == 2).float().mean()),
        "timeout_rate": float((fr == 3).float().mean()),
        "unfinished_rate": float((fr == 0).float().mean()),
        "mean_ticks_to_goal": float(ticks_to_goal[reach].mean()) if n_reach else None,
        "mean_path_ratio": float((path[reach] / straight[reach].clamp(min=1e-6)).mean()) if n_reach else None,
        "extra_time_s": float(extra_time.mean()) if n_reach else None,
        "extra_distance_m": float(extra_dist.mean()) if n_reach else None,
        "average_speed_mps": float(avg_speed.mean()) if n_reach else None,
    }


def main():
    ap = argparse.ArgumentParser(description="circle test (circle_test.py) at scale on the MI355X env")
    ap.add_argument("--circles", type=int, default=1, help="independent 50-robot circles (50 -> 50 000 robots: 1000)")
    ap.add_argument("--policy", default=None, help="state_dict file with the reference's keys (policy/stage2.pth)")
    ap.add_argument("--max-ticks", type=int, default=1200)   # circle_world.py:198 allows 10000
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--robots", type=int, default=50, help="robots per circle (50 = the reference's table; others: scenario.circle_n)")
    ap.add_argument("--radius", type=float, default=25.0)
    ap.add_argument("--perturb", default=None, help="'dxy,dth': start every robot U(-dxy, dxy) m / U(-dth, dth) rad off its table "
                                                     "pose (Philox, seeded by --seed) so that the circles are different scenarios, "
                                                     "e.g. 0.2,0.1; the output carries the mean success rate with a 95 %% interval")
    ap.add_argument("--stage-resolution", action="store_true", help="fidelity mode: the map at Stage's own cell size "
                                                                      "(worlds/circle.world:3) -- the clearances a Stage-trained policy lives with")
    ap.add_argument("--fused", action="store_true", help="policy inference through the fp32 HIP conv front end")
    a = ap.parse_args()
    from .vec_env import VecStageWorld
    if a.robots == 50 and a.radius == 25.0:
        sc = scenario.circle(num_worlds=a.circles, seed=a.seed, stage_resolution=a.stage_resolution)
    else:
        grid = scenario.load_map("circle_rink_r0010") if a.stage_resolution else None
        sc = scenario.circle_n(a.robots, a.radius, num_worlds=a.circles, seed=a.seed, grid=grid)
    env = VecStageWorld(sc)
    if a.policy:
        pol = CNNPolicy(3, 2).to(env.device)
        pol.load_state_dict(torch.load(a.policy, map_location=env.device))
        fn, name = cnn_policy_fn(pol, fused=a.fused, env=env), a.policy
    else:
        fn, name = staggered_roundabout_policy(env.N), "staggered-roundabout stand-in (no checkpoint given)"
    perturb = tuple(float(v) for v in a.perturb.split(",")) if a.perturb else None
    out = circle_test(env, fn, a.max_ticks, perturb=perturb, seed=a.seed)
    out["policy"] = name
    out["robots_per_circle"], out["radius_m"] = a.robots, a.radius
    out["perturb_xy_th"], out["seed"], out["stage_resolution"] = perturb, a.seed, bool(a.stage_resolution)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
