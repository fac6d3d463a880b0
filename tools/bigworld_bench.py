#!/usr/bin/env python3
"""GPU: ONE circle of R robots (radius proportional to R, spacing 3.14 m) -- tick time of the big-world
path (per-robot threads, per-tick spatial hashes, dependency-round collision pass, chunked lidar neighbour lists) at
R = 500 ... 50 000, driven by a go-to-goal controller; and the jam: the same robots packed on a 0.8 m lattice.

    python tools/bigworld_bench.py [--shards K] [R ...]

``--shards K``: additionally times ONE rank's share of the same world sharded over K GPUs (mrca_step_slice with a slice
of R / K robots: the move phase replicated, the lidar and its outputs for the slice only) on this one GPU, and prints
the projected K-GPU speed-up = full tick / one rank's share (the per-tick all-gather of the commands, 8 B per robot, is
not in it)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-collision-avoidance_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as G  # noqa: E402

G.build()
from mrca import scenario as S  # noqa: E402
from mrca.vec_env import VecStageWorld  # noqa: E402


def controller(env):
    lg = env.local_goal
    bearing = torch.atan2(lg[:, 1], lg[:, 0])
    return torch.stack([torch.ones_like(bearing), torch.clamp(2.0 * bearing, -1, 1)], 1).contiguous()


argv = sys.argv[1:]
SHARDS = 0
if "--shards" in argv:
    i = argv.index("--shards")
    SHARDS = int(argv[i + 1])
    del argv[i: i + 2]


def timed(env, n, ray_slice=None):
    env.enable_timing(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        env.step(controller(env), ray_slice=ray_slice)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mv, ry, k = env.read_timing()
    env.enable_timing(False)
    return dt, mv / k * 1e3, ry / k * 1e3


for R in [int(x) for x in (argv or ["500", "5000", "50000"])]:
    env = VecStageWorld(S.circle_big(R))
    for label in ("circle", "jam"):
        if label == "circle":
            env.reset()
        else:
            side = int(np.ceil(np.sqrt(R)))
            rng = np.random.default_rng(1)
            ij = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2)[:R]
            xy = (ij - side / 2) * 0.8 + rng.uniform(-0.12, 0.12, (R, 2))
            poses = np.concatenate([xy, rng.uniform(-np.pi, np.pi, (R, 1))], 1).astype(np.float32)
            env.reset(torch.ones(R, dtype=torch.uint8, device="cuda"), torch.from_numpy(poses).cuda(), None)
        for _ in range(20):
            env.step(controller(env))
        torch.cuda.synchronize()
        n = 200
        dt, mv, ry = timed(env, n)
        out = {"robots": R, "layout": label, "ticks": n, "ms_per_tick_wall": dt / n * 1e3,
               "agent_steps_per_s": R * n / dt, "move_phase_us": mv, "raycast_us": ry,
               "crashed_now": int(env.crashed.sum())}
        if SHARDS > 1:
            per = -(-R // SHARDS)
            _dt, smv, sry = timed(env, 100, ray_slice=(0, per))
            out["one_rank_of_%d" % SHARDS] = {"slice_robots": per, "move_phase_us": smv, "raycast_us": sry,
                                              "projected_speedup": (mv + ry) / (smv + sry)}
        env.check()
        print(json.dumps(out), flush=True)
    env.close()
