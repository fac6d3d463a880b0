#!/usr/bin/env python3
"""What a timed region of K ticks costs beyond K x the steady tick: bench.py's schedule (mrca_step_many, world ranges half a
tick apart) timed for K = 5 .. 320 ticks between two device synchronisations, five times each (median), per number of ranges;
a straight line through the points gives the fixed cost of a region (the intercept: first launch out of an idle queue, the
stagger, the drain, the synchronisation) and the steady tick (the slope).  The driver's bench run is K = 20.

    python tools/region_sweep.py [--schedule native|chained|graph]     (chained: round 5's schedule inside mrca_step_many)
"""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-collision-avoidance_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as G  # noqa: E402

G.build()
import bench  # noqa: E402
from mrca import scenario as S  # noqa: E402
from mrca.vec_env import VecStageWorld  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--schedule", default="native")
a = ap.parse_args()
sc = S.stage1(num_worlds=128, robots_per_world=32, seed=1000)
env = VecStageWorld(sc)
pool = bench.action_pool(sc.num_robots, env.device, 1, depth=320)
env.reset()
for k in range(3):
    env.step(pool[k])
torch.cuda.synchronize()
for chains in (1, 2, 3):
    sched = bench.TickSchedule(env, pool, chains=chains, graph=a.schedule == "graph", native=a.schedule in ("native", "chained"),
                               chained=a.schedule == "chained")
    pts = []
    for K in (5, 10, 20, 40, 80, 160, 320):
        sched.capture(0, K)
        sched.prime(0, K)
        sched.run(0, K)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sched.run(0, K)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e6)
        pts.append((K, statistics.median(ts), min(ts)))
    n = len(pts)
    mx = sum(p[0] for p in pts) / n
    my = sum(p[1] for p in pts) / n
    slope = sum((p[0] - mx) * (p[1] - my) for p in pts) / sum((p[0] - mx) ** 2 for p in pts)
    icpt = my - slope * mx
    print(f"{a.schedule} schedule, {chains} range(s): region = {icpt:6.1f} us + {slope:6.2f} us per tick   " +
          "  ".join(f"K={k}: {m:7.1f} us ({sc.num_robots * k / m:6.1f} M, best {sc.num_robots * k / b:6.1f})" for k, m, b in pts))
env.close()
