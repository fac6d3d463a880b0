#!/usr/bin/env python3
"""A few timed regions of K ticks through bench.py's schedule, nothing else -- to be run under `rocprofv3 --kernel-trace` so that
tools/trace_timeline.py can show ONE whole region launch by launch (which stream, when, how long, what overlapped).

    rocprofv3 --kernel-trace -d out -o t -- python tools/region_once.py [K] [chains] [native|chained]
    python tools/trace_timeline.py out <skip> <count>        # the last region = the last K x (1 + chains) dispatches
"""
import os
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

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 2
schedule = sys.argv[3] if len(sys.argv) > 3 else "native"
sc = S.stage1(num_worlds=128, robots_per_world=32, seed=1000)
env = VecStageWorld(sc)
pool = bench.action_pool(sc.num_robots, env.device, 1, depth=64)
env.reset()
sched = bench.TickSchedule(env, pool, chains=chains, native=True, chained=schedule == "chained")
for rep in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sched.run(0, K)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"region {rep}: {K} ticks, host enqueue {1e6 * (t1 - t0):7.1f} us, until synchronised {1e6 * (t2 - t0):7.1f} us "
          f"({sc.num_robots * K / (t2 - t0) / 1e6:6.1f} M)", flush=True)
env.close()
