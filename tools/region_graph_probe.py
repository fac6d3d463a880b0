#!/usr/bin/env python3
"""A region of K ticks as ONE mrca_step_many call against the SAME call captured once into a hipGraph and replayed: what the
host's share of a short region is (60 launches + 14 event waits for 20 ticks of two ranges; tools/region_once.py under
MRCA_HOST_TIMES measured 285 us of enqueueing against a region of 450).

    python tools/region_graph_probe.py [K] [chains]
"""
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

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sc = S.stage1(num_worlds=128, robots_per_world=32, seed=1000)
env = VecStageWorld(sc)
pool = bench.action_pool(sc.num_robots, env.device, 1, depth=64)
env.reset()
env.step_many(pool, 0, K, chains)          # warm: streams chosen and checked outside any capture
torch.cuda.synchronize()


def timed(fn, reps=9):
    ts, hs = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
        hs.append((t1 - t0) * 1e6)
    return statistics.median(ts), min(ts), statistics.median(hs)


med, best, host = timed(lambda: env.step_many(pool, 5, K, chains))
print(f"direct call : {K} ticks, {chains} ranges: region {med:7.1f} us (best {best:7.1f}), host {host:6.1f} us -> {sc.num_robots * K / med:6.1f} M")
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.graph(g, stream=side):
    env.step_many(pool, 5, K, chains)
g.replay()
torch.cuda.synchronize()
med, best, host = timed(g.replay)
print(f"graph replay: {K} ticks, {chains} ranges: region {med:7.1f} us (best {best:7.1f}), host {host:6.1f} us -> {sc.num_robots * K / med:6.1f} M")
med, best, host = timed(lambda: env.step_many(pool, 5, K, chains))
print(f"direct again: {K} ticks, {chains} ranges: region {med:7.1f} us (best {best:7.1f}), host {host:6.1f} us -> {sc.num_robots * K / med:6.1f} M")
