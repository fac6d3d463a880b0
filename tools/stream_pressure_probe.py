#!/usr/bin/env python3
"""Does mrca_step_many's run-ahead schedule depend on what ELSE the process has created?  bench.py's side figures came out
at 140 - 200 M on the Stage-2 map where the same function measured 310 M in a fresh process (profiles/r06_h_*).  This probe
times bench.env_side_figure(stage2, native, 2 ranges) after creating K extra streams / another env / a captured graph."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-collision-avoidance_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as G  # noqa: E402

G.build()
import bench  # noqa: E402
from mrca import scenario as S  # noqa: E402
from mrca.vec_env import VecStageWorld  # noqa: E402


def fig(tag):
    r = bench.env_side_figure(S.stage2(num_worlds=187, seed=1000), ticks=300, chains=2, schedule="native")
    c = bench.env_side_figure(S.stage2(num_worlds=187, seed=1000), ticks=300, chains=2, schedule="chained")
    print(f"{tag:60s} run-ahead {r['value'] / 1e6:7.1f} M   chained {c['value'] / 1e6:7.1f} M", flush=True)


what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("all", "fresh"):
    fig("fresh process")
keep = []
if what in ("all", "streams"):
    for k in (1, 2, 4, 8):
        while len(keep) < k:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                torch.zeros(8, device="cuda").add_(1)
            keep.append(s)
        torch.cuda.synchronize()
        fig(f"{k} extra torch streams alive")
if what in ("all", "env"):
    e = VecStageWorld(S.stage1(num_worlds=128, robots_per_world=32, seed=1000))
    e.reset()
    pool = bench.action_pool(e.N, e.device, 1, depth=32)
    e.step_many(pool, 0, 20, 2)
    torch.cuda.synchronize()
    fig("+ another env alive that has run mrca_step_many")
if what in ("all", "graph", "graph-first"):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    x = torch.zeros(1 << 20, device="cuda")
    with torch.cuda.graph(g, stream=side):
        for _ in range(8):
            x.add_(1.0)
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    fig("+ a captured hipGraph replayed 50 times")
