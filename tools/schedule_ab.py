#!/usr/bin/env python3
"""mrca_step_many's two schedules side by side in ONE fresh process: run-ahead ("native": the move launches on a stream of
their own, ahead of the ray casts -- DESIGN.md 5.10) against round 5's chained one ("chained": `move, ray, move, ray ...` per
world range), 1 - 3 world ranges, on the four configurations bench.py quotes (Stage-2 map, fidelity mode, reference-shaped
observations, configs[1]); 300 ticks each through bench.env_side_figure.        python tools/schedule_ab.py
(profiles/r06_h_schedule_ab_fresh_process.txt)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-collision-avoidance_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mrca import scenario as S
import __graft_entry__ as G
G.build()
cases = {
 "stage2": (lambda: S.stage2(num_worlds=187, seed=1000), True),
 "stage1_fidelity": (lambda: S.stage1(num_worlds=128, robots_per_world=32, seed=1000, stage_resolution=True), True),
 "stage1_eager_views": (lambda: S.stage1(num_worlds=128, robots_per_world=32, seed=1000), False),
 "stage1": (lambda: S.stage1(num_worlds=128, robots_per_world=32, seed=1000), True),
}
for name, (mk, lazy) in cases.items():
    for sched in ("native", "chained"):
        for chains in (1, 2, 3):
            r = bench.env_side_figure(mk(), ticks=300, chains=chains, lazy_obs=lazy, schedule=sched)
            print(f"{name:20s} {sched:8s} chains={chains}: {r['value']/1e6:7.1f} M  {r['ms_per_step']*1e3:6.2f} us/tick", flush=True)
