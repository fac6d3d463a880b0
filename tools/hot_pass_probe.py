#!/usr/bin/env python3
"""What would a tick's ray cast cost inside a PERSISTENT kernel -- no launch boundary, the free-rectangle field and the robots'
records already in the XCDs' L2s?  The profiling build's debug flag 128 makes every workgroup of raycast_kernel cast its
robot's beams twice inside one launch; T(two passes) - T(one pass) is the cost of a pass that finds everything hot.
("make the field stay hot, then re-measure the floor".)

    python tools/hot_pass_probe.py      # profiling build; launch times by the launches' own begin / end stamps
"""
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
from mrca import _lib  # noqa: E402
from mrca.vec_env import VecStageWorld  # noqa: E402

for name, mk in (("stage1 128 x 32", lambda w: S.stage1(num_worlds=w, robots_per_world=32, seed=1)),
                 ("stage2 187 x 44", lambda w: S.stage2(num_worlds=187 * w // 128, seed=1))):
    for worlds in (64, 128):
        sc = mk(worlds)
        env = VecStageWorld(sc, lib_path=_lib.PROFILING_LIB_PATH)
        pool = bench.action_pool(sc.num_robots, env.device, 1, depth=16)
        env.reset()
        for k in range(30):
            env.step(pool[k % 16])
        res = {}
        for rep in range(2):
            for flags in (0, 128):
                env.set_debug_flags(flags)
                for k in range(10):
                    env.step(pool[k % 16])
                torch.cuda.synchronize()
                env.enable_timing(True)
                for k in range(200):
                    env.step(pool[k % 16])
                mv, ry, n = env.read_timing()
                env.enable_timing(False)
                res.setdefault(flags, []).append(ry / n * 1e3)
        env.set_debug_flags(0)
        one, two = min(res[0]), min(res[128])
        print(f"{name}: {sc.num_robots:5d} robots   one pass {one:6.2f} us   two passes in one launch {two:6.2f} us   "
              f"-> a hot pass {two - one:6.2f} us = {100 * (two - one) / one:5.1f} % of a launch", flush=True)
        env.close()
