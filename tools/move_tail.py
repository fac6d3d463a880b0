#!/usr/bin/env python3
"""The move launch lasts as long as its slowest world: the DISTRIBUTION over worlds of the time between the phase stamps of
move_kernel (profiling build, tools/ablate.py prints the means) -- mean, p90 and max per phase over the worlds of 40 launches,
and which phase the slowest world of a launch spends its extra time in."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-collision-avoidance_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as G  # noqa: E402

G.build()
import bench  # noqa: E402
from mrca import scenario as S, _lib  # noqa: E402
from mrca.vec_env import VecStageWorld  # noqa: E402

PHASES = ["state loaded + integrated", "clearance load + broad phase", "patches to LDS", "outline walks", "ordered collision pass",
          "reward / terminal / group ballots", "restarts", "stores drained"]
for name, sc in (("stage1 128x32", S.stage1(num_worlds=128, robots_per_world=32, seed=1)), ("stage2 187x44", S.stage2(num_worlds=187, seed=1))):
    env = VecStageWorld(sc, lib_path=_lib.PROFILING_LIB_PATH)
    env.lib.mrca_debug_move_stamps_raw.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    pool = bench.action_pool(sc.num_robots, env.device, 1)
    env.reset()
    for k in range(60):
        env.step(pool[k % 16])
    deltas, life = [], []
    for k in range(40):
        env.step(pool[k % 16])
        buf = np.zeros(10 * 4096, np.uint64)
        W = C.c_int32()
        _lib.check(env.lib.mrca_debug_move_stamps_raw(env._h, buf.ctypes.data, C.byref(W)), "stamps")
        st = buf[: 10 * W.value].reshape(10, W.value).astype(np.int64)
        d = st[1:9] - st[0:8]
        deltas.append(d)
        life.append(st[8] - st[0])
    d = np.stack(deltas)            # [launch, phase, world]
    life = np.stack(life)           # [launch, world]
    print(f"{name}: workgroup lifetime over worlds and 40 launches: mean {life.mean():.0f} ticks, p90 {np.percentile(life, 90):.0f}, "
          f"mean of the per-launch MAX {life.max(1).mean():.0f}")
    slow = life.argmax(1)
    for p, ph in enumerate(PHASES):
        x = d[:, p, :]
        of_slowest = x[np.arange(len(slow)), slow].mean()
        print(f"    {ph:<36} mean {x.mean():7.0f}   p90 {np.percentile(x, 90):7.0f}   max {x.max():7.0f}   in the slowest world of a launch {of_slowest:7.0f}")
    env.close()
