#!/usr/bin/env python3
"""The soak legs of tests/test_gpu_parity.py at ten times their length, once, for the record: BASELINE configs[1] / configs[2] at
full size (and configs[1] in fidelity mode) through the schedule bench.py times -- mrca_step_many, two world ranges, calls of 100
ticks, actions i.i.d. per tick -- against the C oracle stepping the same actions tick by tick; every field of every robot compared
every 100 ticks, what each beam hit at the end.  The oracle is the checker here (test infrastructure), the HIP path the thing
checked.

    python tools/long_soak.py [stage1_ticks [stage2_ticks [fidelity_ticks]]]       # defaults 4000 2000 1000
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rl-collision-avoidance_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as G  # noqa: E402

G.build()
import bench  # noqa: E402
import util as U  # noqa: E402
from mrca import scenario as S  # noqa: E402
from mrca.vec_env import VecStageWorld  # noqa: E402

args = [int(a) for a in sys.argv[1:]]
plan = [("stage1", 128, 32, (args + [4000])[0]), ("stage2", 187, 44, (args + [0, 2000])[1]),
        ("stage1_fidelity", 128, 32, (args + [0, 0, 1000])[2])]
for name, worlds, R, ticks in plan:
    if ticks <= 0:
        continue
    fid = name.endswith("_fidelity")
    sc = S.stage1(num_worlds=worlds, robots_per_world=R, seed=177, stage_resolution=fid) if name.startswith("stage1") else \
        S.stage2(num_worlds=worlds, seed=177, stage_resolution=fid)
    env = VecStageWorld(sc)
    ora = U.COracleEnv(sc)
    depth = 1000
    pool = bench.action_pool(sc.num_robots, env.device, 17, depth=depth)
    host_pool = [a.cpu().numpy() for a in pool]
    sched = bench.TickSchedule(env, pool, chains=2, native=True)
    env.reset()
    ora.reset()
    torch.cuda.synchronize()
    U.assert_state_equal(U.HostView(env), ora, what=f"{name} reset")
    t0, gpu_s, timeouts, crashes, reaches = time.time(), 0.0, 0, 0, 0
    for first in range(0, ticks, 100):
        n = min(100, ticks - first)
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        sched.run(first, n)
        torch.cuda.synchronize()
        gpu_s += time.perf_counter() - g0
        for k in range(first, first + n):
            ora.step(host_pool[k % depth])
            res = np.asarray(ora.result)
            timeouts += int((res == 3).sum())
            crashes += int((res == 2).sum())
            reaches += int((res == 1).sum())
        U.assert_state_equal(U.HostView(env), ora, what=f"{name} after {first + n} ticks")
    U.assert_hits_equal(env, ora, what=f"{name}, final tick")
    env.check()
    ep = np.asarray(ora.episode)
    print(f"{name}: {sc.num_robots} robots x {ticks} ticks = {sc.num_robots * ticks / 1e6:.1f} M agent-steps bit-identical to the C oracle "
          f"(every field every 100 ticks; hits at the end); episodes per robot {ep.min()} .. {ep.max()}, terminal events: "
          f"{reaches} goals, {crashes} crashes, {timeouts} time-outs; GPU {gpu_s:.2f} s, oracle + compare {time.time() - t0 - gpu_s:.0f} s",
          flush=True)
    env.close()
