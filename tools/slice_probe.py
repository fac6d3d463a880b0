#!/usr/bin/env python3
"""GPU, PROFILING build: why does ONE rank's slice of a giant world's ray cast take longer than its share of the full launch?
(79 us for 6 250 of 50 000 robots against 61.5 us pro rata, round 3.)

    python tools/slice_probe.py [--robots 50000] [--shards 8] [--lib PATH] [--sweep | --phases]

``--lib``: another profiling build (an experiment's).  ``--sweep``: the product's shape only, slices of 1024 ... R robots -- launch
time against slice size (steps at multiples of the 4096 workgroups the chip holds = residency rounds).

For the launch shapes the big-world kernel is instantiated for (1 / 2 / 4 beams per marching thread: 512 / 256 / 128 threads
per workgroup; several beams one after the other or in lock step) it times the full launch and one rank's slice (begin / end
stamps of the launches themselves) and checks that the slice's ring rows are bit-identical from shape to shape."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-collision-avoidance_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as G  # noqa: E402

G.build()
from mrca import _lib  # noqa: E402
from mrca import scenario as S  # noqa: E402
from mrca.vec_env import VecStageWorld  # noqa: E402

argv = sys.argv[1:]


def opt(name, default):
    if name in argv:
        i = argv.index(name)
        v = int(argv[i + 1])
        del argv[i: i + 2]
        return v
    return default


LIB = _lib.PROFILING_LIB_PATH
if "--lib" in argv:
    i = argv.index("--lib")
    LIB = os.path.abspath(argv[i + 1])
    del argv[i: i + 2]
R = opt("--robots", 50000)
SHARDS = opt("--shards", 8)
PER = -(-R // SHARDS)
SHAPES = ((768, "4 beams per thread one after the other (product), 128 threads"),
          (512, "2 beams per thread one after the other, 256 threads"), (256, "1 beam per thread, 512 threads"),
          (512 + 4096, "2 beams per thread in lock step, 256 threads"), (768 + 4096, "4 beams per thread in lock step, 128 threads"),
          (768 + 2048, "4 beams one after the other + a dedicated preparation wave, 192 threads"),
          (512 + 2048, "2 beams one after the other + a dedicated preparation wave, 320 threads"))


def controller(env):
    lg = env.local_goal
    bearing = torch.atan2(lg[:, 1], lg[:, 0])
    return torch.stack([torch.ones_like(bearing), torch.clamp(2.0 * bearing, -1, 1)], 1).contiguous()


def timed(env, n, ray_slice=None):
    env.enable_timing(1)
    torch.cuda.synchronize()
    for _ in range(n):
        env.step(controller(env), ray_slice=ray_slice)
    torch.cuda.synchronize()
    mv, ry, k = env.read_timing()
    env.enable_timing(False)
    return mv / k * 1e3, ry / k * 1e3


sc = S.circle_big(R)
print(f"library: {os.path.relpath(LIB, ROOT)}")
if "--phases" in argv:
    # where a workgroup's time goes in the big-world launch (the product's shape): s_memtime stamps of waves 0 and 1, means over
    # the first 8192 workgroups, full launch and a 1024-robot slice (a quarter of the chip's residency: latency laid bare)
    import ctypes as C
    names = ("entry", "loads requested", "neighbour list built", "beams marched", "through the barrier", "neighbour slab tests",
             "stores issued")
    env = VecStageWorld(sc, lib_path=LIB)
    env.reset()
    for _ in range(20):
        env.step(controller(env))
    for label, sl in (("full launch", None), ("slice of 1024 robots", (0, 1024))):
        for flags, fl in ((768, "overlap untouched"), (768 + 64, "memory drained at every stamp")):
            env.set_debug_flags(flags)
            acc = [0.0] * 17
            for k in range(8):
                env.step(controller(env), ray_slice=sl)
                out = (C.c_double * 17)()
                _lib.check(env.lib.mrca_debug_ray_stamps(env._h, out), "mrca_debug_ray_stamps")
                acc = [a + t / 8 for a, t in zip(acc, out)]
            print(f"{label}, {fl} (s_memtime ticks since the workgroup's entry):")
            for k, nm in enumerate(names):
                print(f"    {nm:<24} wave 0 {acc[k]:9.1f}   wave 1 {acc[7 + k]:9.1f}")
    env.check()
    env.close()
    sys.exit(0)
if "--sweep" in argv:
    # the product's shape, slices of growing size: does the launch time follow the NUMBER OF RESIDENCY ROUNDS (4096 two-wave
    # workgroups fit the chip at eight waves per SIMD) rather than the number of robots?
    env = VecStageWorld(sc, lib_path=LIB)
    env.reset()
    for _ in range(20):
        env.step(controller(env))
    torch.cuda.synchronize()
    for n in (1024, 2048, 3072, 4096, 4224, 5120, 6144, 6250, 7168, 8192, 8320, 10240, 12288, 12416, 16384, 16512, 25000, R):
        if n > R:
            continue
        _mv, ry = timed(env, 40, ray_slice=(0, n))
        print(json.dumps({"robots": R, "slice_robots": n, "raycast_us": ry, "rounds_of_4096": n / 4096.0,
                          "us_per_1000_robots": ry * 1000.0 / n}), flush=True)
    env.check()
    env.close()
    sys.exit(0)
rows = {}
summary = []
for knob, label in SHAPES:
    env = VecStageWorld(sc, lib_path=LIB)
    env.set_debug_flags(knob)
    env.reset()
    for _ in range(20):
        env.step(controller(env))
    torch.cuda.synchronize()
    mv, ry = timed(env, 60)
    smv, sry = timed(env, 60, ray_slice=(0, PER))
    # 140 ticks in: the slice's newest rows + heads, the same tick under every shape
    env.step(controller(env), ray_slice=(0, PER))
    torch.cuda.synchronize()
    ring = env.scan_ring[:PER].cpu().numpy().view(np.uint32)
    head = env.ring_head[:PER].cpu().numpy()
    rows[knob] = (ring, head)
    same = all(np.array_equal(ring, r) and np.array_equal(head, h) for r, h in rows.values())
    out = {"robots": R, "shape": label, "full": {"move_phase_us": mv, "raycast_us": ry},
           "one_rank_of_%d" % SHARDS: {"slice_robots": PER, "move_phase_us": smv, "raycast_us": sry,
                                       "pro_rata_us": ry * PER / R, "projected_speedup": (mv + ry) / (smv + sry)},
           "slice_rows_equal_to_the_other_shapes": bool(same)}
    print(json.dumps(out), flush=True)
    summary.append(out)
    env.check()
    env.close()
