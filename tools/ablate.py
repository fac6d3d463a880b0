#!/usr/bin/env python3
"""GPU ablation of the kernels (profiling only): times the tick with parts switched off / launch shapes changed via
mrca_set_debug_flags so the cost split march / neighbours / writes is known.  Uses the PROFILING build of the library
(libmrca_env_prof.so, csrc/build.sh --profiling); the product library has no such switches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

sys.path.insert(0, os.path.join(ROOT, "rl-collision-avoidance_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as G  # noqa: E402

G.build()
from mrca import scenario as S  # noqa: E402
from mrca import _lib  # noqa: E402
from mrca.vec_env import VecStageWorld  # noqa: E402

for name, sc in (("stage1 128x32", S.stage1(num_worlds=128, robots_per_world=32, seed=1)),
                 ("stage2 187x44", S.stage2(num_worlds=187, seed=1))):
    env = VecStageWorld(sc, lib_path=_lib.PROFILING_LIB_PATH)
    N = sc.num_robots
    gen = torch.Generator(device=env.device).manual_seed(1)
    pool = [torch.stack([torch.rand(N, generator=gen, device=env.device),
                         torch.rand(N, generator=gen, device=env.device) * 2 - 1], 1).contiguous() for _ in range(8)]
    env.reset()
    for k in range(50):
        env.step(pool[k % 8])
    for flags, label in ((0, "full (product shape)"), (1, "no neighbour tests"), (2, "no march"), (3, "no march, no neighbours"),
                         (8, "move: no outline test"), (16, "move: no collision loop"), (32, "move: no resets"),
                         (56, "move: none of the three"),
                         (256, "1 beam/thread, 512 thr, wave 0 prepares"), (512, "2 beams sequential, wave 0 prepares (product)"),
                         (512 + 4096, "2 beams lock-step, wave 0 prepares"), (768 + 4096, "4 beams lock-step, wave 0 prepares"),
                         (768, "4 beams sequential, wave 0 prepares (2 waves / workgroup)"),
                         (256 + 2048, "1 beam/thread + dedicated prep wave"), (512 + 2048, "2 beams sequential + prep wave"),
                         (512 + 4096 + 2048, "2 beams lock-step + prep wave"),
                         (768 + 4096 + 2048, "4 beams lock-step + prep wave"), (0, "full again")):
        try:
            env.set_debug_flags(flags)
        except Exception as exc:  # a variant this build / map does not support
            print(f"{name:<14} flags={flags} {label:<44} unsupported: {exc}")
            continue
        for k in range(20):
            env.step(pool[k % 8])
        torch.cuda.synchronize()
        env.enable_timing(True)
        for k in range(300):
            env.step(pool[k % 8])
        mv, ry, n = env.read_timing()
        env.enable_timing(False)
        print(f"{name:<14} flags={flags:<5} {label:<52} ray {ry / n * 1e3:8.1f} us   move {mv / n * 1e3:7.1f} us   "
              f"sum {(ry + mv) / n * 1e3:7.1f} us")
    env.set_debug_flags(0)
    # where the move kernel's chain goes: s_memtime stamps of its phases (averaged over the worlds of one launch)
    import ctypes as C
    for k in range(20):
        env.step(pool[k % 8])
    ticks = (C.c_double * 9)()
    acc = [0.0] * 9
    for k in range(16):
        env.step(pool[k % 8])
        _lib.check(env.lib.mrca_debug_move_stamps(env._h, ticks), "mrca_debug_move_stamps")
        acc = [a + t for a, t in zip(acc, ticks)]
    names = ("state loaded + integrated", "clearance load + broad phase", "patches to LDS", "outline walks",
             "ordered collision pass", "reward / terminal / group ballots", "restarts", "stores drained", "entry -> end")
    print(f"{name:<14} move kernel phases (s_memtime ticks, memory drained at every stamp, mean of 16 launches):")
    for nm, a in zip(names, acc):
        print(f"{'':<14}   {nm:<36} {a / 16:9.1f} ticks  {100.0 * a / acc[8]:5.1f} %")
    # the ray cast: where a workgroup's time goes, wave 0 (prepares the neighbour list, then marches) vs wave 1 (marches)
    rnames = ("entry", "loads requested", "neighbour list built", "beams marched", "through the barrier",
              "neighbour slab tests", "stores issued")
    for flags, label in ((0, "overlap untouched"), (64, "memory drained at every stamp"), (64 + 1, "drained, no neighbour tests"),
                         (64 + 2, "drained, no march")):
        env.set_debug_flags(flags)
        out = (C.c_double * 17)()
        acc = [0.0] * 17
        for k in range(20):
            env.step(pool[k % 8])
        for k in range(16):
            env.step(pool[k % 8])
            _lib.check(env.lib.mrca_debug_ray_stamps(env._h, out), "mrca_debug_ray_stamps")
            acc = [a + t / 16 for a, t in zip(acc, out)]
        print(f"{name:<14} ray cast stamps, {label} (s_memtime ticks since the workgroup's entry, mean over workgroups and 16 launches):")
        for k, nm in enumerate(rnames):
            print(f"{'':<14}   {nm:<24} wave 0 {acc[k]:9.1f}   wave 1 {acc[7 + k]:9.1f}")
    env.set_debug_flags(0)
    env.close()
