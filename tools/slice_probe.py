#!/usr/bin/env python3
"""GPU, PROFILING build: why does ONE rank's slice of a giant world's ray cast take longer than its share of the full launch?
(79 us for 6 250 of 50 000 robots against 61.5 us pro rata, round 3.)

    python tools/slice_probe.py [--robots 50000] [--shards 8] [--lib PATH]

``--lib``: another profiling build (an experiment's).

For the launch shapes the big-world kernel is instantiated for (1 / 2 / 4 beams per marching thread: 512 / 256 / 128 threads
per workgroup; several beams one after the other or in lock step) it times the full launch and one rank's slice (begin / end stamps of the launches themselves), checks that the
slice's ring rows are bit-identical from shape to shape, and prints the slice launch's TIMELINE from the per-workgroup
s_memtime stamps of the profiling build: how many workgroups are in flight in each twentieth of the launch, when
they start, how long one takes depending on when it started.  A launch that is "ramp and tail" shows it here."""
import ctypes as C
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
          (512 + 4096, "2 beams per thread in lock step, 256 threads"), (768 + 4096, "4 beams per thread in lock step, 128 threads"))


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


def raw_stamps(env, blocks):
    out = np.zeros((2, 7, blocks), np.uint64)
    _lib.check(env.lib.mrca_debug_ray_stamps_raw(env._h, out.ctypes.data_as(C.c_void_p), blocks), "mrca_debug_ray_stamps_raw")
    return out.astype(np.int64)


def timeline(st, label, launch_us):
    """st[2, 7, blocks]: s_memtime stamps of waves 0 / 1; entry = [0, 0], end = the later of the two waves' last stamps.
    s_memtime is a PER-XCD counter (the eight dies' counters have unrelated origins) at the shader clock: workgroup b runs on
    XCD b % 8 (round-robin dispatch, profiles/r04_b_launch_boundary.txt), so every die's stamps are taken relative to that
    die's first entry, and the tick length follows from the launch's own duration."""
    blocks = st.shape[2]
    xcd = np.arange(blocks) % 8
    start = st[0, 0].copy()
    end = np.maximum(st[0, 6], st[1, 6])
    spans = []
    for g in range(8):
        m = xcd == g
        t0 = start[m].min()
        start[m] -= t0
        end[m] -= t0
        spans.append(int(end[m].max()))
    span = float(max(spans))
    if span > 1e9:
        print(f"  {label}: the stamps of a die do not share an origin (spans {spans}): workgroup -> XCD is not b % 8 here")
        return None
    tick_us = launch_us / span
    dur = (end - start).astype(np.float64)
    print(f"  {label}: {blocks} workgroups; per-XCD first entry -> last end {min(spans)} ... {max(spans)} ticks = the launch's "
          f"{launch_us:.1f} us => {1.0 / tick_us:.0f} ticks per us; a workgroup lives {dur.mean() * tick_us:.2f} us on average "
          f"(min {dur.min() * tick_us:.2f}, max {dur.max() * tick_us:.2f}); workgroup-time / span = {dur.sum() / span:.0f} workgroups "
          f"in flight on average")
    bins = 20
    edges = span * np.arange(bins + 1) / bins
    print("    twentieth   in flight (mean)   started   mean life of those started [us]")
    for b in range(bins):
        lo, hi = edges[b], edges[b + 1]
        overlap = np.clip(np.minimum(end, hi) - np.maximum(start, lo), 0, None).sum() / (hi - lo)
        started = (start >= lo) & (start < hi) if b < bins - 1 else (start >= lo)
        life = dur[started].mean() * tick_us if started.any() else float("nan")
        print(f"    {b:9d}   {overlap:16.0f}   {int(started.sum()):7d}   {life:8.2f}")
    return {"mean_life_us": dur.mean() * tick_us, "mean_in_flight": dur.sum() / span, "ticks_per_us": 1.0 / tick_us}


sc = S.circle_big(R)
print(f"library: {os.path.relpath(LIB, ROOT)}")
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
    if PER <= 8192:
        st = raw_stamps(env, PER)
        out["timeline"] = timeline(st, f"slice launch, {label}", sry)
    summary.append(out)
    env.check()
    env.close()
